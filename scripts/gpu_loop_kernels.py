"""A few synchronous Controller_batch iterations at batch 4096 (run under rocprofv3 --kernel-trace --stats to see the
duration of the loop's kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
from Controller import Controller_batch
B = int(os.environ.get("QRW_EXP_B", "4096"))
dev = torch.device("cuda:0")
q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
ctl = Controller_batch(B, q_init, groups=1, fused=(len(sys.argv) < 2 or sys.argv[1] != 'separate'))
vref = torch.zeros((B, 6), dtype=torch.float64, device=dev); vref[:, 0] = 0.4
qf = torch.zeros((B, 19), dtype=torch.float64, device=dev); qf[:, 2], qf[:, 6] = 0.2229, 1.0
qf[:, 7:] = torch.from_numpy(q_init).to(dev)
vf = torch.zeros((B, 18), dtype=torch.float64, device=dev); vf[:, :6] = vref
rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev); vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)
for _ in range(40):
    r = ctl.compute(vref, qf, vf, rpy, vs)
    qf[:, 7:].copy_(r.q_des); vf[:, 6:].copy_(r.v_des)
torch.cuda.synchronize()
