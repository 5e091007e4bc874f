import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch, synth
from Controller import Controller_batch
B = 4096
dev = torch.device("cuda:0")
sb = synth.SyntheticBatch(B, 16)
q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
for scale in (1.0, 0.5, 0.3):
    ctl = Controller_batch(B, q_init)
    vref = torch.from_numpy(np.ascontiguousarray(sb.vref * scale)).to(dev)
    qf = torch.zeros((B, 19), dtype=torch.float64, device=dev); qf[:, 2], qf[:, 6] = 0.2229, 1.0
    qf[:, 7:] = torch.from_numpy(q_init).to(dev)
    vf = torch.zeros((B, 18), dtype=torch.float64, device=dev); vf[:, :6] = vref
    rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev); vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    hist = []
    for k in range(300):
        r = ctl.compute(vref, qf, vf, rpy, vs)
        qf[:, 7:].copy_(r.q_des); vf[:, 6:].copy_(r.v_des); vs.copy_(r.v_des)
        if k in (19, 59, 99, 199, 299):
            f = ctl.error_flag.cpu().numpy()
            hist.append((k, [int((f == i).sum()) for i in range(4)]))
    print("vref scale", scale, hist, "mean mpc iters", ctl.stats()["mpc"]["iters"].mean())
