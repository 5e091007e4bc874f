"""Phase clocks of wbc_kernel (diagnostic build -DQRW_PROFILE_WBC, build/lib_wbcprof.so): a synchronous Controller_batch loop at
batch 4096, stamps of the last launch, mean over the 256 workgroups.  gpurun -- python3 scripts/gpu_wbc_phases.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["QRW_HIP_LIB"] = os.environ.get("QRW_HIP_LIB", os.path.join(ROOT, "build", "lib_wbcprof.so"))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip
from Controller import Controller_batch
B = 4096
dev = torch.device("cuda:0")
q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
ctl = Controller_batch(B, q_init, groups=1)
rng = np.random.default_rng(3)
vref = torch.from_numpy(rng.uniform(-0.4, 0.4, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.0])).to(dev)
qf = torch.zeros((B, 19), dtype=torch.float64, device=dev); qf[:, 2], qf[:, 6] = 0.2229, 1.0
qf[:, 7:] = torch.from_numpy(q_init).to(dev)
vf = torch.zeros((B, 18), dtype=torch.float64, device=dev); vf[:, :6] = vref
rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev); vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)
lib = qrw_hip.load_library()
names = ["prologue + input loads", "leg kinematics, InvKin, Newton-Euler 1, QP data, early outputs", "qp_build (H, g, bounds)",
         "warm-start loads + ten equilibration passes", "KKT inverse (12 pivots in the quad) + ADMM loop", "solution + state stores",
         "epilogue: kinematics again, Newton-Euler 2, torques, result check"]
for it in range(38):
    r = ctl.compute(vref, qf, vf, rpy, vs)
    qf[:, 7:].copy_(r.q_des); vf[:, 6:].copy_(r.v_des)
    if it in (5, 21, 37):
        torch.cuda.synchronize()
        nb = B // 16 if os.environ.get("QRW_WBC16", "1") == "0" else B // 4  # wbc_kernel: 16 instances per workgroup, wbc16_kernel: 4
        buf = (C.c_ulonglong * (16 * nb))()
        assert lib.qrw_wbc_get_phase_cycles(buf, nb) == 0
        a = np.array(buf, dtype=np.float64).reshape(nb, 16)
        d = np.diff(a[:, :8], axis=1)
        print("iteration %d: %.0f clocks per workgroup in all; ADMM iterations per wavefront (max of its 16 instances): mean %.1f max %d; per instance mean %.1f"
              % (it, (a[:, 7] - a[:, 0]).mean(), a[:, 8].mean(), a[:, 8].max(), ctl.stats()["wbc"]["iters"].mean()))
        for i in range(7):
            print("   %-70s %8.0f  (%4.1f %%)" % (names[i], d[:, i].mean(), 100 * d[:, i].sum() / (a[:, 7] - a[:, 0]).sum()))
