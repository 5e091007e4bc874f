"""Host time of one Controller_batch.compute call (enqueue only, no synchronisation) at batch 64: what the Python / ctypes side of a
control iteration costs, apart from the GPU work."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
from Controller import Controller_batch
B = 64
dev = torch.device("cuda:0")
q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
for mp in (False, True):
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        ctl = Controller_batch(B, q_init, multiprocessing=mp)
        vref = torch.zeros((B, 6), dtype=torch.float64, device=dev); vref[:, 0] = 0.2
        qf = torch.zeros((B, 19), dtype=torch.float64, device=dev); qf[:, 2], qf[:, 6] = 0.2229, 1.0
        qf[:, 7:] = torch.from_numpy(q_init).to(dev)
        vf = torch.zeros((B, 18), dtype=torch.float64, device=dev); vf[:, :6] = vref
        rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev); vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)
        for k in range(30):
            ctl.compute(vref, qf, vf, rpy, vs)
        torch.cuda.synchronize()
        ts = []
        for k in range(30, 230):
            t0 = time.perf_counter()
            ctl.compute(vref, qf, vf, rpy, vs)
            ts.append((time.perf_counter() - t0, k % 10 == 0))
            if k % 20 == 19: torch.cuda.synchronize()
        ns = np.array([t for t, s in ts if not s]) * 1e6
        so = np.array([t for t, s in ts if s]) * 1e6
        print("multiprocessing=%s: host time per compute(): non-solving median %.1f us (p90 %.1f), solving median %.1f us" % (mp, np.median(ns), np.percentile(ns, 90), np.median(so)))
