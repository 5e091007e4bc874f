"""Dump the ADMM iteration counts of 40 consecutive solves of the headline workload (B=4096, N=16, trot) for offline study
of block-order predictors (gpurun_out/iter_trace.npy, shape (steps, B))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth
B, N, S = 4096, 16, 40
sb = synth.SyntheticBatch(B, N, n_seq=S + 1)
g = qrw_hip.Batch(B, N)
its = np.zeros((S, B), dtype=np.int32)
for s in range(S):
    d = sb.step(s)
    g.mpc_solve_host(d["xref"], d["fsteps"], s)
    its[s] = g.mpc_stats()["iters"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "iter_trace.npy"), its)
print(its.mean(1)[:24])
