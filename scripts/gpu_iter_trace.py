"""Dump the ADMM iteration counts of 40 consecutive solves of the headline workload (B=4096, N=16, trot) for offline study
of block-order predictors (gpurun_out/iter_trace.npy, shape (steps, B))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth
B, N, S = 4096, int(os.environ.get("QRW_TRACE_N", "16")), int(os.environ.get("QRW_TRACE_S", "40"))
gaits = tuple(os.environ.get("QRW_TRACE_GAITS", "trot").split(","))
Ng = max(20, N + 4)
sb = synth.SyntheticBatch(B, N, N_gait=Ng, gaits=gaits, n_seq=S + 1)
g = qrw_hip.Batch(B, N, N_gait=Ng, T_gait=0.02 * N)
its = np.zeros((S, B), dtype=np.int32)
for s in range(S):
    d = sb.step(s)
    g.mpc_solve_host(d["xref"], d["fsteps"], s)
    its[s] = g.mpc_stats()["iters"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "iter_trace_n%d.npy" % N), its)
np.save(os.path.join(ROOT, "gpurun_out", "iter_trace_n%d_kind.npy" % N), np.stack([sb.kind, sb.phase]))
np.save(os.path.join(ROOT, "gpurun_out", "iter_trace_n%d_vref.npy" % N), sb.vref)
print(its.mean(1)[:24])
