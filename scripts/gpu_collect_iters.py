"""Per-instance ADMM iteration counts of consecutive solves of the bench workload -> gpurun_out/iters_b{B}.npy ([steps][B], int32):
the input of the scheduling studies (scripts/lpt_study.py) that run on the CPU afterwards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth
B, N, S = int(os.environ.get("QRW_EXP_B", "4096")), 16, 48
sb = synth.SyntheticBatch(B, N, n_seq=S)
g = qrw_hip.Batch(B, N)
out = np.zeros((S, B), np.int32)
for s in range(S):
    d = sb.step(s)
    g.mpc_solve_host(d["xref"], d["fsteps"], s)
    out[s] = g.mpc_stats()["iters"]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "iters_b%d.npy" % B), out)
print(out.mean(1)[:8], out.max(1))
