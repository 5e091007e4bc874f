"""What predicts an instance's ADMM iteration count? Conditional means on simple features of the new inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth
B, N = 4096, 16
sb = synth.SyntheticBatch(B, N, n_seq=40)
g = qrw_hip.Batch(B, N)
prev_it = prev_fs = None
rows = []
for s in range(36):
    d = sb.step(s)
    g.mpc_solve_host(d["xref"], d["fsteps"], s)
    it = g.mpc_stats()["iters"].astype(float)
    gait = (d["fsteps"].reshape(B, -1, 4, 3)[:, :N, :, 0] != 0).astype(int)   # (B, N, 4)
    if prev_it is not None and s >= 4:
        sw0 = (gait[:, 0] != prev_gait[:, 0]).any(1)            # contact set of the first step changed since the last solve
        sw1 = (gait[:, 1] != gait[:, 0]).any(1)                 # a switch between horizon steps 0 and 1
        nsw = (gait[:, 1:] != gait[:, :-1]).any(2).sum(1)       # number of switches inside the horizon
        vnorm = np.abs(d["xref"][:, 6:9, 1]).sum(1)
        rows.append(np.stack([it, prev_it, sw0, sw1, nsw, vnorm], 1))
    prev_it, prev_gait = it, gait
R = np.concatenate(rows)
it, pit, sw0, sw1, nsw, vn = R.T
print("overall mean %.0f" % it.mean())
for name, f in (("first-step contacts changed", sw0), ("switch between steps 0-1", sw1)):
    print("%s: yes %.0f (n=%d)  no %.0f (n=%d)" % (name, it[f == 1].mean(), (f == 1).sum(), it[f == 0].mean(), (f == 0).sum()))
for k_ in np.unique(nsw):
    print("switches in horizon = %d: mean %.0f n=%d" % (k_, it[nsw == k_].mean(), (nsw == k_).sum()))
print("corr(it, prev) %.3f  corr(it, |v|) %.3f  corr(it, sw0) %.3f corr(it, sw1) %.3f" % (np.corrcoef(it, pit)[0, 1], np.corrcoef(it, vn)[0, 1], np.corrcoef(it, sw0)[0, 1], np.corrcoef(it, sw1)[0, 1]))
# long instances
big = it > 1500
print("share of >1500-iteration solves with sw0: %.2f, with sw1: %.2f; base rates %.2f %.2f" % (sw0[big].mean(), sw1[big].mean(), sw0.mean(), sw1.mean()))
