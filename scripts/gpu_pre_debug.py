import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip, synth
N, B, chunk = 32, 1300, 600
Ng = 36
sb = synth.SyntheticBatch(B, N, N_gait=Ng, gaits=("walk", "trot", "bounding"), seed0=20300000 + N)
os.environ["QRW_PREEMPT_CHUNK"] = "0"
plain = qrw_hip.Batch(B, n_steps=N, N_gait=Ng, T_gait=0.02 * N)
os.environ["QRW_PREEMPT_CHUNK"] = str(chunk); os.environ["QRW_PREEMPT_MIN_BATCH"] = "8"
sliced = qrw_hip.Batch(B, n_steps=N, N_gait=Ng, T_gait=0.02 * N)
for s in range(3):
    d = sb.step(s)
    x, f = torch.from_numpy(d["xref"]).cuda(), torch.from_numpy(d["fsteps"]).cuda()
    a = plain.mpc_solve(x, f, s).cpu().numpy(); b = sliced.mpc_solve(x, f, s).cpu().numpy()
    sa, sb_ = plain.mpc_stats(), sliced.mpc_stats()
    bad = np.nonzero(sa["rho"] != sb_["rho"])[0]
    print("step", s, "iters equal", np.array_equal(sa["iters"], sb_["iters"]), "rho differs on", len(bad), "of", B,
          "| out differs on", int((np.abs(a - b).max(axis=(1, 2)) > 0).sum()))
    for i in bad[:8]:
        print("   inst %d iters %d rho %.17g vs %.17g rel %.2e  out maxdiff %.2e  pri %.3e/%.3e" % (i, sa["iters"][i], sa["rho"][i], sb_["rho"][i],
              abs(sa["rho"][i] / sb_["rho"][i] - 1), np.abs(a[i] - b[i]).max(), sa["pri_res"][i], sb_["pri_res"][i]))
    it = sa["iters"]
    print("   iters of differing:", np.sort(it[bad])[:10], "... min iters among differing", it[bad].min() if len(bad) else None,
          "| #instances with iters>chunk:", int((it > chunk).sum()))
