"""How many N = 32 instances are resident per compute unit?  Times qrw_mpc_solve for batches of 256 / 512 / 1024 identical
instances (same seed: every instance takes the same number of iterations): with two workgroups per compute unit 256 and 512
take the same time and 1024 twice as long; with one, 512 already takes twice the time of 256."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip, synth
N = int(os.environ.get("QRW_OCC_N", "32"))
Ng = max(20, N + 4)
sb = synth.SyntheticBatch(1, N, N_gait=Ng, gaits=("trot",))
d = [sb.step(s) for s in range(4)]
for B in (256, 512, 1024, 2048):
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=Ng, T_gait=0.02 * N)
    ts = []
    for s in range(4):
        x = torch.from_numpy(np.repeat(d[s]["xref"], B, 0)).cuda()
        f = torch.from_numpy(np.repeat(d[s]["fsteps"], B, 0)).cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.mpc_solve(x, f, s); torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    it = eng.mpc_stats()["iters"]
    print("N=%d B=%4d: ms per call %s, iterations of the last call %d..%d, us per iteration (last call) %.3f" % (
        N, B, " ".join("%.2f" % t for t in ts), it.min(), it.max(), 1e3 * ts[-1] / it.max()))
    eng.close()
