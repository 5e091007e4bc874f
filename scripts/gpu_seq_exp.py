"""Experiment: qrw_mpc_solve_sequence against consecutive qrw_mpc_solve calls on the bench workload (batch 4096, N = 16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip, synth
B, N, W = 4096, 16, 4
Ks = [int(a) for a in sys.argv[1:]] or [1, 2, 5, 20]
KM = max(Ks)
sb = synth.SyntheticBatch(B, N, N_gait=20, gaits=("trot",), n_seq=W + KM)
steps = [sb.step(s) for s in range(W + KM)]
dev = torch.device("cuda", 0)
xs = torch.from_numpy(np.stack([st["xref"] for st in steps])).to(dev)
fs = torch.from_numpy(np.stack([st["fsteps"] for st in steps])).to(dev)
for K in Ks:
    for mode in ("calls", "sequence"):
        eng = qrw_hip.Batch(B, n_steps=N, N_gait=20)
        out = torch.empty((K, B, 24, N), dtype=torch.float64, device=dev)
        its = torch.zeros((K, B), dtype=torch.int32, device=dev)
        for s in range(W):
            eng.mpc_solve(xs[s], fs[s], s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "calls":
            for i in range(K):
                eng.mpc_solve(xs[W + i], fs[W + i], W + i, out=out[i])
        else:
            eng.mpc_solve_sequence(xs[W:W + K].contiguous(), fs[W:W + K].contiguous(), W, out=out, iters=its)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if mode == "sequence": eng.mpc_sequence_timed_out()
        print("K=%2d %-8s %.3f ms per call (%.0f solves/s)%s" % (K, mode, 1e3 * el / K, B * K / el,
              "  mean iters %.0f" % its.float().mean().item() if mode == "sequence" else ""), flush=True)
        eng.close()
        if mode == "sequence" and K >= 5:
            it = its.cpu().numpy().astype(np.float64)
            chain = it.sum(axis=0)
            top = np.sort(chain)[::-1][:6]
            print("      per-instance chains over %d calls: mean %.0f iterations, longest %s; per-call maxima %s" % (
                K, chain.mean(), top.astype(int).tolist(), it.max(axis=1).astype(int).tolist()), flush=True)
            print("      work bound %.2f ms/call at 2.31 us per iteration on 1024 slots; longest chain %.2f ms/call" % (
                it.sum() * 2.31e-3 / 1024 / K, top[0] * 2.31e-3 / K), flush=True)
