"""Soak of qrw_mpc_solve_sequence against consecutive qrw_mpc_solve calls at full size: every word of every call's result
and the iteration counts must be identical (the hand-off between workgroups is the risk: stale L1 / cross-XCD L2 lines show
up as rare wrong words under uneven load).  python scripts/gpu_seq_soak.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
bad = 0
for (N, B, K, gaits) in [(16, 4096, 20, ("trot",)), (16, 3000, 12, ("walk", "trot", "bounding", "pacing")), (32, 1024, 6, ("walk", "trot", "bounding"))]:
    Ng = max(20, N + 4)
    sb = synth.SyntheticBatch(B, N, N_gait=Ng, gaits=gaits, n_seq=rounds * K + 1, seed0=20290000 + N)
    a, b = (qrw_hip.Batch(B, n_steps=N, N_gait=Ng, T_gait=0.02 * N) for _ in range(2))
    call = 0
    for r in range(rounds):
        steps = [sb.step(call + s) for s in range(K)]
        xs = torch.from_numpy(np.stack([st["xref"] for st in steps])).to(dev)
        fs = torch.from_numpy(np.stack([st["fsteps"] for st in steps])).to(dev)
        ref = torch.empty((K, B, 24, N), dtype=torch.float64, device=dev)
        rit = torch.zeros((K, B), dtype=torch.int32, device=dev)
        for s in range(K):
            a.mpc_solve(xs[s], fs[s], call + s, out=ref[s])
            a.copy_mpc_iters(rit[s])
        its = torch.zeros((K, B), dtype=torch.int32, device=dev)
        out = b.mpc_solve_sequence(xs, fs, call, iters=its)
        torch.cuda.synchronize()
        same = bool(torch.equal(out.view(torch.int64), ref.view(torch.int64))) and bool(torch.equal(its, rit))
        n_diff = int((out.view(torch.int64) != ref.view(torch.int64)).sum().item())
        print("N=%d B=%d K=%d round %d: %s (differing words %d, timed out %s, mean iters %.0f)" % (
            N, B, K, r, "identical" if same else "MISMATCH", n_diff, b.mpc_sequence_timed_out(), its.float().mean().item()), flush=True)
        bad += 0 if same else 1
        call += K
    a.close(); b.close()
print("soak:", "ok" if bad == 0 else "%d mismatching rounds" % bad)
sys.exit(0 if bad == 0 else 1)
