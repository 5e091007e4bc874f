"""Experiment: the batch cut into S independent groups (own handle, own stream), each group stepping on its own with no
cross-group synchronisation inside the timed region, so that one group's straggling solves overlap the other groups'
next steps.  python scripts/gpu_subbatch_exp.py [S ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip, synth
B, N, W, K = 4096, int(os.environ.get("QRW_EXP_N", "16")), 4, int(os.environ.get("QRW_EXP_K", "20"))
NG = max(20, N + 4)
sb = synth.SyntheticBatch(B, N, N_gait=NG, gaits=("trot",) if N <= 16 else ("walk", "trot", "bounding"), n_seq=W + K)
steps = [sb.step(s) for s in range(W + K)]
dev = torch.device("cuda", 0)
keys = ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    Bs = B // S
    engs = [qrw_hip.Batch(Bs, n_steps=N, N_gait=NG, T_gait=0.02 * N) for _ in range(S)]
    if os.environ.get("QRW_EXP_OWN_STREAMS") == "1":
        own = [qrw_hip.CuStream(0, 0, 0) for _ in range(S)]  # hipStreamCreateWithFlags(hipStreamNonBlocking), one each
        streams = [o.torch for o in own]
    else:
        streams = [torch.cuda.Stream(dev, priority=0) for _ in range(S)]
    data = [[{k: torch.from_numpy(np.ascontiguousarray(st[k][g * Bs:(g + 1) * Bs])).to(dev) for k in keys} for st in steps] for g in range(S)]
    outs = [torch.empty((Bs, 24, N), dtype=torch.float64, device=dev) for _ in range(S)]
    fcs = [torch.empty((Bs, 12), dtype=torch.float64, device=dev) for _ in range(S)]
    wbs = [None] * S
    def step(s):
        for g in range(S):
            with torch.cuda.stream(streams[g]):
                d = data[g][s]
                engs[g].mpc_solve(d["xref"], d["fsteps"], s, out=outs[g])
                fcs[g].copy_(outs[g][:, 12:, 0])
                wbs[g] = engs[g].wbc_compute(d["q"], d["dq"], fcs[g], d["contacts"], d["pgoals"], d["vgoals"], d["agoals"], out=wbs[g])
    torch.cuda.synchronize()
    for s in range(W): step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): step(W + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("S=%d groups of %d: %.0f steps/s (%.3f ms per step of the whole batch)" % (S, Bs, B * K / el, 1e3 * el / K), flush=True)
    del engs
