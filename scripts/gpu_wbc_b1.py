import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch, qrw_hip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sb = synth.SyntheticBatch(B, 16, n_seq=12)
eng = qrw_hip.Batch(B, 16)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
mpc_out = wbc_out = None
for s in range(10):
    d = sb.step(s)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    x, f = t(d["xref"]), t(d["fsteps"])
    a = [t(d[k]) for k in ("q", "dq", "contacts", "pgoals", "vgoals", "agoals")]
    torch.cuda.synchronize()
    e[0].record()
    mpc_out = eng.mpc_solve(x, f, s, out=mpc_out)
    e[1].record()
    fc = mpc_out[:, 12:, 0].contiguous()
    e[2].record()
    wbc_out = eng.wbc_compute(a[0], a[1], fc, a[2], a[3], a[4], a[5], out=wbc_out)
    e[3].record()
    torch.cuda.synchronize()
    print(s, "mpc %.3f ms  copy %.3f ms  wbc %.3f ms" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])),
          "wbc iters", eng.wbc_stats()["iters"][:2], "mpc iters", eng.mpc_stats()["iters"][:2])
