"""Study for the wild-input WBC parity test: per (instance, call) the deviation HIP vs strict oracle, strict vs fast oracle (two builds
of the same source: the rounding sensitivity of the inputs themselves), iteration counts and rho of all three."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle, qrw_hip, synth
oracle.build(fast=False); oracle.build(fast=True)
B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 10
gen = synth.RandomWbcInputs(B, seed0=40000000)
eng = qrw_hip.Batch(B)
a, b = oracle.WbcBatch(B, 0.002, fast=False), oracle.WbcBatch(B, 0.002, fast=True)
def dev(x, y):
    e = np.zeros(B)
    for u, v in zip(x, y):
        e = np.maximum(e, np.abs(u - v).reshape(B, -1).max(1) / np.maximum(np.abs(v).reshape(B, -1).max(1), 1e-12))
    return e
tainted = np.zeros(B, bool)
for c in range(K):
    d = gen.step(c)
    args = (d["q"], d["dq"], d["f_cmd"], d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
    o = eng.wbc_compute_host(*args)
    ra, rb = a.compute(*args, 16), b.compute(*args, 16)
    st = eng.wbc_stats()
    ia, sa, rhoa = a.qp_stats(); ib, sb, rhob = b.qp_stats()
    oh = (o["tau_ff"], o["qdes"], o["vdes"], o["f_with_delta"])
    e_hs, e_sf = dev(oh, ra), dev(rb, ra)
    rho_sf = np.abs(rhob / rhoa - 1); rho_hs = np.abs(st["rho"] / rhoa - 1)
    tainted |= (e_sf > 1e-9) | (ia != ib) | (rho_sf > 1e-9)
    bad = (e_hs >= 1e-4) | (st["iters"] != ia)
    print("call %d: HIP vs strict: iter mismatches %d, e>=1e-4: %d, e>=1e-6: %d, worst %.2e | strict vs fast: iter mismatches %d, e>=1e-4: %d, e>=1e-6: %d, worst %.2e | rho dev max hs %.1e sf %.1e | tainted so far %d | bad & untainted: %s"
          % (c, int((st["iters"] != ia).sum()), int((e_hs >= 1e-4).sum()), int((e_hs >= 1e-6).sum()), e_hs.max(), int((ia != ib).sum()), int((e_sf >= 1e-4).sum()), int((e_sf >= 1e-6).sum()), e_sf.max(),
             rho_hs.max(), rho_sf.max(), int(tainted.sum()), np.nonzero(bad & ~tainted)[0][:10].tolist()), flush=True)
    for k in np.nonzero(bad)[0][:6]:
        print("    instance %d: iters hip %d strict %d fast %d; rho hip %.8g strict %.8g fast %.8g; e_hs %.2e e_sf %.2e tainted %s" % (k, st["iters"][k], ia[k], ib[k], st["rho"][k], rhoa[k], rhob[k], e_hs[k], e_sf[k], tainted[k]))
