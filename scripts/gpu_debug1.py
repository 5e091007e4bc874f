"""First GPU bring-up: MFMA layout self-test, MPC and WBC vs the oracle on a few instances."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import qrw_hip, oracle, synth
np.set_printoptions(precision=6, suppress=True, linewidth=220)
print("mfma selftest:", qrw_hip.selftest_mfma(), flush=True)
B, N = 8, 16
sb = synth.SyntheticBatch(B, N, seed0=20260000)
g = qrw_hip.Batch(B, N)
print("Y diag", g.base_inertia_diag())
om = [oracle.MPC(0.02, N, 0.32, 20) for _ in range(B)]
x0 = None
for s in range(6):
    d = sb.step(s, x0)
    t0 = time.time(); out = g.mpc_solve_host(d["xref"], d["fsteps"], s); t1 = time.time()
    st = g.mpc_stats()
    ref = np.zeros_like(out); its = []
    for b in range(B):
        om[b].run(s, d["xref"][b], d["fsteps"][b]); ref[b] = om[b].get_latest_result(); its.append(om[b].iter)
    err = np.abs(out - ref).max(axis=(1, 2)); scale = np.abs(ref).max(axis=(1, 2))
    print("step", s, "gpu iters", st["iters"], "status", st["status"], "rho", st["rho"][:3], "\n   ora iters", its, "rho", [round(o.rho, 6) for o in om[:3]],
          "\n   max abs err", err.max(), "rel", (err / scale).max(), "time %.1f ms" % ((t1 - t0) * 1e3), flush=True)
    if s == 0:
        stt = g.mpc_state(0); x, z, y = om[0].iterates()
        print("   state x err", np.abs(stt["x"] - x).max(), "z", np.abs(stt["z"] - z).max(), "y", np.abs(stt["y"] - y).max())
    x0 = ref[:, :12, 0]
# WBC
ow = [oracle.WbcController(0.002) for _ in range(B)]
for s in range(4):
    d = sb.step(s)
    c = d["contacts"]; f = np.zeros((B, 12)); f[:, 2::3] = c * 24.5 / np.maximum(c.sum(1, keepdims=True), 1); f[:, 0::3] = c * 0.5
    o = g.wbc_compute_host(d["q"], d["dq"], f, c, d["pgoals"], d["vgoals"], d["agoals"])
    st = g.wbc_stats(); e = {}
    for b in range(B):
        ow[b].compute(d["q"][b], d["dq"][b], f[b], c[b], d["pgoals"][b], d["vgoals"][b], d["agoals"][b])
        for kname, ref in (("tau_ff", ow[b].tau_ff), ("qdes", ow[b].qdes), ("vdes", ow[b].vdes[:, 0]), ("f_with_delta", ow[b].f_with_delta[:, 0]), ("ddq_res", ow[b].ddq_res)):
            e[kname] = max(e.get(kname, 0), np.abs(o[kname][b] - ref).max())
    print("wbc step", s, "iters", st["iters"], "ora", [w.qp_iter for w in ow], "err", {k: float("%.3g" % v) for k, v in e.items()}, flush=True)
