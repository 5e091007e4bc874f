"""One-off soak: GPU vs CPU oracle on many instances and steps — iteration counts, status and results."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle, qrw_hip, synth
oracle.build(fast=False)
B, N, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(os.environ.get("QRW_SOAK_N", "16")), 8
NG = max(20, N + 4)
gaits = ("trot", "walk", "bounding", "pacing") if len(sys.argv) > 2 else ("trot",)
sb = synth.SyntheticBatch(B, N, N_gait=NG, gaits=gaits, n_seq=steps + 1, seed0=77000000)
eng = qrw_hip.Batch(B, n_steps=N, N_gait=NG, T_gait=0.02 * N)
ref = oracle.MPCBatch(B, 0.02, N, 0.02 * N, NG, fast=False)
wref = oracle.WbcBatch(B, 0.002, fast=False)
worst = worst_w = 0.0
bad = 0
for s in range(steps):
    d = sb.step(s)
    out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
    st = eng.mpc_stats()
    r = ref.run(s, d["xref"], d["fsteps"], 16)
    its = np.array([ref._lib.mpc_oracle_iter(ref._hs[b]) for b in range(B)])
    e = np.abs(out - r).reshape(B, -1).max(1) / np.maximum(np.abs(r).reshape(B, -1).max(1), 1e-9)
    worst = max(worst, e.max())
    if its is not None:
        bad += int((its != st["iters"]).sum())
    w = eng.wbc_compute_host(d["q"], d["dq"], np.ascontiguousarray(out[:, 12:, 0]), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
    wr = wref.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"], 16)
    tau_ref = wr[0] if isinstance(wr, (tuple, list)) else wr["tau_ff"]
    ew = np.abs(w["tau_ff"] - tau_ref).max() / max(np.abs(tau_ref).max(), 1e-9)
    worst_w = max(worst_w, ew)
    print("step %d: mpc worst rel err %.2e (max iters %d, statuses %s), torque rel err %.2e, iteration-count mismatches so far %d" %
          (s, e.max(), st["iters"].max(), np.unique(st["status"]).tolist(), ew, bad), flush=True)
print("SOAK", B, gaits, "worst mpc %.3e worst torque %.3e mismatches %d" % (worst, worst_w, bad))
