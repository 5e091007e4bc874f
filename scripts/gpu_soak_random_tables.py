"""Soak of the MPC kernels against the CPU oracle on ARBITRARY contact tables (synth.RandomContactTables; the suite's form is
tests/test_gpu_mpc_random_tables.py).  Per (N, B, K[, full]) block: K warm-started calls of B instances; every solve must take
the oracle's iteration count and status and match its result to 1e-4.  A mismatch is printed with the (seed0, instance, call)
that reproduces it.  usage: gpu_soak_random_tables.py [N:B:K[:full] | wbc:B:K ...]   (default: >= 50 000 MPC solves in total)
A `wbc:B:K` block soaks the whole-body step on synth.RandomWbcInputs the same way (QP iteration counts, torques, q_des / v_des, forces).
QRW_SOAK_SEED shifts every block's seeds (several runs = several independent samples)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle, qrw_hip, synth
oracle.build(fast=False)
blocks = sys.argv[1:] or ["16:4096:5", "12:2048:5", "5:2048:4", "16:1024:5:full", "24:1024:5", "32:1024:5", "32:256:5:full"]
threads = max(1, min(16, len(os.sched_getaffinity(0))))
total = bad_it = bad_st = bad_res = borderline = bad_loose = sens_flips = sens_total = 0
worst_all = 0.0
t_start = time.time()
seed_shift = int(os.environ.get("QRW_SOAK_SEED", "0")) * 1000003
wbc_total = wbc_bad = 0
wbc_worst = 0.0
for spec in blocks:
    p = spec.split(":")
    if p[0] == "wbc":
        B, K = int(p[1]), int(p[2])
        gen = synth.RandomWbcInputs(B, seed0=40000000 + seed_shift)
        eng = qrw_hip.Batch(B)
        oracle.build(fast=True)
        # two builds of the oracle (strict IEEE / -O3 -march=native): a robot whose two oracle results disagree (rho or any output
        # beyond 1e-9, or the iteration count) at this or an earlier call is ROUNDING-SENSITIVE -- OSQP's adaptive rho taken from a
        # residual that has converged to noise, tests/test_gpu_wbc.py::test_wbc_on_wild_inputs_matches_oracle -- and held to 1e-2 /
        # one termination check; every other robot to the iteration count and 1e-4
        ref, ref2 = oracle.WbcBatch(B, 0.002, fast=False), oracle.WbcBatch(B, 0.002, fast=True)
        sens = np.zeros(B, bool)

        def dev(x, y):
            # relative to the output's own scale, floored at 1 (N m, rad, rad/s, N): a robot in flight has contact forces of
            # ~1e-6 N (the QP's tolerance), whose relative deviation means nothing
            e = np.zeros(B)
            for u, v in zip(x, y):
                e = np.maximum(e, np.abs(u - v).reshape(B, -1).max(1) / np.maximum(np.abs(v).reshape(B, -1).max(1), 1.0))
            return np.where(np.isnan(e), np.inf, e)

        for c in range(K):
            d = gen.step(c)
            args = (d["q"], d["dq"], d["f_cmd"], d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
            o = eng.wbc_compute_host(*args)
            ra, rb = ref.compute(*args, threads), ref2.compute(*args, threads)
            (it, sto, rho), (it2, _, rho2) = ref.qp_stats(), ref2.qp_stats()
            st = eng.wbc_stats()
            sens |= (dev(rb, ra) > 1e-9) | (it != it2) | (np.abs(rho2 / rho - 1) > 1e-9)
            e = dev((o["tau_ff"], o["qdes"], o["vdes"], o["f_with_delta"]), ra)
            m = np.where(sens, (np.abs(st["iters"] - it) > 25) | (e >= 1e-2), (it != st["iters"]) | (e >= 1e-4)) | (st["status"] != sto)
            for b in np.nonzero(m)[0][:5]:
                print("MISMATCH wbc: seed0=%d instance %d call %d: hip iters %d status %d, oracle iters %d status %d, rel err %.3e, sensitive %s"
                      % (40000000 + seed_shift, b, c, st["iters"][b], st["status"][b], it[b], sto[b], e[b], sens[b]), flush=True)
            wbc_bad += int(m.sum()); wbc_total += B
            wbc_worst = max(wbc_worst, float(e[~sens].max()))
            print("WBC call %d: %d steps, %d robots rounding-sensitive so far; worst rel err %.2e on the others (iteration counts equal), %.2e on the sensitive ones; "
                  "QP iterations %d..%d; so far %d steps, %d mismatches" % (c, B, int(sens.sum()), e[~sens].max(), e[sens].max() if sens.any() else 0.0, it.min(), it.max(), wbc_total, wbc_bad), flush=True)
        continue
    N, B, K = int(p[0]), int(p[1]), int(p[2])
    full = len(p) > 3
    NG = N if full else max(20, N + 4)
    seed0 = 30000000 + 100000 * N + (50000 if full else 0) + seed_shift
    gen = synth.RandomContactTables(B, N, N_gait=NG, seed0=seed0)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=NG, T_gait=0.02 * N)
    ref = oracle.MPCBatch(B, 0.02, N, 0.02 * N, NG, fast=False)
    # the SHADOW: the same oracle fed the same inputs perturbed in the last bits (x (1 + 1e-13 u), u uniform in [-1, 1]).  OSQP's
    # adaptive rho is taken from a ratio of residual norms that are differences of O(10) terms; on about half a per cent of these
    # (hard) solves that amplifies 1e-13 into 1e-6 ... 1e-2 of rho, rho persists, and from then on iteration counts flip by a
    # check or two and results move at 1e-6 ... 1e-4 -- in the oracle against ITSELF (docs/HISTORY.md 8).  An instance is
    # ROUNDING-SENSITIVE from the first call at which shadow and oracle differ in rho by more than 1e-6 relative or in their
    # iteration count; "identical iteration counts" is asserted on all the others.
    shadow = oracle.MPCBatch(B, 0.02, N, 0.02 * N, NG, fast=False)
    prng = np.random.default_rng(seed0 ^ 0x5EED)
    sens = np.zeros(B, bool)
    worst, its_max, statuses, single = 0.0, 0, set(), 0
    parted = np.zeros(B, bool)  # instances whose solve ended one termination check apart from the oracle's on a threshold decision
    for c in range(K):
        d = gen.step(c)
        single += int((d["gait"].sum(2) == 1).sum())
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], c)
        st = eng.mpc_stats()
        r = ref.run(c, d["xref"], d["fsteps"], threads)
        it, stat = ref.iters()
        ratios = ref.check_ratios()
        shadow.run(c, d["xref"] * (1.0 + 1e-13 * prng.uniform(-1, 1, d["xref"].shape)), d["fsteps"], threads)
        rho_o = np.array([ref._lib.mpc_oracle_rho(h) for h in ref._hs])
        rho_s = np.array([shadow._lib.mpc_oracle_rho(h) for h in shadow._hs])
        sens |= (np.abs(rho_s / rho_o - 1) > 1e-6) | (shadow.iters()[0] != it)
        sx = np.maximum(np.abs(r[:, :12]).reshape(B, -1).max(1), 1e-12)
        sf = np.maximum(np.abs(r[:, 12:]).reshape(B, -1).max(1), 1e-12)
        e = np.maximum(np.abs(out[:, :12] - r[:, :12]).reshape(B, -1).max(1) / sx, np.abs(out[:, 12:] - r[:, 12:]).reshape(B, -1).max(1) / sf)
        e = np.where(np.isnan(e), np.inf, e)
        # OSQP ends a solve at the first check (every 25 iterations) at which BOTH residuals are below their tolerances.  A solve
        # whose deciding residual / tolerance is within 1e-3 of 1 at a check is decided by rounding (the residual is a difference
        # of O(10) terms at 1e-5: iterates that agree to 1e-10 give residuals that agree to ~1e-4): another arithmetic may end it
        # one check earlier or later.  Such an event is reported as BORDERLINE with its margin, not as a mismatch, and the
        # instance (whose warm start now differs) leaves the comparison for the rest of its block.
        early = (st["iters"] == it - 25) & (np.maximum(ratios[:, 2], ratios[:, 3]) >= 1.0) & (np.maximum(ratios[:, 2], ratios[:, 3]) < 1.0 + 1e-5)
        late = (st["iters"] == it + 25) & (np.maximum(ratios[:, 0], ratios[:, 1]) < 1.0) & (np.maximum(ratios[:, 0], ratios[:, 1]) > 1.0 - 1e-5)
        new_parted = (early | late) & ~parted
        for b in np.nonzero(new_parted)[0]:
            borderline += 1
            print("BORDERLINE: N=%d N_gait=%d seed0=%d instance %d call %d: hip %d iterations, oracle %d; the oracle's deciding residual / tolerance at "
                  "iteration %d was %.9f (rel err of the results %.2e); the instance leaves the comparison"
                  % (N, NG, seed0, b, c, st["iters"][b], it[b], min(st["iters"][b], it[b]),
                     max(ratios[b, 2], ratios[b, 3]) if early[b] else max(ratios[b, 0], ratios[b, 1]), e[b]), flush=True)
        parted |= new_parted
        # rounding-sensitive instances: the loose criterion (same status, results within 1e-3), reported in the summary.  Their
        # iteration counts are NOT bounded: OSQP applies a new rho only if the estimate is beyond 5x / a fifth of the current one, and
        # an instance whose estimate sits at that threshold takes 650 or 1650 iterations in the oracle ITSELF depending on the last
        # bits of its inputs (N = 32, seed0 42200027, instance 899, call 3: docs/HISTORY.md 8) -- the kernel took 1650.
        loose_bad = sens & ~parted & ((stat != st["status"]) | (e >= 1e-3))
        for b in np.nonzero(loose_bad)[0][:5]:
            print("MISMATCH (rounding-sensitive instance, loose criterion): N=%d seed0=%d instance %d call %d: hip iters %d status %d, oracle iters %d status %d, rel err %.3e"
                  % (N, seed0, b, c, st["iters"][b], st["status"][b], it[b], stat[b], e[b]), flush=True)
        bad_loose += int(loose_bad.sum())
        sens_flips += int((sens & ~parted & (st["iters"] != it)).sum())
        live = ~parted & ~sens
        for name, m in (("iterations", (it != st["iters"]) & live), ("status", (stat != st["status"]) & live), ("result", (e >= 1e-4) & live)):
            for b in np.nonzero(m)[0][:5]:
                print("MISMATCH %s: N=%d N_gait=%d seed0=%d instance %d call %d: hip iters %d status %d, oracle iters %d status %d, rel err %.3e, check ratios %s"
                      % (name, N, NG, seed0, b, c, st["iters"][b], st["status"][b], it[b], stat[b], e[b], ratios[b].tolist()), flush=True)
        bad_it += int(((it != st["iters"]) & live).sum()); bad_st += int(((stat != st["status"]) & live).sum()); bad_res += int(((e >= 1e-4) & live).sum())
        worst = max(worst, float(e[live].max())); its_max = max(its_max, int(it.max())); statuses |= set(stat.tolist())
        total += int(live.sum())
        print("N=%d%s call %d: %d solves compared, worst rel err %.2e, iterations %d..%d (mean %.0f), statuses %s; so far %d solves, mismatches it/status/result %d/%d/%d, borderline terminations %d, %.0f s"
              % (N, " full" if full else "", c, int(live.sum()), e[live].max(), it.min(), it.max(), it.mean(), sorted(set(stat.tolist())), total, bad_it, bad_st, bad_res, borderline, time.time() - t_start), flush=True)
    worst_all = max(worst_all, worst)
    sens_total += int(sens.sum())
    print("BLOCK N=%d: %d of %d instances rounding-sensitive by the end (shadow oracle on inputs perturbed by 1e-13)" % (N, int(sens.sum()), B), flush=True)
    print("BLOCK N=%d N_gait=%d B=%d K=%d seed0=%d: worst rel err %.3e, max iterations %d, statuses %s, single-stance rows %d"
          % (N, NG, B, K, seed0, worst, its_max, sorted(statuses), single), flush=True)
print("SOAK random contact tables: %d solves, worst rel err %.3e, mismatches: iterations %d, status %d, result (>= 1e-4) %d; "
      "borderline terminations (the oracle's deciding residual within 1e-5 of its tolerance, one check apart, reported above): %d; "
      "rounding-sensitive instances (oracle against itself on inputs perturbed by 1e-13: rho moves by > 1e-6) %d, on which the kernel's "
      "iteration count differed %d times and the loose criterion (status, results within 1e-3) failed %d times"
      % (total, worst_all, bad_it, bad_st, bad_res, borderline, sens_total, sens_flips, bad_loose))
if wbc_total:
    print("SOAK wild WBC inputs: %d whole-body steps, worst rel err %.3e on the robots that are not rounding-sensitive, mismatches (QP iterations / status / result) %d" % (wbc_total, wbc_worst, wbc_bad))
sys.exit(1 if (bad_it or bad_st or bad_res or bad_loose or wbc_bad) else 0)
