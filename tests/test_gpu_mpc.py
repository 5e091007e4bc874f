"""GPU parity tests of the MPC hot path: HIP (through the C ABI) vs the CPU oracle on identical seeded
inputs.  Tolerance: 1e-4 relative (BASELINE.json north_star); in practice the two agree to ~1e-12
because both run the same ADMM iterate sequence (identical iteration counts are asserted)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(a, ref):
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12)


def run_sequence(oracle_mod, synth_mod, B, N, gaits, steps, seed0, closed_loop=True, N_gait=None):
    import qrw_hip

    N_gait = N_gait or max(20, N + 4)
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, seed0=seed0)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    refs = [oracle_mod.MPC(0.02, N, 0.02 * N, N_gait) for _ in range(B)]
    x0 = None
    worst = 0.0
    for s in range(steps):
        d = sb.step(s, x0)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        st = eng.mpc_stats()
        ref = np.zeros_like(out)
        for b in range(B):
            assert refs[b].run(s, d["xref"][b], d["fsteps"][b]) == 0
            ref[b] = refs[b].get_latest_result()
            assert st["iters"][b] == refs[b].iter, (s, b, st["iters"][b], refs[b].iter)
            assert st["status"][b] == refs[b].status
            assert np.isclose(st["rho"][b], refs[b].rho, rtol=1e-9)
            e = max(rel_err(out[b, :12], ref[b, :12]), rel_err(out[b, 12:], ref[b, 12:]))
            worst = max(worst, e)
            assert e < RTOL, (s, b, e)
        if closed_loop:
            x0 = ref[:, :12, 0]
    return eng, refs, worst


def test_sweeps_selftest():
    import qrw_hip

    rc, err = qrw_hip.selftest_sweeps()
    assert rc == 0 and err < 1e-12


def test_trot_batch_matches_oracle(oracle_mod, synth_mod):
    eng, refs, worst = run_sequence(oracle_mod, synth_mod, 12, 16, ("trot",), 8, 20260000)
    assert worst < 1e-8  # far inside the 1e-4 bar
    # persisted warm-start state (scaled iterates) equals the oracle's workspace
    for b in (0, 7):
        stt = eng.mpc_state(b)
        x, z, y = refs[b].iterates()
        assert np.allclose(stt["x"], x, rtol=1e-6, atol=1e-9)
        assert np.allclose(stt["z"], z, rtol=1e-6, atol=1e-9)
        assert np.allclose(stt["y"], y, rtol=1e-6, atol=1e-9)
        gait, S = eng.mpc_gait(b)
        assert np.array_equal(gait, refs[b].get_gait()) and np.array_equal(S, refs[b].get_Sgait())


def test_mixed_gaits_match_oracle(oracle_mod, synth_mod):
    run_sequence(oracle_mod, synth_mod, 10, 16, ("walk", "trot", "bounding", "pacing", "static"), 6, 31000)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 8, 12, 15])
def test_short_horizons(oracle_mod, synth_mod, N):
    run_sequence(oracle_mod, synth_mod, 3, N, ("trot",), 4, 41000 + N)


@pytest.mark.parametrize("N", [17, 24, 31, 32])
def test_long_horizons_two_wavefronts(oracle_mod, synth_mod, N):
    """N > 16 runs two wavefronts per instance (BASELINE config 4: N = 32, mixed walk/trot/bound schedules)."""
    eng, refs, worst = run_sequence(oracle_mod, synth_mod, 5, N, ("walk", "trot", "bounding"), 5, 43000 + N)
    stt = eng.mpc_state(3)
    x, z, y = refs[3].iterates()
    assert np.allclose(stt["x"], x, rtol=1e-6, atol=1e-9) and np.allclose(stt["y"], y, rtol=1e-6, atol=1e-9)
    gait, S = eng.mpc_gait(3)
    assert np.array_equal(gait, refs[3].get_gait()) and np.array_equal(S, refs[3].get_Sgait())


def test_open_loop_noisy_states(oracle_mod, synth_mod):
    run_sequence(oracle_mod, synth_mod, 6, 16, ("trot", "walk"), 6, 51000, closed_loop=False)


def test_fourstance_immobile_properties():
    """scripts/test_mpc.py:54-85 on the HIP path: equal forces, sum f_z = m g, state -> reference."""
    import qrw_hip

    N = 16
    eng = qrw_hip.Batch(1, N)
    xref = np.zeros((1, 12, N + 1))
    xref[0, 2, :] = 0.24474949993103629
    fsteps = np.zeros((1, 20, 12))
    fsteps[0, :N, :] = [0.195, 0.147, 0., 0.195, -0.147, 0., -0.195, 0.147, 0., -0.195, -0.147, 0.]
    for i in range(60):
        r = eng.mpc_solve_host(xref, fsteps, i)[0]
        xref[0, :, 0] = r[:12, 0]
    assert np.allclose(r[12:, 0], np.tile(r[12:15, 0], 4), atol=1e-8)
    assert np.allclose(r[:12, 0], xref[0, :, 1], atol=1e-3)
    assert abs(r[14::3, 0].sum() - 9.81 * 2.50000279) < 1e-3


def test_batch_position_and_size_invariance(synth_mod):
    """An instance's result does not depend on where it sits in the batch or on the batch size."""
    import qrw_hip

    N = 16
    big = synth_mod.SyntheticBatch(37, N, gaits=("trot", "walk"), seed0=61000)
    eng_big = qrw_hip.Batch(37, N)
    sub = [5, 36, 0, 17]
    eng_sub = qrw_hip.Batch(len(sub), N)
    for s in range(3):
        d = big.step(s)
        a = eng_big.mpc_solve_host(d["xref"], d["fsteps"], s)
        b = eng_sub.mpc_solve_host(d["xref"][sub], d["fsteps"][sub], s)
        assert np.array_equal(a[sub], b)


def test_not_setup_and_resetup(oracle_mod, synth_mod):
    import qrw_hip

    N = 16
    sb = synth_mod.SyntheticBatch(2, N, seed0=71000)
    eng = qrw_hip.Batch(2, N)
    d = sb.step(0)
    out = eng.mpc_solve_host(d["xref"], d["fsteps"], 5)  # never set up: the reference would crash
    assert np.isnan(out).all() and (eng.mpc_stats()["status"] == -100).all()
    # per-instance num_iter: instance 0 sets up, instance 1 still not
    out = eng.mpc_solve_host(d["xref"], d["fsteps"], np.array([0, 3], np.int32))
    st = eng.mpc_stats()
    assert st["status"][0] == 1 and st["status"][1] == -100 and np.isnan(out[1]).all() and np.isfinite(out[0]).all()
    # a later num_iter == 0 re-creates the problem: same answer as a fresh object
    eng.mpc_solve_host(d["xref"], d["fsteps"], 0)
    d1 = sb.step(1)
    eng.mpc_solve_host(d1["xref"], d1["fsteps"], 1)
    again = eng.mpc_solve_host(d["xref"], d["fsteps"], 0)
    fresh = qrw_hip.Batch(2, N).mpc_solve_host(d["xref"], d["fsteps"], 0)
    assert np.array_equal(again, fresh)
    ref = oracle_mod.MPC(0.02, N, 0.32, 20)
    ref.run(0, d["xref"][0], d["fsteps"][0])
    assert rel_err(again[0], ref.get_latest_result()) < RTOL


def test_device_api_matches_host_api(synth_mod):
    import torch

    import qrw_hip

    N, B = 16, 9
    sb = synth_mod.SyntheticBatch(B, N, seed0=81000)
    e1, e2 = qrw_hip.Batch(B, N), qrw_hip.Batch(B, N)
    for s in range(3):
        d = sb.step(s)
        a = e1.mpc_solve_host(d["xref"], d["fsteps"], s)
        out = e2.mpc_solve(torch.from_numpy(d["xref"]).cuda(), torch.from_numpy(d["fsteps"]).cuda(), s)
        torch.cuda.synchronize()
        assert np.array_equal(a, out.cpu().numpy())
    ni = torch.full((B,), 3, dtype=torch.int32, device="cuda")
    d = sb.step(3)
    out = e2.mpc_solve(torch.from_numpy(d["xref"]).cuda(), torch.from_numpy(d["fsteps"]).cuda(), ni)
    assert np.array_equal(e1.mpc_solve_host(d["xref"], d["fsteps"], 3), out.cpu().numpy())


def test_full_size_batch_properties(synth_mod):
    """BASELINE config sizes (batch 4096): size-independent properties instead of an oracle run —
    every instance solved, dynamics rows satisfied, friction cone and unilaterality respected,
    swing feet unloaded, and a random subset checked against an independent small-batch run."""
    import qrw_hip

    B, N = 4096, 16
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot",), seed0=20260000)
    eng = qrw_hip.Batch(B, N)
    for s in range(3):
        d = sb.step(s)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
    st = eng.mpc_stats()
    assert (st["status"] == 1).all() and (st["iters"] % 25 == 0).all()
    f = out[:, 12:, :].transpose(0, 2, 1).reshape(B, N, 4, 3)
    gait = d["gait"][:, :N]
    assert np.abs(f[gait == 0]).max() < 1e-3
    fs = f[gait == 1]
    mu = np.float64(np.float32(0.9))
    assert (fs[:, 2] > -1e-3).all() and (fs[:, 2] < 25 + 1e-3).all()
    assert (np.abs(fs[:, 0]) <= mu * fs[:, 2] + 1e-3).all() and (np.abs(fs[:, 1]) <= mu * fs[:, 2] + 1e-3).all()
    # discrete dynamics of the predicted trajectory, linear velocity rows: v_{k+1} = v_k + dt (sum f / m - g)
    mass, dt = np.float64(np.float32(2.50000279)), 0.02
    v = np.concatenate([d["xref"][:, 6:9, 0:1], out[:, 6:9, :]], axis=2)
    acc = f.sum(axis=2).transpose(0, 2, 1) / mass
    acc[:, 2, :] -= np.float64(np.float32(9.81))
    assert np.abs(v[:, :, 1:] - v[:, :, :-1] - dt * acc).max() < 1e-4
    idx = np.array([0, 1, 777, 2048, 4095])
    small = qrw_hip.Batch(len(idx), N)
    for s in range(3):
        d2 = sb.step(s)
        o2 = small.mpc_solve_host(d2["xref"][idx], d2["fsteps"][idx], s)
    assert np.array_equal(o2, out[idx])


@pytest.mark.parametrize("B", [1025, 4096, 5000])
def test_block_order_is_a_permutation_sorted_by_the_moving_average(synth_mod, B):
    """The longest-first block order the next launch uses (mpc_order_kernel, one wavefront): a permutation of 0..B-1
    whose keys (moving average of the iteration counts, in bins of 25) never increase; none for batches <= 1024.
    An order entry out of range would send a workgroup to another instance's memory, so this is checked directly."""
    import qrw_hip

    N = 16
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=20261100)
    eng = qrw_hip.Batch(B, N)
    assert eng.mpc_order() is None
    ema_ref = np.zeros(B, np.float32)
    for s in range(3):
        d = sb.step(s)
        eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        it = eng.mpc_stats()["iters"].astype(np.float32)
        ema_ref = np.where(ema_ref == 0, it, ema_ref + (it - ema_ref) * np.float32(0.125)).astype(np.float32)
        order, ema = eng.mpc_order()
        assert np.array_equal(np.sort(order), np.arange(B, dtype=np.int32))
        assert np.allclose(ema, ema_ref, rtol=1e-6)
        key = np.clip((ema * np.float32(0.04)).astype(np.int32), 0, 160)
        assert (np.diff(key[order]) <= 0).all()
    small = qrw_hip.Batch(64, N)
    d = synth_mod.SyntheticBatch(64, N).step(0)
    small.mpc_solve_host(d["xref"], d["fsteps"], 0)
    assert small.mpc_order() is None


@pytest.mark.parametrize("N", [16, 12, 32])
def test_nan_input_poisons_one_instance_only(oracle_mod, synth_mod, N):
    """A NaN in one instance's reference trajectory: OSQP's residual norms (maxima of |.|) drop NaN, so the solve ends as
    "solved" at its first termination check with a NaN result and keeps its NaN iterate as the next call's warm start
    (store_solution only cold-starts on infeasible / non-convex statuses) -- reference behaviour, kept: the controller's fourth
    error code is what stops such a robot (controller_glue.h).  The other instances of the batch must not notice.  GPU and
    oracle must agree on iterations, status and NaN pattern, at every kind of horizon (compile-time, runtime, two wavefronts)."""
    import qrw_hip

    B, N_gait = 4, max(20, N)
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=("trot",), seed0=20260500)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    refs = [oracle_mod.MPC(0.02, N, 0.02 * N, N_gait) for _ in range(B)]
    for s in range(3):
        d = sb.step(s)
        xref = d["xref"].copy()
        if s == 1:
            xref[1, 7, 3] = np.nan  # a velocity entry of the bounds of instance 1
            xref[3, 2, 5] = 1e300   # instance 3: overflow (Inf - Inf in the residuals): whatever OSQP's arithmetic does with it
        out = eng.mpc_solve_host(xref, d["fsteps"], s)
        st = eng.mpc_stats()
        for b in range(B):
            assert refs[b].run(s, xref[b], d["fsteps"][b]) == 0
            ref = refs[b].get_latest_result()
            assert st["iters"][b] == refs[b].iter and st["status"][b] == refs[b].status, (s, b, st["iters"][b], refs[b].iter,
                                                                                         st["status"][b], refs[b].status)
            assert np.array_equal(np.isnan(out[b]), np.isnan(ref)), (s, b)
            ok = ~np.isnan(ref)
            if ok.any():
                assert np.abs(out[b][ok] - ref[ok]).max() <= RTOL * max(np.abs(ref[ok]).max(), 1e-12), (s, b)
        if s >= 1:
            assert np.isnan(out[1]).any() and np.isfinite(out[0]).all() and np.isfinite(out[2]).all()



def test_truncated_gait_keeps_stale_rows(oracle_mod, synth_mod):
    """fsteps with an all-zero row before the horizon ends: construct_gait stops there (MPC.cpp:686-701), update_ML only
    rewrites the B blocks and S flags of the rows before it and the later ones keep what earlier calls left
    (MPC.cpp:418-464).  The cut moves from call to call, including to row 0 (nothing rewritten) and back to full length."""
    import qrw_hip

    B, N, N_gait = 6, 16, 20
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=("trot", "walk"), seed0=20260700)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    refs = [oracle_mod.MPC(0.02, N, 0.02 * N, N_gait) for _ in range(B)]
    cuts = [[16, 16, 16, 16, 16, 16], [9, 16, 3, 12, 1, 16], [16, 5, 0, 12, 7, 15], [4, 16, 16, 0, 16, 2], [16, 16, 16, 16, 16, 16]]
    for s, cut in enumerate(cuts):
        d = sb.step(s)
        fsteps = d["fsteps"].copy()
        for b in range(B):
            fsteps[b, cut[b]:, :] = 0.0
        out = eng.mpc_solve_host(d["xref"], fsteps, s)
        st = eng.mpc_stats()
        for b in range(B):
            assert refs[b].run(s, d["xref"][b], fsteps[b]) == 0
            ref = refs[b].get_latest_result()
            assert st["iters"][b] == refs[b].iter and st["status"][b] == refs[b].status, (s, b, st["iters"][b], refs[b].iter)
            assert max(rel_err(out[b, :12], ref[:12]), rel_err(out[b, 12:], ref[12:])) < RTOL, (s, b)
            gait, S = eng.mpc_gait(b)
            assert np.array_equal(gait, refs[b].get_gait()), (s, b)


def test_reference_known_answers_on_the_hip_path(oracle_mod):
    """The reference's own scenarios (scripts/test_mpc.py:87-110 four-stance non-centred, :136-160 two-stance trot
    centred; tests/trot_kat.py) driven through the HIP path as a batch of two, each instance with its own feedback
    loop, 500 receding-horizon calls.  The reference's criteria must hold, and every call must take the same number
    of ADMM iterations as the oracle fed the same inputs (the oracle is stepped on the HIP path's trajectory)."""
    import qrw_hip
    import trot_kat

    N, NG = trot_kat.N, trot_kat.N_GAIT
    eng = qrw_hip.Batch(2, n_steps=N, N_gait=NG, T_gait=0.32)
    refs = [oracle_mod.MPC(0.02, N, 0.32, NG) for _ in range(2)]
    xref = np.zeros((2, 12, N + 1))
    xref[:, 2, :] = trot_kat.H_REF
    xref[0, :, 0] = trot_kat.NOT_CENTERED  # instance 0: four-stance, non-centred
    four = np.zeros((NG, 12))
    four[:N, :] = [0.195, 0.147, 0., 0.195, -0.147, 0., -0.195, 0.147, 0., -0.195, -0.147, 0.]
    plan = trot_kat.CompressedTrot()  # instance 1: two-stance trot, centred
    worst = 0.0
    for i in range(500):
        fsteps = np.stack([four, plan.fsteps()])
        out = eng.mpc_solve_host(xref, fsteps, i)
        st = eng.mpc_stats()
        assert (st["status"] == 1).all()
        for b in range(2):
            assert refs[b].run(i, xref[b], fsteps[b]) == 0
            assert st["iters"][b] == refs[b].iter, (i, b, st["iters"][b], refs[b].iter)
            r = refs[b].get_latest_result()
            worst = max(worst, rel_err(out[b, :12], r[:12]), rel_err(out[b, 12:], r[12:]))
        plan.roll()
        xref[0, :, 0] = out[0, :12, 0]
        if i > 0:
            xref[1, :, 0] = out[1, :12, 0]
    assert worst < RTOL
    r = out[0]
    assert np.allclose(r[12:, 0], np.tile(r[12:15, 0], 4)) and np.allclose(r[:12, 0], xref[0, :, 1], atol=1e-3)
    assert np.allclose(out[1, :12, 0], xref[1, :, 1], atol=1e-2)


def test_config2_batch256_mpc_only(oracle_mod, synth_mod):
    """BASELINE config 2 (batch 256, N = 16, MPC only) at its full size: properties on every instance, oracle parity
    (iterations, status, rho, 1e-4) on a spread of 24 of them over 4 receding-horizon calls."""
    import qrw_hip

    B, N = 256, 16
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot",), seed0=20260000)
    eng = qrw_hip.Batch(B, N)
    idx = np.arange(0, B, 11)[:24]
    refs = {int(b): oracle_mod.MPC(0.02, N, 0.32, 20) for b in idx}
    x0 = None
    for s in range(4):
        d = sb.step(s, x0)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        st = eng.mpc_stats()
        assert (st["status"] == 1).all() and (st["iters"] % 25 == 0).all()
        for b, m in refs.items():
            assert m.run(s, d["xref"][b], d["fsteps"][b]) == 0
            r = m.get_latest_result()
            assert st["iters"][b] == m.iter and np.isclose(st["rho"][b], m.rho, rtol=1e-9), (s, b)
            assert max(rel_err(out[b, :12], r[:12]), rel_err(out[b, 12:], r[12:])) < RTOL
        x0 = out[:, :12, 0]
    f = out[:, 12:, :].transpose(0, 2, 1).reshape(B, N, 4, 3)
    gait = d["gait"][:, :N]
    assert np.abs(f[gait == 0]).max() < 1e-3 and (f[gait == 1][:, 2] > -1e-3).all()


@pytest.mark.parametrize("N,gaits,B", [(16, ("trot",), 1024), (16, ("walk", "trot", "bounding", "pacing"), 512),
                                       (32, ("walk", "trot", "bounding"), 256)])
def test_wide_oracle_parity_sample(oracle_mod, synth_mod, N, gaits, B):
    """Oracle parity on a wide sample in the driver-run suite (the oracle stepped with one instance per host thread):
    EVERY instance of every call must take the oracle's iteration count and status and match it to 1e-4 (measured
    ~1e-10).  Open-loop noisy states, 6 calls: warm start, adaptive rho and (N = 32) max-iter exits are all exercised."""
    import qrw_hip

    N_gait = max(20, N + 4)
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, seed0=20270000 + N)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    ref = oracle_mod.MPCBatch(B, 0.02, N, 0.02 * N, N_gait, fast=False)  # the strict checker build
    threads = max(1, min(16, len(__import__("os").sched_getaffinity(0))))
    worst = 0.0
    seen_status = set()
    for s in range(6):
        d = sb.step(s)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        r = ref.run(s, d["xref"], d["fsteps"], threads)
        it, st = ref.iters()
        g = eng.mpc_stats()
        assert np.array_equal(g["iters"], it), (s, np.nonzero(g["iters"] != it)[0][:8])
        assert np.array_equal(g["status"], st), s
        seen_status |= set(st.tolist())
        scale_x = np.abs(r[:, :12]).max(axis=(1, 2), keepdims=True)
        scale_f = np.maximum(np.abs(r[:, 12:]).max(axis=(1, 2), keepdims=True), 1e-12)
        e = max((np.abs(out[:, :12] - r[:, :12]) / scale_x).max(), (np.abs(out[:, 12:] - r[:, 12:]) / scale_f).max())
        worst = max(worst, e)
        assert e < RTOL, (s, e)
    assert 1 in seen_status
    print("wide parity N=%d B=%d: worst relative deviation %.2e, statuses %s" % (N, B, worst, sorted(seen_status)))


@pytest.mark.parametrize("N,B,K,gaits", [(16, 8, 5, ("trot",)), (16, 1500, 6, ("trot", "walk")), (12, 40, 4, ("trot",)),
                                         (32, 24, 4, ("walk", "trot", "bounding")), (24, 10, 3, ("trot",))])
def test_sequence_launch_equals_consecutive_calls(synth_mod, N, B, K, gaits):
    """qrw_mpc_solve_sequence (one workgroup per task, per-instance task queues, no device-wide barrier between the calls)
    must give what K calls of qrw_mpc_solve give, bit for bit: every call's result, iteration counts, statuses of the last
    call and the warm-start state it leaves behind (checked through one more ordinary call on both handles)."""
    import torch

    import qrw_hip

    N_gait = max(20, N + 4)
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, seed0=20280000 + N)
    steps = [sb.step(s) for s in range(K + 1)]
    a, b = (qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N) for _ in range(2))
    dev = torch.device("cuda", 0)
    xs = torch.from_numpy(np.stack([st["xref"] for st in steps[:K]])).to(dev)
    fs = torch.from_numpy(np.stack([st["fsteps"] for st in steps[:K]])).to(dev)
    ref_out, ref_it = [], []
    for s in range(K):
        ref_out.append(a.mpc_solve(xs[s], fs[s], s).cpu().numpy())
        ref_it.append(a.mpc_stats()["iters"].copy())
    its = torch.zeros((K, B), dtype=torch.int32, device=dev)
    out = b.mpc_solve_sequence(xs, fs, 0, iters=its)
    torch.cuda.synchronize()
    assert not b.mpc_sequence_timed_out()
    assert np.array_equal(its.cpu().numpy(), np.stack(ref_it))
    assert np.array_equal(out.cpu().numpy(), np.stack(ref_out))
    sa, sb_ = a.mpc_stats(), b.mpc_stats()
    for key in ("iters", "status"):
        assert np.array_equal(sa[key], sb_[key]), key
    assert np.array_equal(sa["rho"], sb_["rho"])
    # the state left behind: one more ordinary call on both
    x1 = torch.from_numpy(steps[K]["xref"]).to(dev)
    f1 = torch.from_numpy(steps[K]["fsteps"]).to(dev)
    assert np.array_equal(a.mpc_solve(x1, f1, K).cpu().numpy(), b.mpc_solve(x1, f1, K).cpu().numpy())
    # a second sequence continuing from there (first_num_iter > 0) against ordinary calls
    o2 = b.mpc_solve_sequence(xs[:2].contiguous(), fs[:2].contiguous(), K + 1)
    r2 = [a.mpc_solve(xs[s], fs[s], K + 1 + s).cpu().numpy() for s in range(2)]
    torch.cuda.synchronize()
    assert np.array_equal(o2.cpu().numpy(), np.stack(r2)) and not b.mpc_sequence_timed_out()


def test_sequence_tail_longer_than_the_give_up_clock(synth_mod):
    """the sequence kernel's 2 s give-up clock must count from the last observed progress, not from the
    moment a workgroup starts to look for work.  ONE N = 32 instance, K = 640 calls: the launch has 640 workgroups of which
    512 are resident at once; the first takes call 0, the other 511 reserve queue slots and wait for them -- the last of
    them for ~510 solves of ~5 ms, longer than 2 s.  No workgroup may give up, every call must have run (no NaN pre-fill
    left, no -1 iteration count), and the results must be those of consecutive calls."""
    import time

    import torch

    import qrw_hip

    N, B, K = 32, 1, 640
    Ng = 36
    sb = synth_mod.SyntheticBatch(B, N, N_gait=Ng, gaits=("walk",), seed0=20290000)
    d = [sb.step(s) for s in range(8)]
    dev = torch.device("cuda", 0)
    xs = torch.from_numpy(np.stack([d[s % 8]["xref"] for s in range(K)])).to(dev)
    fs = torch.from_numpy(np.stack([d[s % 8]["fsteps"] for s in range(K)])).to(dev)
    a, b = (qrw_hip.Batch(B, n_steps=N, N_gait=Ng, T_gait=0.02 * N) for _ in range(2))
    its = torch.zeros((K, B), dtype=torch.int32, device=dev)
    t0 = time.perf_counter()
    out = b.mpc_solve_sequence(xs, fs, 0, iters=its)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert not b.mpc_sequence_timed_out()
    it = its.cpu().numpy()
    o = out.cpu().numpy()
    assert (it > 0).all() and np.isfinite(o).all()
    print("sequence of %d calls of one N = 32 instance: %.2f s, %.0f iterations per call" % (K, el, it.mean()))
    assert el > 2.2, "the chain was meant to outlast the 2 s clock (%.2f s): raise K" % el
    for s in range(16):  # a prefix against ordinary calls (same code path as the parity test of the sequence launch)
        r = a.mpc_solve(xs[s], fs[s], s).cpu().numpy()
        assert np.array_equal(r, o[s]), s
        assert a.mpc_stats()["iters"][0] == it[s, 0]


def test_sequence_on_a_masked_stream_beside_another_stream_group(synth_mod):
    """seq_groups is sized from the whole device, but a sequence may run on a CU-masked stream or beside
    another stream's solves, where fewer workgroups are resident than dealt call-0 tasks.  Results must not change."""
    import torch

    import qrw_hip

    N, B, K = 16, 700, 4
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=20291000)
    steps = [sb.step(s) for s in range(K)]
    dev = torch.device("cuda", 0)
    xs = torch.from_numpy(np.stack([st["xref"] for st in steps])).to(dev)
    fs = torch.from_numpy(np.stack([st["fsteps"] for st in steps])).to(dev)
    a, b, c = (qrw_hip.Batch(B, n_steps=N) for _ in range(3))
    ref = np.stack([a.mpc_solve(xs[s], fs[s], s).cpu().numpy() for s in range(K)])
    n_cu = qrw_hip.device_cu_count(0)
    masked = qrw_hip.CuStream(0, 0, 64)          # 64 of the compute units: 256 resident workgroups for 700 dealt tasks
    other = qrw_hip.CuStream(0, 64, n_cu - 64)
    torch.cuda.synchronize()
    with torch.cuda.stream(other.torch):          # a second group's solves in flight on the other units
        for s in range(K):
            c.mpc_solve(xs[s], fs[s], s)
    with torch.cuda.stream(masked.torch):
        out = b.mpc_solve_sequence(xs, fs, 0)
    torch.cuda.synchronize()
    assert not b.mpc_sequence_timed_out()
    assert np.array_equal(out.cpu().numpy(), ref)
    masked.close()
    other.close()


@pytest.mark.parametrize("N,B,chunk,gaits,levels,lbin", [
    (32, 1300, 600, ("walk", "trot", "bounding"), 9, 400), (32, 700, 200, ("trot", "bounding"), 9, 400),
    (24, 900, 400, ("trot", "walk"), 9, 400), (32, 900, 400, ("walk", "trot", "bounding"), 1, 400),
    (32, 900, 400, ("walk", "trot", "bounding"), 4, 100)])
def test_time_sliced_launch_is_independent_of_the_slicing(synth_mod, N, B, chunk, gaits, levels, lbin, monkeypatch):
    """qrw_mpc_solve at N > 16 with more instances than resident slots time-slices the solves round robin inside the
    launch (slices of `chunk` iterations, parked solves resumed by later workgroups, mpc_kernel.hip PRE).  A resumed solve
    must be the uninterrupted one BIT FOR BIT: results, iteration counts, status, rho and the warm-start state left behind
    (checked through the following calls) are compared between slices of `chunk` iterations and slices of 3800 (the same
    kernel, where practically nothing is ever parked).  Against a handle with the slicing switched off -- a different
    instantiation of the kernel, whose floating-point contraction the compiler chooses on its own -- iteration counts and
    status must be identical and the results equal to rounding (measured: identical at N = 32, <= 1e-9 at N = 24).
    Which parked solve a free workgroup takes next -- one FIFO (levels = 1) or the priority levels by predicted remaining
    iterations (the default: 9 levels, bins of 200 iterations; 4 levels of 100 here as a third shape) -- must not show in any word."""
    import torch

    import qrw_hip

    N_gait = max(20, N + 4)
    monkeypatch.setenv("QRW_PREEMPT_LEVELS", str(levels))
    monkeypatch.setenv("QRW_PREEMPT_BIN", str(lbin))
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, seed0=20300000 + N)
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", "0")
    plain = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    monkeypatch.setenv("QRW_PREEMPT_MIN_BATCH", "8")
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", "3800")
    whole = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", str(chunk))
    sliced = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    dev = torch.device("cuda", 0)
    many = 0
    for s in range(4):
        d = sb.step(s)
        x, f = torch.from_numpy(d["xref"]).to(dev), torch.from_numpy(d["fsteps"]).to(dev)
        a = plain.mpc_solve(x, f, s).cpu().numpy()
        w = whole.mpc_solve(x, f, s).cpu().numpy()
        b = sliced.mpc_solve(x, f, s).cpu().numpy()
        sa, sw, sb_ = plain.mpc_stats(), whole.mpc_stats(), sliced.mpc_stats()
        for key in ("iters", "status", "rho", "pri_res", "dua_res"):
            assert np.array_equal(sw[key], sb_[key]), (s, key, np.nonzero(sw[key] != sb_[key])[0][:8])
        assert np.array_equal(w, b, equal_nan=True), s
        assert np.array_equal(sa["iters"], sb_["iters"]) and np.array_equal(sa["status"], sb_["status"]), s
        assert np.allclose(sa["rho"], sb_["rho"], rtol=1e-7) and rel_err(b, a) < 1e-8, (s, rel_err(b, a))
        many += int((sb_["iters"] > 2 * chunk).sum())
    assert many > 0, "no solve needed more than two slices: the test does not exercise a resumed solve twice"


def test_time_sliced_launch_bookkeeping(synth_mod, monkeypatch):
    """The queue's own accounts after a time-sliced launch (qrw_mpc_get_slice_stats): every instance counted as finished; a
    solve is parked at the end of a slice only when a taker workgroup is granted for it (otherwise it goes on in place: nobody
    waits), so there are AT MOST ceil(iterations / slice) - 1 parks per solve -- and most of them while the batch outnumbers
    the resident slots --, at most one first park per solve longer than a slice (level 0), later ones in the
    predicted-remainder levels -- more than one of them in use --, and EXACTLY as many takers that took a solve as parks
    (every parked solve found its taker, once)."""
    import torch

    import qrw_hip

    N, B, chunk, N_gait = 32, 1100, 400, 36
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", str(chunk))
    monkeypatch.setenv("QRW_PREEMPT_MIN_BATCH", "8")
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=("walk", "trot", "bounding"), seed0=20330000)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    dev = torch.device("cuda", 0)
    for s in range(3):
        d = sb.step(s)
        eng.mpc_solve(torch.from_numpy(d["xref"]).to(dev), torch.from_numpy(d["fsteps"]).to(dev), s)
        it = eng.mpc_stats()["iters"].astype(np.int64)
        st = eng.mpc_slice_stats()
        parks = st["parks_per_level"].astype(np.int64)
        assert st["levels"] == 9 and st["chunk"] == chunk and st["finished"] == B
        slices = -(-it // chunk)
        print("step %d: parks per level %s of at most %d, takers %d" % (s, parks.tolist(), (slices - 1).sum(), st["takers"]))
        assert 0.5 * (slices - 1).sum() <= parks.sum() <= (slices - 1).sum(), (s, parks, (slices - 1).sum())
        assert 0.5 * (it > chunk).sum() <= parks[0] <= (it > chunk).sum(), (s, parks[0], (it > chunk).sum())
        assert (parks[1:] > 0).sum() >= 3, parks
        assert st["takers"] == parks.sum()


def test_time_sliced_launch_matches_the_oracle(oracle_mod, synth_mod, monkeypatch):
    """The time-sliced launch straight against the CPU oracle (slices of 200 iterations forced on a small batch): every
    instance takes the oracle's iteration count and status, results within 1e-4."""
    import qrw_hip

    N, B, N_gait = 32, 48, 36
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", "200")
    monkeypatch.setenv("QRW_PREEMPT_MIN_BATCH", "1")
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=("walk", "trot", "bounding"), seed0=20310000)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    ref = oracle_mod.MPCBatch(B, 0.02, N, 0.02 * N, N_gait, fast=False)
    threads = max(1, min(16, len(__import__("os").sched_getaffinity(0))))
    for s in range(3):
        d = sb.step(s)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        r = ref.run(s, d["xref"], d["fsteps"], threads)
        it, st = ref.iters()
        g = eng.mpc_stats()
        assert np.array_equal(g["iters"], it) and np.array_equal(g["status"], st), s
        assert rel_err(out[:, :12], r[:, :12]) < RTOL and rel_err(out[:, 12:], r[:, 12:]) < RTOL, s


def test_twostance_closed_loop_grows_at_the_monodromy_rate_on_the_hip_path():
    """The red reference-held pin (scripts/test_mpc.py:162-190), HIP side of tests/test_oracle_mpc.py::
    test_twostance_monodromy_explains_the_red_known_answer: from a start 1 % of the way to the reference's non-centred
    state the HIP path's closed loop (state := first predicted state) deviates from the reference at the rate the one-period
    map of the unconstrained MPC law predicts for the weights of src/MPC.cpp:330 (spectral radius 1.20 per gait period,
    tests/monodromy.py: independent assembly + dense KKT solves) -- the scenario is unstable for the QP as written, on
    the CPU oracle and on the GPU alike."""
    import monodromy as mono
    import qrw_hip
    import trot_kat

    N = trot_kat.N
    rho = mono.spectral_radius(mono.W_330)
    centre = np.zeros(12)
    centre[2] = trot_kat.H_REF
    start = centre + 0.01 * (trot_kat.NOT_CENTERED - centre)
    eng = qrw_hip.Batch(1, n_steps=N, N_gait=trot_kat.N_GAIT)
    err = []

    def solve(i, xref, fsteps):
        r = eng.mpc_solve_host(xref[None], fsteps[None], i)[0]
        assert eng.mpc_stats()["status"][0] == 1
        err.append(np.abs(r[:12, 0] - xref[:, 1]).max())
        return r

    old = trot_kat.NOT_CENTERED
    trot_kat.NOT_CENTERED = start
    try:
        trot_kat.run_twostance(solve, 161, centered=False)
    finally:
        trot_kat.NOT_CENTERED = old
    growth = err[160] / err[96]
    assert 1.17 < rho < 1.23 and 0.85 * rho ** 4 < growth < 1.15 * rho ** 4, (rho, growth)


def test_time_sliced_launch_edge_cases(synth_mod, monkeypatch):
    """The time-sliced launch where its bookkeeping could go wrong: instances that are not set up (num_iter != 0 on their
    first call: NaN result, status NOT_SETUP, counted as finished), per-instance iteration arguments, a compute-unit-masked
    stream with fewer resident slots than instances (every later workgroup of the grid waits on the queue), and a second
    stream's solves beside it.  Everything must equal the unsliced launch (iteration counts / status exactly, results to
    rounding), every call must complete (no stale numbers: the result buffer is pre-filled with NaN)."""
    import torch

    import qrw_hip

    N, B, N_gait = 32, 600, 36
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=("trot", "walk", "bounding"), seed0=20320000)
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", "0")
    plain = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", "400")
    monkeypatch.setenv("QRW_PREEMPT_MIN_BATCH", "8")
    sliced, other = (qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N) for _ in range(2))
    dev = torch.device("cuda", 0)
    n_cu = qrw_hip.device_cu_count(0)
    masked, rest = qrw_hip.CuStream(0, 0, 96), qrw_hip.CuStream(0, 96, n_cu - 96)  # 192 resident slots for 600 instances
    for s in range(3):
        d = sb.step(s)
        x, f = torch.from_numpy(d["xref"]).to(dev), torch.from_numpy(d["fsteps"]).to(dev)
        ni = torch.full((B,), s, dtype=torch.int32, device=dev)
        if s == 0:
            ni[5::97] = 3  # these instances see num_iter != 0 before any set-up call
        a = plain.mpc_solve(x, f, ni).cpu().numpy()
        torch.cuda.synchronize()
        with torch.cuda.stream(rest.torch):
            other.mpc_solve(x, f, ni)
        with torch.cuda.stream(masked.torch):
            b = sliced.mpc_solve(x, f, ni)
        torch.cuda.synchronize()
        b = b.cpu().numpy()
        sa, sb_ = plain.mpc_stats(), sliced.mpc_stats()
        assert np.array_equal(sa["iters"], sb_["iters"]) and np.array_equal(sa["status"], sb_["status"]), s
        assert np.array_equal(np.isnan(a), np.isnan(b)), s
        ok = ~np.isnan(a)
        assert np.abs(a[ok] - b[ok]).max() <= 1e-8 * np.abs(a[ok]).max(), s
        if s == 0:
            assert (sb_["status"][5::97] == -100).all() and np.isnan(b[5::97]).all() and np.isfinite(b[0]).all()
        else:
            assert (sb_["status"] != -100).sum() == B - len(range(5, B, 97))  # the un-set-up ones stay so (reference: a null workspace)
    masked.close()
    rest.close()


@pytest.mark.parametrize("N", [16, 8, 32])
def test_full_gait_table_without_a_zero_row(oracle_mod, synth_mod, N):
    """N_gait == n_steps with every horizon step planned: no zero row ends the table, where the reference's construct_gait runs
    one row past its matrix (/root/reference/src/MPC.cpp:686-701, undefined behaviour).  Kernel and oracle stop at the table's end
    (the oracle's one deliberate departure): HIP against the oracle on the full table, and bit-equal to the HIP path
    fed the same rows followed by zero rows in a larger table -- the form that is defined in the reference."""
    import qrw_hip

    eng, refs, worst = run_sequence(oracle_mod, synth_mod, 4, N, ("trot", "walk"), 4, 5000, closed_loop=False, N_gait=N)
    assert worst < 1e-6
    padded = synth_mod.SyntheticBatch(4, N, N_gait=N + 4, gaits=("trot", "walk"), seed0=5000)
    full = synth_mod.SyntheticBatch(4, N, N_gait=N, gaits=("trot", "walk"), seed0=5000)
    a = qrw_hip.Batch(4, n_steps=N, N_gait=N, T_gait=0.02 * N)
    b = qrw_hip.Batch(4, n_steps=N, N_gait=N + 4, T_gait=0.02 * N)
    for s in range(4):
        da, db = full.step(s), padded.step(s)
        assert (np.abs(da["fsteps"]).sum(axis=2) > 0).all()
        oa, ob = a.mpc_solve_host(da["xref"], da["fsteps"], s), b.mpc_solve_host(db["xref"], db["fsteps"], s)
        assert np.array_equal(oa, ob), s
        assert np.array_equal(a.mpc_stats()["iters"], b.mpc_stats()["iters"])
    ga, Sa = a.mpc_gait(0)
    gb, Sb = b.mpc_gait(0)
    assert np.array_equal(ga, gb[:N]) and np.array_equal(Sa, Sb) and np.array_equal(ga, refs[0].get_gait())


@pytest.mark.parametrize("scale", [5.0, 12.0])
def test_large_disturbances_with_saturated_forces(oracle_mod, synth_mod, scale):
    """States far from the reference (the synthetic state noise times 5 / 12, reference velocities times 1.6: up to 0.6 rad of
    roll / pitch error, 1.2 m/s of velocity error) on trot / bounding / pacing / walk: vertical forces sit on their 25 N bound
    (src/MPC.cpp:293-300) and friction-cone rows are active over much of the horizon, solves take up to ~2 000 ADMM iterations
    with several rho updates.  HIP against the oracle: iteration counts, status, rho, results."""
    import qrw_hip

    B, N = 48, 16
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "bounding", "pacing", "walk"), seed0=880000)
    sb.noise_x0 *= scale
    sb.vref *= 1.6
    eng = qrw_hip.Batch(B, N)
    ref = oracle_mod.MPCBatch(B, 0.02, N, 0.32, 20, fast=False)
    threads = max(1, min(16, len(__import__("os").sched_getaffinity(0))))
    saturated = 0
    for s in range(4):
        d = sb.step(s)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        st = eng.mpc_stats()
        r = ref.run(s, d["xref"], d["fsteps"], threads)
        it, stat = ref.iters()
        assert np.array_equal(st["iters"], it) and np.array_equal(st["status"], stat), (s, st["iters"][:8], it[:8])
        for b in range(B):
            assert max(rel_err(out[b, :12], r[b, :12]), rel_err(out[b, 12:], r[b, 12:])) < RTOL, (s, b)
        saturated += int((r[:, 14::3, :] > 24.99).sum())
    assert saturated > 50  # the bound is really active in this test
    assert st["iters"].max() >= 1000


def test_config5_total_batch_on_one_gpu(synth_mod):
    """BASELINE config 5's TOTAL size (8 x 4096 = 32 768 robots, N = 16) as ONE shard on one GPU -- the largest batch anybody asks of a
    handle (1 GB of solver state; sharding.py deals 4096 per GPU): every instance solved, forces of swing feet zero, friction cone
    and unilaterality respected, and a spread of instances bit-equal to a small-batch run of the same instances (an instance's
    result does not depend on the batch it sits in, nor on the longest-first block order a batch this size is launched in)."""
    import qrw_hip

    B, N = 32768, 16
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot",), seed0=20260000)
    eng = qrw_hip.Batch(B, N)
    idx = np.array([0, 1, 4095, 4096, 12345, 20000, 32767])
    small = qrw_hip.Batch(len(idx), N)
    for s in range(3):
        d = sb.step(s)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        o2 = small.mpc_solve_host(d["xref"][idx], d["fsteps"][idx], s)
        assert np.array_equal(out[idx], o2), s
    st = eng.mpc_stats()
    assert (st["status"] == 1).all() and (st["iters"] % 25 == 0).all() and np.isfinite(out).all()
    order, _ = eng.mpc_order()
    assert np.array_equal(np.sort(order), np.arange(B, dtype=np.int32))
    f = out[:, 12:, :].transpose(0, 2, 1).reshape(B, N, 4, 3)
    gait = d["gait"][:, :N]
    assert np.abs(f[gait == 0]).max() < 1e-3
    fs = f[gait == 1]
    mu = np.float64(np.float32(0.9))
    assert (fs[:, 2] > -1e-3).all() and (fs[:, 2] < 25 + 1e-3).all()
    assert (np.abs(fs[:, 0]) <= mu * fs[:, 2] + 1e-3).all() and (np.abs(fs[:, 1]) <= mu * fs[:, 2] + 1e-3).all()
