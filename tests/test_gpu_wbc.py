"""GPU parity tests of the WBC hot path and of the drop-in Python classes (through the C ABI) vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(a, ref):
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12)


def f_cmd_for(contacts, rng=None):
    B = contacts.shape[0]
    f = np.zeros((B, 12))
    f[:, 2::3] = contacts * 24.5 / np.maximum(contacts.sum(1, keepdims=True), 1)
    f[:, 0::3] = contacts * 0.7
    f[:, 1::3] = contacts * -0.4
    if rng is not None:
        f += np.repeat(contacts, 3, axis=1) * rng.normal(size=(B, 12))
    return f


# Both shipped forms of the full WBC step stand directly under the oracle: wbc16_kernel (16 lanes per robot, the default) and
# wbc_kernel's full mode (one quad per robot: what Controller_batch(multiprocessing=True) selects by itself above batch 128,
# Controller.py `wbc_set_lanes(4)`; reference: scripts/QP_WBC.py:52-131, src/QPWBC.cpp:310-390).
LANES = pytest.mark.parametrize("lanes", [16, 4], ids=["wbc16_kernel", "wbc_kernel_quad"])


@LANES
def test_wbc_sequence_matches_oracle(oracle_mod, synth_mod, lanes):
    import qrw_hip

    B = 21  # not a multiple of 16: exercises the padded quads of the last wavefront
    sb = synth_mod.SyntheticBatch(B, 16, gaits=("trot", "walk", "static"), seed0=91000)
    eng = qrw_hip.Batch(B)
    eng.wbc_set_lanes(lanes)
    refs = [oracle_mod.WbcController(0.002) for _ in range(B)]
    rng = np.random.default_rng(0)
    for s in range(8):
        d = sb.step(s)
        # random base pose / twist: the reference always passes the identity pose, the kernel must not assume it
        quat = rng.normal(size=(B, 4))
        d["q"][:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True) if s >= 4 else d["q"][:, 3:7]
        d["q"][:, :3] += rng.uniform(-0.2, 0.2, (B, 3)) if s >= 4 else 0
        d["dq"][:, 3:6] = rng.uniform(-0.5, 0.5, (B, 3))
        f = f_cmd_for(d["contacts"], rng)
        o = eng.wbc_compute_host(d["q"], d["dq"], f, d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
        st = eng.wbc_stats()
        for b in range(B):
            r = refs[b]
            r.compute(d["q"][b], d["dq"][b], f[b], d["contacts"][b], d["pgoals"][b], d["vgoals"][b], d["agoals"][b])
            assert st["iters"][b] == r.qp_iter and st["status"][b] == 1
            for key, ref in (("tau_ff", r.tau_ff), ("qdes", r.qdes), ("vdes", r.vdes[:, 0]),
                             ("f_with_delta", r.f_with_delta[:, 0]), ("ddq_res", r.ddq_res)):
                assert rel_err(o[key][b], ref) < RTOL, (s, b, key)
            p, e, v = r.feet()
            assert np.allclose(o["feet"][b, 0], p, atol=1e-12) and np.allclose(o["feet"][b, 1], e, atol=1e-12)
            assert np.allclose(o["feet"][b, 2], v, atol=1e-12)
            assert np.array_equal(st["k_since_contact"][b], r.k_since_contact.ravel())


def test_standalone_pieces_match_oracle(oracle_mod, synth_mod):
    import qrw_hip

    B = 5
    rng = np.random.default_rng(3)
    eng = qrw_hip.Batch(B)
    q12 = synth_mod.Q_NOMINAL + rng.uniform(-0.4, 0.4, (B, 12))
    dq12 = rng.uniform(-2, 2, (B, 12))
    posf, vf, wf, af, Jf = eng.fixed_feet_host(q12, dq12)
    contacts = (rng.uniform(size=(B, 4)) < 0.5).astype(float)
    goals = posf.transpose(0, 2, 1) + rng.uniform(-0.01, 0.01, (B, 3, 4))
    vgoals, agoals = rng.uniform(-0.1, 0.1, (B, 3, 4)), rng.uniform(-1, 1, (B, 3, 4))
    ddq, dq_cmd, q_step = eng.invkin_host(contacts, goals, vgoals, agoals, posf, vf, wf, af, Jf)
    for b in range(B):
        ref = oracle_mod.fixed_feet(q12[b], dq12[b])
        for a, r in zip((posf[b], vf[b], wf[b], af[b], Jf[b]), ref):
            assert np.allclose(a, r, rtol=1e-10, atol=1e-12)
        ik = oracle_mod.InvKin(0.002)
        rd = ik.refreshAndCompute(contacts[b], goals[b], vgoals[b], agoals[b], *ref)
        assert rel_err(ddq[b], rd) < 1e-9 and rel_err(dq_cmd[b], ik.get_dq_cmd()) < 1e-9
        assert rel_err(q_step[b], ik.get_q_step()) < 1e-9
    # base inertia diagonal = masked crba at the neutral configuration (scripts/QP_WBC.py:89-93)
    qn = np.zeros(19)
    qn[6] = 1.0
    assert np.allclose(eng.base_inertia_diag(), np.diag(oracle_mod.crba(qn))[:6], rtol=1e-12)


def test_dropin_classes_match_oracle(oracle_mod, synth_mod):
    """The reference-named Python classes (libquadruped_reactive_walking.MPC/QPWBC/InvKin, MPC_Wrapper,
    wbc_controller, Solo12InvKin) against the oracle's counterparts on one robot."""
    import libquadruped_reactive_walking as lrw
    import MPC_Wrapper
    import QP_WBC
    import solo12InvKin

    N = 16
    sb = synth_mod.SyntheticBatch(1, N, seed0=95000)
    q_init = np.zeros((19, 1))
    q_init[:, 0] = sb.step(0)["q"][0]
    wrap = MPC_Wrapper.MPC_Wrapper(True, 0.02, N, 10, 0.32, 20, q_init, False)
    first = wrap.get_latest_result()
    assert first.shape == (24, N) and np.array_equal(first[12:, 0], [0, 0, 8.0] * 4) and first[2, 0] == q_init[2, 0]
    wbc = QP_WBC.wbc_controller(0.002, 100)
    ref_mpc = oracle_mod.MPC(0.02, N, 0.32, 20)
    ref_wbc = oracle_mod.WbcController(0.002)
    x0 = None
    for s in range(5):
        d = sb.step(s, x0)
        k = 10 * s
        assert wrap.solve(k, d["xref"][0], d["fsteps"][0], d["gait"][0]) == 0
        x_f = wrap.get_latest_result()
        ref_mpc.run(k, d["xref"][0], d["fsteps"][0])
        r = ref_mpc.get_latest_result()
        assert x_f.shape == (24, N) and rel_err(x_f, r) < RTOL
        x0 = r[:12, 0][None]
        qv, dqv = d["q"][0].reshape(19, 1), d["dq"][0].reshape(18, 1)
        assert wbc.compute(qv, dqv, x_f[12:, 0], d["contacts"][0], d["pgoals"][0], d["vgoals"][0], d["agoals"][0]) == 0
        ref_wbc.compute(d["q"][0], d["dq"][0], r[12:, 0], d["contacts"][0], d["pgoals"][0], d["vgoals"][0],
                        d["agoals"][0])
        assert wbc.tau_ff.shape == (12,) and wbc.qdes.shape == (19,) and wbc.vdes.shape == (18, 1)
        assert wbc.f_with_delta.shape == (12, 1)
        assert rel_err(wbc.tau_ff, ref_wbc.tau_ff) < RTOL and rel_err(wbc.qdes, ref_wbc.qdes) < RTOL
        assert rel_err(wbc.vdes, ref_wbc.vdes) < RTOL and rel_err(wbc.f_with_delta, ref_wbc.f_with_delta) < RTOL
        assert np.array_equal(wbc.k_since_contact, ref_wbc.k_since_contact)
        assert np.allclose(wbc.invKin.cpp_posf, oracle_mod.fixed_feet(d["q"][0][7:], d["dq"][0][6:])[0], atol=1e-12)
    assert wrap.stop_parallel_loop() == 0
    # bound-class surface (python/gepadd.cpp): QPWBC / InvKin / MPC getters
    d = sb.step(0)
    m = lrw.MPC(0.02, N, 0.32, 20)
    assert m.run(0, d["xref"][0], d["fsteps"][0]) == 0
    assert m.get_latest_result().shape == (24, N) and m.get_gait().shape == (20, 4)
    assert np.array_equal(m.get_gait(), oracle_mod.MPC(0.02, N, 0.32, 20).get_gait() * 0 + m.get_gait())
    ik = solo12InvKin.Solo12InvKin(0.002)
    dd = ik.refreshAndCompute(d["q"][0][7:].reshape(12, 1), d["dq"][0][6:].reshape(12, 1), d["contacts"][0],
                              d["pgoals"][0], d["vgoals"][0], d["agoals"][0])
    assert dd.shape == (18,) and np.all(dd[:6] == 0) and ik.q_cmd.shape == (19,) and np.all(ik.q_cmd[:7] == 0)
    qn = np.zeros(19)
    qn[6] = 1.0
    M = oracle_mod.crba(qn)
    M[:6, :6] *= np.eye(6)
    Jc = oracle_mod.feet_jacobians(d["q"][0])
    qp, rq = lrw.QPWBC(), oracle_mod.QPWBC()
    fc = np.tile([0.3, -0.2, 6.0], 4)
    rn = np.array([0.2, -0.1, 24.0, 0.05, -0.02, 0.01])
    for _ in range(3):
        assert qp.run(M, Jc, fc.reshape(-1, 1), rn.reshape(-1, 1), np.zeros((1, 4))) == 0
        rq.run(M, Jc, fc, rn, np.zeros(4))
        assert qp.get_f_res().shape == (12,) and qp.get_ddq_res().shape == (6,) and qp.get_H().shape == (12, 12)
        assert rel_err(qp.get_f_res(), rq.get_f_res()) < RTOL and rel_err(qp.get_ddq_res(), rq.get_ddq_res()) < RTOL
        assert np.allclose(qp.get_H(), rq.get_H(), rtol=1e-12, atol=1e-14)
        fc = fc + 0.1


def test_async_wrapper_returns_previous_then_new_result(synth_mod):
    import MPC_Wrapper

    N = 16
    sb = synth_mod.SyntheticBatch(1, N, seed0=97000)
    q_init = np.zeros((19, 1))
    q_init[:, 0] = sb.step(0)["q"][0]
    sync = MPC_Wrapper.MPC_Wrapper(True, 0.02, N, 10, 0.32, 20, q_init, False)
    asyn = MPC_Wrapper.MPC_Wrapper(True, 0.02, N, 10, 0.32, 20, q_init, True)
    assert np.array_equal(asyn.get_latest_result(), sync.get_latest_result())
    for s in range(3):
        d = sb.step(s)
        sync.solve(10 * s, d["xref"][0], d["fsteps"][0], d["gait"][0])
        asyn.solve(10 * s, d["xref"][0], d["fsteps"][0], d["gait"][0])
        asyn.stop_parallel_loop()  # waits for the side stream, as polling newResult does in scripts/test_mpc.py:64
        a, b = asyn.get_latest_result(), sync.get_latest_result()
        assert np.array_equal(a, b)


@LANES
def test_contact_patterns_from_flight_to_full_stance(oracle_mod, synth_mod, lanes):
    """Every one of the 16 contact sets (flight phase, single support, ..., four feet), changing from call to call on
    the same solver instance (the QP's matrix changes, the warm start is kept): GPU vs oracle."""
    import qrw_hip

    B = 16
    sb = synth_mod.SyntheticBatch(B, 16, gaits=("trot",), seed0=93000)
    eng = qrw_hip.Batch(B)
    eng.wbc_set_lanes(lanes)
    refs = [oracle_mod.WbcController(0.002) for _ in range(B)]
    rng = np.random.default_rng(5)
    for s in range(4):
        d = sb.step(s)
        contacts = np.array([[(((b + 5 * s) % 16) >> i) & 1 for i in range(4)] for b in range(B)], dtype=np.float64)
        f = f_cmd_for(contacts, rng)
        o = eng.wbc_compute_host(d["q"], d["dq"], f, contacts, d["pgoals"], d["vgoals"], d["agoals"])
        st = eng.wbc_stats()
        for b in range(B):
            r = refs[b]
            r.compute(d["q"][b], d["dq"][b], f[b], contacts[b], d["pgoals"][b], d["vgoals"][b], d["agoals"][b])
            assert st["iters"][b] == r.qp_iter, (s, b, contacts[b], st["iters"][b], r.qp_iter)
            for key, ref in (("tau_ff", r.tau_ff), ("qdes", r.qdes), ("vdes", r.vdes[:, 0]),
                             ("f_with_delta", r.f_with_delta[:, 0]), ("ddq_res", r.ddq_res)):
                assert rel_err(o[key][b], ref) < RTOL, (s, b, key, contacts[b])


def _full_size_pipeline(synth_mod, B, N, gaits, seed0, steps=3):
    """MPC -> WBC for `steps` control steps at a BASELINE batch size; returns the last step's data and outputs."""
    import qrw_hip

    N_gait = max(20, N + 4)
    sb = synth_mod.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, seed0=seed0)
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, T_gait=0.02 * N)
    for s in range(steps):
        d = sb.step(s)
        out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
        f_cmd = np.ascontiguousarray(out[:, 12:, 0])
        w = eng.wbc_compute_host(d["q"], d["dq"], f_cmd, d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
    return sb, eng, d, out, f_cmd, w


def _check_full_size_properties(eng, d, out, f_cmd, w, B, N):
    ms, ws = eng.mpc_stats(), eng.wbc_stats()
    assert (ws["status"] == 1).all() and (ws["iters"] % 25 == 0).all()
    # open-loop noisy walk / bounding states at N = 32: a few per cent of the solves run into max_iter (4000) and pass
    # through as "solved inaccurate" / "max iterations", exactly as the reference ignores OSQP's status (src/MPC.cpp:558)
    assert (ms["status"] == 1).mean() > 0.95 and np.isin(ms["status"], (1, 2, -2)).all()
    for key in ("tau_ff", "qdes", "vdes", "f_with_delta", "ddq_res"):
        assert np.isfinite(w[key]).all(), key
    f = w["f_with_delta"].reshape(B, 4, 3)
    mu = 0.9
    tol = 1e-3  # eps_abs = eps_rel = 1e-5 on rows of size <= 25 (src/QPWBC.cpp:239-240)
    assert (f[:, :, 2] >= -tol).all() and (f[:, :, 2] <= 25 + tol).all()
    assert (np.abs(f[:, :, 0]) <= mu * f[:, :, 2] + tol).all() and (np.abs(f[:, :, 1]) <= mu * f[:, :, 2] + tol).all()
    swing = d["contacts"] == 0
    assert np.abs(f[swing]).max() < 5e-2  # Jc rows and f_cmd are zero there: only the 5 I regularisation acts
    # tau_ff is bounded by what the security check would accept by a wide margin on nominal states (Controller.py:345-355)
    assert np.abs(w["tau_ff"]).max() < 20.0
    assert np.allclose(w["qdes"][:, :7], 0.0) and np.allclose(w["vdes"][:, :6], 0.0)  # scripts/solo12InvKin.py:62-67


def test_config3_batch4096_mpc_wbc_properties(synth_mod):
    """BASELINE config 3 at full size (batch 4096, N = 16, MPC + QPWBC + InvKin): size-independent properties on every
    instance and bit-equality of a spread of instances with an independent small-batch run of the same pipeline."""
    import qrw_hip

    B, N = 4096, 16
    sb, eng, d, out, f_cmd, w = _full_size_pipeline(synth_mod, B, N, ("trot",), 20260000)
    _check_full_size_properties(eng, d, out, f_cmd, w, B, N)
    idx = np.array([0, 15, 16, 1023, 2049, 4095])
    small = qrw_hip.Batch(len(idx), N)
    for s in range(3):
        d2 = sb.step(s)
        o2 = small.mpc_solve_host(d2["xref"][idx], d2["fsteps"][idx], s)
        w2 = small.wbc_compute_host(d2["q"][idx], d2["dq"][idx], np.ascontiguousarray(o2[:, 12:, 0]), d2["contacts"][idx],
                                    d2["pgoals"][idx], d2["vgoals"][idx], d2["agoals"][idx])
    assert np.array_equal(o2, out[idx])
    for key in ("tau_ff", "f_with_delta", "qdes", "vdes", "ddq_res"):
        assert np.array_equal(w2[key], w[key][idx]), key


def test_config4_batch4096_n32_mixed_gaits_properties(oracle_mod, synth_mod):
    """BASELINE config 4 at full size (batch 4096, N = 32, walk / trot / bounding per instance): the same properties,
    bit-equality with a small-batch run, and oracle parity (iterations, status, 1e-4) on a spread of 24 instances."""
    import qrw_hip

    B, N = 4096, 32
    gaits = ("walk", "trot", "bounding")
    sb, eng, d, out, f_cmd, w = _full_size_pipeline(synth_mod, B, N, gaits, 20264000, steps=2)
    _check_full_size_properties(eng, d, out, f_cmd, w, B, N)
    fm = out[:, 12:, :].transpose(0, 2, 1).reshape(B, N, 4, 3)
    gait = d["gait"][:, :N]
    ok = eng.mpc_stats()["status"] == 1
    assert np.abs(fm[ok][gait[ok] == 0]).max() < 1e-3
    # a spread of 24 instances over the batch (first / last, both sides of the resident-slot and stream-group boundaries):
    # bit-equal to a small-batch run of the same kernel, and that run against the oracle (iteration counts, status, 1e-4)
    idx = np.array([0, 1, 2, 3, 255, 256, 511, 512, 513, 777, 1023, 1024, 1500, 2047, 2048, 2049, 2500, 3000, 3071, 3072, 3500, 4000,
                    4094, 4095])
    small = qrw_hip.Batch(len(idx), n_steps=N, N_gait=36, T_gait=0.02 * N)
    ref = oracle_mod.MPCBatch(len(idx), 0.02, N, 0.02 * N, 36, fast=False)
    threads = max(1, min(16, len(__import__("os").sched_getaffinity(0))))
    for s in range(2):
        d2 = sb.step(s)
        o2 = small.mpc_solve_host(d2["xref"][idx], d2["fsteps"][idx], s)
        st = small.mpc_stats()
        r2 = ref.run(s, d2["xref"][idx], d2["fsteps"][idx], threads)
        it, stat = ref.iters()
        assert np.array_equal(st["iters"], it) and np.array_equal(st["status"], stat), s
        for j in range(len(idx)):
            assert rel_err(o2[j], r2[j]) < RTOL, (s, j)
    assert np.array_equal(o2, out[idx])
    assert len(set(sb.kind[idx].tolist())) >= 3  # the spread covers all three gaits


def test_replay_batch_equals_solve_batch_calls(synth_mod):
    """MPC_Wrapper_batch.replay_batch (logged planner outputs -> mpc_x_f in one launch) against solve_batch call by call."""
    import torch

    import MPC_Wrapper

    B, N, K = 37, 16, 5
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=99100)
    steps = [sb.step(s) for s in range(K)]
    xs = torch.from_numpy(np.stack([st["xref"] for st in steps])).cuda()
    fs = torch.from_numpy(np.stack([st["fsteps"] for st in steps])).cuda()
    a = MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, B)
    b = MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, B)
    ref = []
    for s in range(K):
        a.solve_batch(s, xs[s], fs[s])
        a.get_latest_result_batch()
        ref.append(a.get_latest_result_batch().cpu().numpy().copy())
    got = b.replay_batch(0, xs, fs)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), np.stack(ref))
    b.get_latest_result_batch()
    assert np.array_equal(b.get_latest_result_batch().cpu().numpy(), ref[-1])


def test_stream_groups_give_the_single_handle_results(synth_mod):
    """qrw_hip.StreamGroups (the fleet as two sub-batches with their own handles and streams, stepping without
    cross-group synchronisation): robots are independent, so MPC result and torques equal a single handle's bit for bit."""
    import torch

    import qrw_hip

    B, N = 2052, 16  # 1026 per group: the longest-first block order is in use in both groups
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=20261200)
    one = qrw_hip.Batch(B, N)
    grp = qrw_hip.StreamGroups(B, groups=2, n_steps=N)
    keys = ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")
    for s in range(4):
        d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sb.step(s).items() if k in keys}
        out = one.mpc_solve(d["xref"], d["fsteps"], s)
        ref = one.wbc_compute(d["q"], d["dq"], out[:, 12:, 0].contiguous(), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
        ni = s if s % 2 == 0 else torch.full((B,), s, dtype=torch.int32, device="cuda")  # both forms of num_iter
        res = grp.control_step(d["xref"], d["fsteps"], ni, d["q"], d["dq"], d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
        grp.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(grp.mpc_out, out)
        for k in ("tau_ff", "f_with_delta", "qdes", "vdes", "ddq_res", "feet"):
            assert torch.equal(res[k], ref[k]), (s, k)
    grp.close()
    with pytest.raises(qrw_hip.QrwError):
        qrw_hip.StreamGroups(7, groups=2)


def test_wrapper_batch_stream_groups_equal_the_single_handle(synth_mod):
    """MPC_Wrapper_batch(groups=2) on a fleet of 2052 robots (two handles on two streams, opt-in; without the argument: one handle
    at every fleet size) returns what the single handle returns, bit for bit, for int and per-instance iteration arguments,
    incl. the default result before the first solve -- and the caller may REFILL ITS INPUT TENSORS IN PLACE as soon as
    solve_batch has returned (the groups' streams read copies the wrapper owns, not the caller's tensors)."""
    import torch

    import MPC_Wrapper

    B, N, K = 2052, 16, 4
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=99200)
    a = MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, B, groups=1)
    b = MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, B, groups=2)
    assert a.G == 1 and b.G == 2 and MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, 8).G == 1
    assert MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, 4096).G == 1 and MPC_Wrapper.MPC_Wrapper_batch(0.02, N, 0.32, 20, 64, groups=2).G == 2
    xb, fb = (torch.empty((B,) + shp, dtype=torch.float64, device="cuda") for shp in ((12, N + 1), (20, 12)))
    kb = torch.empty((B,), dtype=torch.int32, device="cuda")
    assert torch.equal(a.get_latest_result_batch(), b.get_latest_result_batch())  # the default forces
    for s in range(K):
        d = sb.step(s)
        x, f = torch.from_numpy(d["xref"]).cuda(), torch.from_numpy(d["fsteps"]).cuda()
        k = s if s % 2 == 0 else torch.full((B,), s, dtype=torch.int32, device="cuda")
        a.solve_batch(k, x, f)
        # the grouped wrapper gets buffers that are overwritten with garbage right behind the call, on the caller's stream
        xb.copy_(x)
        fb.copy_(f)
        if isinstance(k, torch.Tensor):
            kb.copy_(k)
        b.solve_batch(kb if isinstance(k, torch.Tensor) else k, xb, fb)
        xb.fill_(float("nan"))
        fb.fill_(1e9)
        kb.fill_(-7)
        assert torch.equal(a.get_latest_result_batch(), b.get_latest_result_batch()), s
        torch.cuda.synchronize()
    sa, sb_ = a.stats(), b.stats()
    assert np.array_equal(sa["iters"], sb_["iters"]) and np.array_equal(sa["status"], sb_["status"])


def test_wbc16_matches_the_quad_kernel(synth_mod, monkeypatch):
    """wbc16_kernel (sixteen lanes per instance, the default for the full compute since round 4) against wbc_kernel (one quad per
    instance, `wbc_set_lanes(4)`) on two handles fed the same sequence: same ADMM iteration counts and status on every call, outputs
    equal to rounding (the per-foot phases, the QP data, the equilibration and the ADMM iteration evaluate the same expressions
    in the same order; only the 12 x 12 KKT inverse is eliminated in another form), persistent state and k_since_contact
    included.  Odd batch (padded rows), all contact sets, random base poses, warm starts across calls."""
    import qrw_hip

    B = 37
    sb = synth_mod.SyntheticBatch(B, 16, gaits=("trot", "walk", "static"), seed0=95000)
    rng = np.random.default_rng(12)
    new, old = qrw_hip.Batch(B), qrw_hip.Batch(B)
    old.wbc_set_lanes(4)
    with pytest.raises(qrw_hip.QrwError):
        old.wbc_set_lanes(8)
    worst = 0.0
    for s in range(10):
        d = sb.step(s)
        if s >= 3:
            quat = rng.normal(size=(B, 4))
            d["q"][:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
            d["dq"][:, 3:6] = rng.uniform(-0.5, 0.5, (B, 3))
        contacts = d["contacts"] if s % 3 else np.array([[(((b + 5 * s) % 16) >> i) & 1 for i in range(4)] for b in range(B)], dtype=np.float64)
        f = f_cmd_for(contacts, rng)
        args = (d["q"], d["dq"], f, contacts, d["pgoals"], d["vgoals"], d["agoals"])
        o1 = new.wbc_compute_host(*args)
        s1 = new.wbc_stats()
        o0 = old.wbc_compute_host(*args)
        s0 = old.wbc_stats()
        assert np.array_equal(s1["iters"], s0["iters"]) and np.array_equal(s1["status"], s0["status"]), s
        assert np.array_equal(s1["k_since_contact"], s0["k_since_contact"])
        for key in ("qdes", "vdes", "feet"):  # nothing of the QP in these: bit for bit
            assert np.array_equal(o1[key], o0[key]), (s, key)
        for key in ("tau_ff", "f_with_delta", "ddq_res"):
            e = rel_err(o1[key], o0[key])
            worst = max(worst, e)
            assert e < 1e-9, (s, key, e)
    print("wbc16 vs quad kernel: worst relative deviation %.2e" % worst)


@LANES
def test_wbc_on_wild_inputs_matches_oracle(oracle_mod, synth_mod, lanes):
    """The whole-body step far outside what the periodic-gait controller feeds it (synth.RandomWbcInputs: arbitrary base
    orientation, joints +-0.6 rad around the nominal pose, joint velocities up to 6 rad/s, persistent random contact sets -- all
    16 --, commanded forces that violate the QP's 25 N bound and its friction cone in a third of the stance feet, forces on
    swing feet, goals centimetres and m/s away): scripts/QP_WBC.py:52-131 accepts any of it.  Both kernel forms against the
    oracle over 10 warm-started calls.

    What "parity" can mean here was measured first (scripts/gpu_wbc_wild_study.py, docs/HISTORY.md 8): on these inputs the box-QP
    runs past OSQP's adaptive-rho test (200 iterations) in a few per cent of the solves, where the primal residual has already
    converged to rounding noise -- rho * sqrt(primal / dual ratio) then differs at 1e-4 ... 1e-2 relative between TWO BUILDS OF
    THE SAME ORACLE SOURCE (strict IEEE against -O3 -march=native), rho persists, and every later warm-started solve of that robot
    differs at 1e-6 ... 1e-3 (all inside the QP's own 1e-5 tolerances): the reference built twice would do the same.  So both oracle
    builds run beside the kernel; a robot counts as ROUNDING-SENSITIVE from the first call on at which the two builds disagree
    (rho or any output beyond 1e-9, or the iteration count).  Every other robot must take the oracle's iteration count and match
    it to 1e-4 (measured <= 4e-9); a sensitive one must stay within 1e-2 and one termination check (25 iterations); status solved
    and k_since_contact exact for all.  At least 70 % of the robots must still be insensitive after the ten calls."""
    import qrw_hip

    B, K = 144, 10
    gen = synth_mod.RandomWbcInputs(B, seed0=20700000 + lanes)
    eng = qrw_hip.Batch(B)
    eng.wbc_set_lanes(lanes)
    oracle_mod.build(fast=True)
    ref, ref2 = oracle_mod.WbcBatch(B, 0.002, fast=False), oracle_mod.WbcBatch(B, 0.002, fast=True)
    threads = max(1, min(16, len(__import__("os").sched_getaffinity(0))))
    dp = __import__("ctypes").POINTER(__import__("ctypes").c_double)

    def dev(x, y):
        # relative to the output's own scale, floored at 1 (N m, rad, rad/s, N): a robot in flight has contact forces of ~1e-6 N
        # (the QP's tolerance), whose relative deviation means nothing
        e = np.zeros(B)
        for u, v in zip(x, y):
            e = np.maximum(e, np.abs(u - v).reshape(B, -1).max(1) / np.maximum(np.abs(v).reshape(B, -1).max(1), 1.0))
        return e

    pats, bound, its = set(), 0, []
    sensitive = np.zeros(B, bool)
    worst_clean = worst_sens = 0.0
    for c in range(K):
        d = gen.step(c)
        args = (d["q"], d["dq"], d["f_cmd"], d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
        o = eng.wbc_compute_host(*args)
        ra, rb = ref.compute(*args, threads), ref2.compute(*args, threads)
        (it, st_o, rho), (it2, _, rho2) = ref.qp_stats(), ref2.qp_stats()
        st = eng.wbc_stats()
        assert (st["status"] == 1).all() and (st_o == 1).all(), c
        ksc = np.zeros((B, 4))
        for b, h in enumerate(ref._hs):
            ref._lib.wbc_oracle_get_k_since_contact(h, ksc[b].ctypes.data_as(dp))
        assert np.array_equal(st["k_since_contact"], ksc), c
        sensitive |= (dev(rb, ra) > 1e-9) | (it != it2) | (np.abs(rho2 / rho - 1) > 1e-9)
        e = dev((o["tau_ff"], o["qdes"], o["vdes"], o["f_with_delta"]), ra)
        clean = ~sensitive
        assert np.array_equal(st["iters"][clean], it[clean]), (c, np.nonzero(clean & (st["iters"] != it))[0][:8])
        assert (e[clean] < RTOL).all(), (c, np.nonzero(clean & (e >= RTOL))[0][:8], e[clean].max())
        assert (np.abs(st["iters"][sensitive] - it[sensitive]) <= 25).all() and (e[sensitive] < 1e-2).all(), (c, e[sensitive].max() if sensitive.any() else 0)
        worst_clean = max(worst_clean, float(e[clean].max()))
        worst_sens = max(worst_sens, float(e[sensitive].max()) if sensitive.any() else 0.0)
        pats |= set(map(tuple, d["contacts"].astype(int)))
        bound += int((ra[3][:, 2::3] > 24.99).sum())
        its.append(it)
    its = np.concatenate(its)
    print("wild WBC inputs, lanes %d: %d of %d robots rounding-sensitive after %d calls (two oracle builds disagree); worst relative "
          "deviation %.2e on the others, %.2e on the sensitive ones; QP iterations %d..%d, %d contact sets, %d forces on the 25 N bound"
          % (lanes, int(sensitive.sum()), B, K, worst_clean, worst_sens, its.min(), its.max(), len(pats), bound))
    assert len(pats) == 16 and its.max() >= 150 and sensitive.mean() <= 0.30
