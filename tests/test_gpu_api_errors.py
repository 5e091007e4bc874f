"""API misuse and edge sizes through the C ABI on the GPU: errors are reported (never a crash, never a silent fallback),
the smallest and the largest supported shapes work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_invalid_configurations_are_refused():
    import qrw_hip

    for kw in (dict(batch=0), dict(batch=-3), dict(batch=2, n_steps=0), dict(batch=2, n_steps=33),
               dict(batch=2, n_steps=16, N_gait=12), dict(batch=2, device=99)):
        args = dict(batch=2, n_steps=16, N_gait=20)
        args.update(kw)
        with pytest.raises(qrw_hip.QrwError):
            qrw_hip.Batch(args.pop("batch"), **args)
    assert b"" != qrw_hip.load_library().qrw_last_error()


def test_wrong_buffers_are_refused(synth_mod):
    import torch
    import qrw_hip

    B, N = 3, 16
    eng = qrw_hip.Batch(B, N)
    d = synth_mod.SyntheticBatch(B, N).step(0)
    with pytest.raises((qrw_hip.QrwError, ValueError)):
        eng.mpc_solve_host(d["xref"][:2], d["fsteps"], 0)             # wrong batch dimension
    x = torch.from_numpy(d["xref"]).cuda()
    f = torch.from_numpy(d["fsteps"]).cuda()
    with pytest.raises(qrw_hip.QrwError):
        eng.mpc_solve(x.float(), f, 0)                                # wrong dtype
    with pytest.raises(qrw_hip.QrwError):
        eng.mpc_solve(x.transpose(1, 2), f, 0)                        # not contiguous / wrong shape
    with pytest.raises(qrw_hip.QrwError):
        eng.mpc_solve(x.cpu(), f, 0)                                  # host tensor on the device API
    with pytest.raises(qrw_hip.QrwError):
        eng.planner_step(0, torch.zeros((B, 7), dtype=torch.float64, device="cuda"),
                         torch.zeros((B, 6), dtype=torch.float64, device="cuda"),
                         torch.zeros((B, 6), dtype=torch.float64, device="cuda"))  # planner not initialised
    out = eng.mpc_solve(x, f, 0)                                      # and the handle still works afterwards
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())


def test_single_instance_and_odd_batch(oracle_mod, synth_mod):
    import qrw_hip

    for B in (1, 67):  # 67: not a multiple of the 16 instances a WBC wavefront holds nor of 64
        sb = synth_mod.SyntheticBatch(B, 16, seed0=990000)
        eng = qrw_hip.Batch(B, 16)
        ref_m = [oracle_mod.MPC(0.02, 16, 0.32, 20) for _ in range(B)]
        ref_w = [oracle_mod.WbcController(0.002) for _ in range(B)]
        for s in range(3):
            d = sb.step(s)
            out = eng.mpc_solve_host(d["xref"], d["fsteps"], s)
            w = eng.wbc_compute_host(d["q"], d["dq"], np.ascontiguousarray(out[:, 12:, 0]), d["contacts"], d["pgoals"],
                                     d["vgoals"], d["agoals"])
            for b in sorted({0, B - 1}):
                ref_m[b].run(s, d["xref"][b], d["fsteps"][b])
                r = ref_m[b].get_latest_result()
                assert np.allclose(out[b], r, rtol=1e-6, atol=1e-8)
                ref_w[b].compute(d["q"][b], d["dq"][b], r[12:, 0], d["contacts"][b], d["pgoals"][b], d["vgoals"][b], d["agoals"][b])
                assert np.allclose(w["tau_ff"][b], ref_w[b].tau_ff, rtol=1e-4, atol=1e-8)


def test_gait_initialize_raises_like_the_reference():
    import libquadruped_reactive_walking as lqrw

    g = lqrw.Gait()
    with pytest.raises(ValueError):  # src/Gait.cpp:30-31 throws std::invalid_argument
        g.initialize(0.02, 0.32, 0.32, 10)


def test_qpwbc_general_pseudo_inverse_of_the_base_block(oracle_mod):
    """src/QPWBC.cpp:486 pseudo-inverts the full 6x6 block M[:6,:6] (pseudoInverse<>, include/qrw/InvKin.hpp:60-66: JacobiSVD,
    singular values below eps * 6 * s_0 dropped).  The reference's caller masks the block to its diagonal (scripts/QP_WBC.py:93),
    which the kernel inverts in place; any other block goes through a Jacobi SVD on the device.  Checked through H = A'Q1 A + Q2
    with A = pinv(Y) X against numpy.linalg.pinv with the reference's threshold (an independent LAPACK SVD) for a symmetric
    positive definite block (the unmasked CRBA), an upper-triangular one (what Pinocchio's crba fills), a non-symmetric one and a
    rank-deficient one, and through f_res / ddq_res against the oracle's QPWBC on the symmetric block."""
    import libquadruped_reactive_walking as lrw

    q = np.zeros(19)
    q[6] = 1.0
    q[7:] = [0.1, 0.7, -1.4, 0.0, 0.6, -1.3, 0.0, -0.7, 1.4, -0.1, -0.7, 1.4]
    M0 = oracle_mod.crba(q)  # unmasked: the base block has off-diagonal terms
    Jc = oracle_mod.feet_jacobians(q)
    f_cmd = np.tile([0.0, 0.0, 6.0], 4)
    RNEA = np.array([0.1, -0.2, 24.0, 0.05, 0.02, -0.01])
    rng = np.random.default_rng(7)
    sym = M0.copy()
    sym[:6, :6] = 0.5 * (M0[:6, :6] + M0[:6, :6].T)
    tri = M0.copy()
    tri[:6, :6] = np.triu(sym[:6, :6])
    gen = M0.copy()
    gen[:6, :6] = sym[:6, :6] + 0.05 * rng.standard_normal((6, 6))
    sing = M0.copy()
    Y = sym[:6, :6].copy()
    Y[:, 5] = Y[:, 4]  # two equal columns: rank 5
    sing[:6, :6] = Y
    X = Jc[:, :6].T
    for name, M in (("symmetric", sym), ("upper triangular", tri), ("non-symmetric", gen), ("rank-deficient", sing)):
        qp = lrw.QPWBC()
        assert qp.run(M, Jc, f_cmd, RNEA, np.zeros(4)) == 0
        A = np.linalg.pinv(M[:6, :6], rcond=6 * np.finfo(float).eps) @ X
        H = A.T @ (0.1 * np.eye(6)) @ A + 5.0 * np.eye(12)
        assert np.allclose(qp.get_H(), H, rtol=1e-9, atol=1e-9), (name, np.abs(qp.get_H() - H).max())
        if name == "symmetric":
            rq = oracle_mod.QPWBC()
            rq.run(M, Jc, f_cmd, RNEA, np.zeros(4))
            assert np.allclose(qp.get_f_res().ravel(), np.asarray(rq.get_f_res()).ravel(), rtol=1e-6, atol=1e-8)
            assert np.allclose(qp.get_ddq_res().ravel(), np.asarray(rq.get_ddq_res()).ravel(), rtol=1e-6, atol=1e-8)
    # the masked block (the hot path's case) still goes the direct way and gives the same as the general routine on it
    Mm = M0.copy()
    Mm[:6, :6] *= np.eye(6)
    qp = lrw.QPWBC()
    assert qp.run(Mm, Jc, f_cmd, RNEA, np.zeros(4)) == 0
    A = np.diag(1.0 / np.diag(Mm[:6, :6])) @ X
    assert np.allclose(qp.get_H(), A.T @ (0.1 * np.eye(6)) @ A + 5.0 * np.eye(12), rtol=1e-12, atol=1e-12)


def test_tensor_on_wrong_device_or_dtype_is_refused():
    import torch

    import qrw_hip

    eng = qrw_hip.Batch(2, 16)
    with pytest.raises(qrw_hip.QrwError):
        eng.mpc_solve(torch.zeros((2, 12, 17), dtype=torch.float32, device="cuda"),
                      torch.zeros((2, 20, 12), dtype=torch.float64, device="cuda"), 0)
    with pytest.raises(qrw_hip.QrwError):
        eng.copy_mpc_iters(torch.zeros((2,), dtype=torch.int64, device="cuda"))
    # the handle's device survives a change of torch's current device only through the explicit device check
    assert eng.device == 0 and torch.cuda.current_device() == 0


def test_time_sliced_give_up_is_reported_at_the_next_call_and_stops_the_robot(synth_mod, oracle_mod, monkeypatch):
    """A time-sliced MPC launch (N > 16) whose queue gives up -- forced here (QRW_PREEMPT_FORCE_GIVEUP=1: the launch starts with its
    error word set, so the queue-fed workgroups leave at once, as they do when one of them has given up after 2 s without progress,
    and nobody finishes the parked solves) -- must be visible where the loop runs: (1) the unfinished results are NaN, (2) the controller's fourth error code stops those robots on the same
    iteration (the reference's three `> limit` tests are blind to NaN, scripts/Controller.py:341-365), (3) the NEXT
    qrw_mpc_solve returns -12, once, without a device sync in between, (4) the call after it runs and starts the unfinished
    instances cold: same iteration count and result as an oracle that is re-created at that call's inputs."""
    import torch
    import qrw_hip
    from Controller import Controller_batch

    monkeypatch.setenv("QRW_PREEMPT_MIN_BATCH", "0")
    monkeypatch.setenv("QRW_PREEMPT_FORCE_GIVEUP", "1")
    B, N, Ng = 6, 32, 36
    sb = synth_mod.SyntheticBatch(B, N, N_gait=Ng, gaits=("trot",), seed0=424200)
    eng = qrw_hip.Batch(B, N, N_gait=Ng, T_gait=0.02 * N)
    d0, d1, d2 = sb.step(0), sb.step(1), sb.step(2)
    x = lambda d: torch.from_numpy(d["xref"]).cuda()
    f = lambda d: torch.from_numpy(d["fsteps"]).cuda()
    out0 = eng.mpc_solve(x(d0), f(d0), 0).clone()          # gives up: every solve needs more than one slice
    torch.cuda.synchronize()
    assert bool(torch.isnan(out0).any(dim=(1, 2)).all()), "every first solve of this workload takes more than 600 iterations"
    with pytest.raises(qrw_hip.QrwError) as ei:
        eng.mpc_solve(x(d1), f(d1), 1)                     # (3): reported here, nothing launched
    assert "(-12)" in str(ei.value)
    with pytest.raises(qrw_hip.QrwError):
        eng.mpc_stats()                                    # the getter still says so too (it reads the launch's own counters)
    monkeypatch.delenv("QRW_PREEMPT_FORCE_GIVEUP")         # (read at creation: `eng` keeps forcing)
    # (4) the next call runs; a forcing handle gives up again, so take a handle without the knob and poke the
    # aborted state into it: pause_it != 0 and the slots holding loop variables -- what the failed launch left behind
    eng2 = qrw_hip.Batch(B, N, N_gait=Ng, T_gait=0.02 * N)
    eng2.mpc_solve(x(d0), f(d0), 0)
    torch.cuda.synchronize()
    assert qrw_hip.load_library().qrw_test_poke_aborted(eng2._handle, 600) == 0
    out = eng2.mpc_solve(x(d1), f(d1), 1)
    torch.cuda.synchronize()
    it = eng2.mpc_stats()["iters"]
    for b in (0, B - 1):
        m = oracle_mod.MPC(0.02, N, 0.02 * N, Ng)
        m.run(0, d0["xref"][b], d0["fsteps"][b])           # sets B / S up as the first call did
        m.cold_start()                                     # OSQP: x, z, y = 0, rho = 0.1 (store_solution after a failed solve)
        m.run(1, d1["xref"][b], d1["fsteps"][b])
        assert m.iter == it[b]
        np.testing.assert_allclose(out[b].cpu().numpy(), m.get_latest_result(), rtol=1e-6, atol=1e-8)

    # (2) through the control loop: NaN forces -> code 4 -> security output on the same iteration
    monkeypatch.setenv("QRW_PREEMPT_FORCE_GIVEUP", "1")
    monkeypatch.setenv("QRW_PREEMPT_CHUNK", "200")         # (a solve from standstill may end within 600 iterations: cut at 200)
    q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
    ctl = Controller_batch(B, q_init, T_gait=0.02 * N, T_mpc=0.02 * N, N_gait=Ng)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    vref = t(np.tile(np.array([0.3, 0.0, 0, 0, 0, 0.1]), (B, 1)))
    qf = np.zeros((B, 19))
    qf[:, 2], qf[:, 6], qf[:, 7:] = 0.2229, 1.0, q_init
    qf, vf, rpy, vs = t(qf), t(np.zeros((B, 18))), t(np.zeros((B, 3))), t(np.zeros((B, 12)))
    raised_at = None
    for k in range(21):
        try:
            r = ctl.compute(vref, qf, vf, rpy, vs)
        except qrw_hip.QrwError as e:
            raised_at = k
            assert "(-12)" in str(e)
            break
        torch.cuda.synchronize()
        if k >= 1:  # (iteration 0 still uses the default result, scripts/MPC_Wrapper.py:123-126)
            assert bool((ctl.error_flag == 4).all()), (k, ctl.error_flag)
            assert float(r.tau_ff.abs().max()) == 0.0 and float(r.P.abs().max()) == 0.0 and bool((r.D == 0.1).all())
    assert raised_at == 10  # the next MPC call of the loop


@pytest.mark.parametrize("pattern", [0xFFFFFFFFFFFFFFFF, 0x7FF0000000000000], ids=["nan", "inf"])
def test_known_answers_do_not_depend_on_what_other_kernels_left_in_lds(pattern):
    """Every compute unit's LDS filled with NaN / Inf (and every vector register with all-ones), then the known-answer solve of every kind of horizon and launch form
    (qrw_test_known_answer, no cache).  Round 4 found the runtime-horizon kernels (N < 16) multiplying a neighbour value of an
    idle step -- derived from LDS nobody had written -- by a zero coefficient: right as long as the leftovers were finite, "solved"
    after 25 iterations with NaN results when they were not (seen once as a failed qrw_create in a test run)."""
    import ctypes as C
    import qrw_hip

    lib = qrw_hip.load_library()
    probe = (C.c_uint32 * 3)()
    assert lib.qrw_test_poison_probe(probe) == 0
    assert list(probe) == [0xFFFFFFFF] * 3, [hex(v) for v in probe]  # the fills are what a new wavefront sees (LDS, VGPR, AGPR)
    for mode in (0, 1, 2):
        for N in (1, 2, 5, 7, 8, 12, 15, 16, 17, 24, 31, 32):
            if mode == 1 and N <= 16:
                continue
            it, st, rho, err = C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
            rc = lib.qrw_test_known_answer(N, mode, pattern, 1, C.byref(it), C.byref(st), C.byref(rho), C.byref(err))
            assert rc == 0, (mode, N, rc, it.value, st.value, rho.value, err.value)


def test_getters_and_host_calls_wait_for_their_own_handle_only(synth_mod):
    """Synchronisation is per handle (include/qrw_hip.h, conventions): while handle B's long N = 32 launch runs on a BLOCKING
    stream (a compute-unit-masked one: the legacy default stream and every blocking copy synchronise with it), handle A's
    getters (`mpc_stats`, `wbc_stats`, `mpc_gait`), a `_host` solve and the creation of a third handle all return -- B's
    event has not completed when they are back.  B's own getter then waits for B.  Before round 5 every one of these calls
    ended in hipDeviceSynchronize."""
    import time

    import torch
    import qrw_hip

    n_cu = qrw_hip.device_cu_count(0)
    N2, Ng2, B2 = 32, 36, 4096
    sb2 = synth_mod.SyntheticBatch(B2, N2, N_gait=Ng2, gaits=("walk", "trot", "bounding"), seed0=515000)
    d2 = sb2.step(0)
    x2, f2 = torch.from_numpy(d2["xref"]).cuda(), torch.from_numpy(d2["fsteps"]).cuda()
    big = qrw_hip.Batch(B2, n_steps=N2, N_gait=Ng2, T_gait=0.02 * N2)
    out2 = torch.empty((B2, 24, N2), dtype=torch.float64, device="cuda")
    sb1 = synth_mod.SyntheticBatch(4, 16, seed0=516000)
    d1 = sb1.step(0)
    small = qrw_hip.Batch(4, 16)
    ref = small.mpc_solve_host(d1["xref"], d1["fsteps"], 0)   # (also: A's state families have launched once)
    small.wbc_compute_host(d1["q"], d1["dq"], np.ascontiguousarray(ref[:, 12:, 0]), d1["contacts"], d1["pgoals"], d1["vgoals"], d1["agoals"])
    side = qrw_hip.CuStream(0, 0, n_cu // 2)  # half the chip, so that A's kernels find free compute units at once
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    with torch.cuda.stream(side.torch):
        for s_ in range(3):  # three launches back to back (~45 ms each on half the chip): a wide margin for A's calls below
            big.mpc_solve(x2, f2, s_, out=out2)
        done.record(side.torch)
    t0 = time.perf_counter()
    st = small.mpc_stats()
    ws = small.wbc_stats()
    gait, _ = small.mpc_gait(0)
    again = small.mpc_solve_host(d1["xref"], d1["fsteps"], 0)
    third = qrw_hip.Batch(2, 16)
    t1 = time.perf_counter()
    still_running = not done.query()
    bst = big.mpc_stats()                      # B's own getter: waits for B's launch (on the masked stream)
    assert done.query()
    assert still_running, "handle A's calls took %.1f ms and outlived handle B's launch" % (1e3 * (t1 - t0))
    assert (st["status"] == 1).all() and (ws["status"] == 1).all() and gait.shape == (20, 4)
    assert np.array_equal(again, ref)          # same inputs, num_iter 0: the cold solve again
    assert np.isin(bst["status"], (1, 2, -2)).all() and (bst["iters"] > 0).all()
    third.close()
    side.close()
    assert (big.mpc_stats()["iters"] == bst["iters"]).all()  # the stream is gone: the getter has nothing to wait for and says the same


def test_two_caller_threads_with_their_own_handles(synth_mod):
    """Handles are single-caller, the library is not: two Python threads (ctypes releases the GIL during the calls), each creating its own
    handles (the known-answer gate and the registry of live handles are behind mutexes, the error text is thread-local) and stepping
    MPC + WBC on its own stream, get bit for bit what one thread gets alone."""
    import threading

    import torch
    import qrw_hip

    B, N, S = 96, 16, 5
    sbs = [synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=616000 + 1000 * i) for i in range(2)]
    data = [[sb.step(s) for s in range(S)] for sb in sbs]

    def run(i, out):
        with torch.cuda.stream(torch.cuda.Stream()):
            res = []
            for rep in range(3):  # handles created and destroyed while the other thread is mid-launch
                eng = qrw_hip.Batch(B, N)
                for s in range(S):
                    d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in data[i][s].items() if k != "gait" and k != "x0"}
                    o = eng.mpc_solve(d["xref"], d["fsteps"], s)
                    w = eng.wbc_compute(d["q"], d["dq"], o[:, 12:, 0].contiguous(), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"])
                torch.cuda.current_stream().synchronize()
                res.append((o.cpu().numpy().copy(), w["tau_ff"].cpu().numpy().copy(), eng.mpc_stats()["iters"].copy()))
                eng.close()
            out[i] = res

    alone = [None, None]
    for i in range(2):
        run(i, alone)
    both = [None, None]
    errs = []

    def guarded(i):
        try:
            run(i, both)
        except Exception as e:  # pragma: no cover - reported below
            errs.append(repr(e))

    ts = [threading.Thread(target=guarded, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        for rep in range(3):
            for a, b in zip(alone[i][rep], both[i][rep]):
                assert np.array_equal(a, b), (i, rep)


def test_one_handle_per_visible_device_stepped_from_one_thread(synth_mod):
    """Multi-GPU readiness (the batch is sharded over the GPUs of a node, one shard per device): ONE
    process creates a handle on EVERY visible device and steps them from one thread while torch's current device stays 0 -- every
    entry point has to switch to its handle's device and back (DeviceScope, qrw_api.hip), the known-answer gate runs per device,
    and tensors on another device than the handle's are refused.  Every device must return, bit for bit, what device 0 returns for
    the same shard (same kernels, same arithmetic).  Skips on a box with one GPU: the first multi-GPU box that runs the suite
    exercises it."""
    import torch

    import qrw_hip

    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one visible device: DeviceScope's switch has nothing to switch to")
    n_dev = min(n_dev, 6)  # (the pool's process guard: at most 6 of a user's processes / contexts on one box's cards)
    B, N, S = 64, 16, 3
    sb = synth_mod.SyntheticBatch(B, N, gaits=("trot", "walk"), seed0=717000)
    steps = [sb.step(s) for s in range(S)]
    keys = ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")
    engs = [qrw_hip.Batch(B, N, device=d) for d in range(n_dev)]
    assert torch.cuda.current_device() == 0
    outs = [[] for _ in range(n_dev)]
    for s in range(S):
        for d in range(n_dev):  # interleaved: the current device never changes on the caller's side
            t = {k: torch.from_numpy(np.ascontiguousarray(steps[s][k])).to("cuda:%d" % d) for k in keys}
            o = engs[d].mpc_solve(t["xref"], t["fsteps"], s)
            w = engs[d].wbc_compute(t["q"], t["dq"], o[:, 12:, 0].contiguous(), t["contacts"], t["pgoals"], t["vgoals"], t["agoals"])
            outs[d].append((o, w["tau_ff"]))
            assert torch.cuda.current_device() == 0 and o.device.index == d
    for d in range(n_dev):
        torch.cuda.synchronize(d)
    for d in range(1, n_dev):
        for s in range(S):
            assert torch.equal(outs[d][s][0].cpu(), outs[0][s][0].cpu()) and torch.equal(outs[d][s][1].cpu(), outs[0][s][1].cpu()), (d, s)
        assert np.array_equal(engs[d].mpc_stats()["iters"], engs[0].mpc_stats()["iters"])
        host = engs[d].mpc_solve_host(steps[0]["xref"], steps[0]["fsteps"], 0)  # the _host path on device d, current device 0
        assert np.array_equal(host, outs[0][0][0].cpu().numpy()) and torch.cuda.current_device() == 0
    with pytest.raises(qrw_hip.QrwError):  # a tensor of device 0 handed to device 1's handle
        engs[1].mpc_solve(torch.zeros((B, 12, N + 1), dtype=torch.float64, device="cuda:0"),
                          torch.zeros((B, 20, 12), dtype=torch.float64, device="cuda:0"), 0)
    for e in engs:
        e.close()
