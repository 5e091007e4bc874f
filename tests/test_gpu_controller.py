"""GPU parity tests of the controller glue (controller_kernel.hip through the C ABI) against oracle/controller_oracle.py,
and of the closed device-resident control loop (Controller_batch) against the chained CPU oracles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q_INIT = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])  # scripts/main_solo12_control.py:120


def _t(x):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()


def test_glue_stages_match_oracle():
    import torch
    import controller_oracle as co
    import qrw_hip

    B, N = 7, 16
    rng = np.random.default_rng(11)
    eng = qrw_hip.Batch(B, N)
    qi = Q_INIT + rng.uniform(-0.05, 0.05, (B, 12))
    eng.controller_init(_t(qi), 0.2229)
    refs = [co.ControllerGlue(qi[b], 0.2229, 0.002) for b in range(B)]
    for it in range(12):
        jv = rng.uniform(-0.6, 0.6, (B, 6))
        qf = np.zeros((B, 19))
        qf[:, 2] = 0.2229 + rng.uniform(-0.01, 0.01, B)
        qf[:, 7:] = qi + rng.uniform(-0.1, 0.1, (B, 12))
        vf = rng.uniform(-0.5, 0.5, (B, 18))
        rpy = rng.uniform(-0.2, 0.2, (B, 3))
        if it == 7:
            qf[1, 7 + 4] = 1.45  # > 80 deg on a hip-pitch joint
        st = eng.controller_update_state(_t(jv), _t(qf), _t(vf), _t(rpy))
        xf = rng.uniform(-5, 20, (B, 24, N))
        xr = rng.uniform(-1, 1, (B, 12, N + 1))
        pva = rng.uniform(-0.3, 0.3, (B, 3, 3, 4))
        wi = eng.controller_wbc_inputs(_t(xf), _t(xr), _t(pva), st["v"])
        tau = rng.uniform(-3, 3, (B, 12))
        qd = rng.uniform(-1, 1, (B, 19))
        vd = rng.uniform(-1, 1, (B, 18))
        vs = rng.uniform(-20, 20, (B, 12))
        if it == 5:
            tau[2, 7] = 8.5      # torque limit
            vs[4, 3] = -51.0     # joint velocity limit
        if it == 7:
            vs[3, 0] = 60.0
            tau[3, 1] = -9.0     # both: the torque flag wins (security_check order)
        rs = eng.controller_result(_t(tau), _t(qd), _t(vd), _t(qf), _t(vs))
        torch.cuda.synchronize()
        g = {k: v.cpu().numpy() for d in (st, wi, rs) for k, v in d.items()}
        for b in range(B):
            r = refs[b]
            oRh, oTh = r.update_state(jv[b], qf[b], vf[b], rpy[b])
            assert np.allclose(g["q"][b], r.q[:, 0], rtol=1e-13, atol=1e-15), (it, b)
            assert np.array_equal(g["v"][b], r.v[:, 0])
            assert np.allclose(g["h_v"][b], r.h_v[:6, 0], rtol=1e-13, atol=1e-15)
            assert np.array_equal(g["v_ref"][b], r.v_ref[:6, 0])
            assert np.allclose(g["oRh_oTh"][b, :9].reshape(3, 3), oRh, rtol=1e-13, atol=1e-15)
            assert np.allclose(g["oRh_oTh"][b, 9:], oTh[:, 0], rtol=1e-13, atol=1e-15)
            xw, qw, bv = r.wbc_inputs(xf[b], xr[b], oRh, oTh, pva[b, 0], pva[b, 1], pva[b, 2])
            assert np.allclose(g["x_f_wbc"][b], xw, rtol=1e-14, atol=0)
            assert np.array_equal(g["f_cmd"][b], xw[12:])
            assert np.array_equal(g["q_wbc"][b], qw[:, 0])
            assert np.array_equal(g["b_v"][b], bv[:, 0])
            assert np.allclose(g["feet_cmd"][0, b], r.feet_p_cmd, rtol=1e-12, atol=1e-15)
            assert np.allclose(g["feet_cmd"][1, b], r.feet_v_cmd, rtol=1e-12, atol=1e-15)
            assert np.allclose(g["feet_cmd"][2, b], r.feet_a_cmd, rtol=1e-12, atol=1e-14)
            P, D, q_des, v_des, t8 = r.result(tau[b], qd[b], vd[b], qf[b], vs[b])
            assert g["error_flag"][b] == r.error_flag, (it, b)
            for i, ref in enumerate((P, D, q_des, v_des, t8)):
                assert np.array_equal(g["result"][b, i], ref), (it, b, i)
    assert [r.error_flag for r in refs] == [0, 1, 3, 3, 2, 0, 0]


DEFAULT_CFG = dict(dt_wbc=0.002, dt_mpc=0.02, k_mpc=10, T_gait=0.32, T_mpc=0.32, N_gait=20, h_ref=0.2229)
# other timings than the reference's yaml: 1 kHz WBC, MPC every 20th iteration, a 0.40 s gait on a 0.24 s horizon
ALT_CFG = dict(dt_wbc=0.001, dt_mpc=0.02, k_mpc=20, T_gait=0.40, T_mpc=0.24, N_gait=26, h_ref=0.21)


def test_closed_loop_other_timings(oracle_mod):
    _closed_loop(oracle_mod, "sync", True, ALT_CFG, 70)


def test_closed_loop_three_gait_periods(oracle_mod):
    """480 iterations (three gait periods, 48 MPC solves per robot): the footstep table has rolled through every phase three times
    and every foot has flown three times when the comparison with the chained oracles ends."""
    _closed_loop(oracle_mod, "sync", True, DEFAULT_CFG, 480, B=4)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "separate"])
@pytest.mark.parametrize("mode", ["sync", "async_lag0", "async_lag3"])
def test_closed_loop_matches_chained_oracles(oracle_mod, mode, fused):
    _closed_loop(oracle_mod, mode, fused, DEFAULT_CFG, 160 if (mode == "sync" and fused) else 45)


def _shift_last_available(res, gait, n_steps):
    """scripts/MPC_Wrapper.py:89-102 restated: roll the force rows one horizon step, refill the last column from the gait."""
    res[12:(12 + n_steps), :] = np.roll(res[12:(12 + n_steps), :], -1, axis=1)
    pt = 0
    while np.any(gait[pt, :]):
        pt += 1
    if not np.array_equal(gait[0, :], gait[pt - 1, :]):
        F = 9.81 * 2.5 / np.sum(gait[pt - 1, :])
        res[12:, n_steps - 1] = 0.0
        for i in range(4):
            if gait[pt - 1, i] == 1:
                res[12 + 3 * i + 2, n_steps - 1] = F


def test_async_closed_loop_on_the_quad_wbc_kernel(oracle_mod):
    """The asynchronous mode at a batch where Controller_batch picks wbc_kernel's full mode by itself (one quad per robot:
    the loop's stream owns too few SIMDs for wbc16_kernel's batch / 4 wavefronts, Controller.py `wbc_set_lanes(4)`; every batch
    above 128 with the default loop_cus): the closed loop against the chained oracles on a spread of instances -- first and last
    rows of a wavefront's sixteen quads, both sides of a wavefront boundary, the last instance."""
    _closed_loop(oracle_mod, "async_lag3", True, DEFAULT_CFG, 45, B=160, check=(0, 1, 15, 16, 63, 64, 131, 159), want_lanes=4)


def test_closed_loop_with_the_thread_form_of_control_pre(oracle_mod, monkeypatch):
    """qrw_control_pre's one-thread-per-robot form (QRW_PRE_QUAD=0, an A/B knob read per call) in the closed loop against the
    chained oracles: the knob's other position stands under the oracle too, not only under the quad form."""
    monkeypatch.setenv("QRW_PRE_QUAD", "0")
    _closed_loop(oracle_mod, "sync", True, DEFAULT_CFG, 45)


def _closed_loop(oracle_mod, mode, fused, cfg, iters, B=5, check=None, want_lanes=None):
    """Controller_batch (planners -> MPC every 10th -> glue -> WBC -> result, all on the device) against the same chain
    made of the CPU oracles, with the measurements fed back from the oracle's own desired joint state.  The
    asynchronous modes run the MPC on its own compute-unit-masked stream and adopt a result a fixed number of
    iterations after it was issued (lag 0 = the synchronous semantics).  check: the instances compared with the oracle
    (default all); the others are fed back their own device results."""
    import torch
    import controller_oracle as co
    from Controller import Controller_batch

    k_mpc, n_steps = cfg["k_mpc"], int(round(cfg["T_mpc"] / cfg["dt_mpc"]))
    lag = {"sync": 0, "async_lag0": 0, "async_lag3": 3}[mode]
    rng = np.random.default_rng(3)
    ctl = Controller_batch(B, Q_INIT, multiprocessing=(mode != "sync"), mpc_lag=lag, fused=fused, groups=1, **cfg)
    if want_lanes is not None:
        assert ctl.wbc_lanes == want_lanes
    check = tuple(range(B)) if check is None else tuple(check)
    glue = {b: co.ControllerGlue(Q_INIT, cfg["h_ref"], cfg["dt_wbc"]) for b in check}
    plan = {b: oracle_mod.Planner(dt_mpc=cfg["dt_mpc"], dt_wbc=cfg["dt_wbc"], T_gait=cfg["T_gait"], T_mpc=cfg["T_mpc"],
                                  N_gait=cfg["N_gait"], k_mpc=k_mpc, h_ref=cfg["h_ref"]) for b in check}
    mpc = {b: oracle_mod.MPC(cfg["dt_mpc"], n_steps, cfg["T_gait"], cfg["N_gait"]) for b in check}
    wbc = {b: oracle_mod.WbcController(cfg["dt_wbc"]) for b in check}
    first = np.zeros((24, n_steps))
    first[2, 0] = cfg["h_ref"]
    first[12:, 0] = [0.0, 0.0, 8.0] * 4
    not_first = [False] * B
    issued = [[] for _ in range(B)]     # (iteration issued, result) per instance, not delivered yet
    # scripts/MPC_Wrapper.py:70-71: what get_latest_result returns until a solve delivers (asynchronous mode) -- and what
    # MPC_Wrapper.solve keeps shifting (:89-102)
    last_available = [first.copy() for _ in range(B)]
    vref = rng.uniform(-0.4, 0.4, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.0])
    qf = np.zeros((B, 19))
    qf[:, 2], qf[:, 6], qf[:, 7:] = cfg["h_ref"], 1.0, Q_INIT
    vf = np.zeros((B, 18))
    worst = 0.0
    for k in range(iters):
        rpy = rng.uniform(-0.02, 0.02, (B, 3))
        vf[:, :6] = vref + rng.uniform(-0.05, 0.05, (B, 6))
        vs = vf[:, 6:].copy()
        res = ctl.compute(_t(vref), _t(qf), _t(vf), _t(rpy), _t(vs))
        torch.cuda.synchronize()
        got = ctl._res["result"].cpu().numpy()
        flags = ctl.error_flag.cpu().numpy()
        nq, nv = qf.copy(), vf.copy()
        nq[:, 7:], nv[:, 6:] = got[:, 2], got[:, 3]  # instances outside `check`: their own device targets
        for b in check:
            g = glue[b]
            oRh, oTh = g.update_state(vref[b], qf[b], vf[b], rpy[b])
            plan[b].step(k, g.q[:7, 0], g.h_v[:6, 0], g.v_ref[:6, 0], 0)
            xref, (fsteps, _, _), cgait = plan[b].xref(), plan[b].footsteps(), plan[b].gaits()[1]
            if k % k_mpc == 0:
                mpc[b].run(k, xref, fsteps)
                issued[b].append((k, mpc[b].get_latest_result().copy()))
                if mode != "sync" and k > 2:  # MPC_Wrapper.solve bookkeeping (:89-102), dead code in the synchronous mode
                    _shift_last_available(last_available[b], cgait, n_steps)
            if not_first[b]:  # get_latest_result (:106-126): the first call returns the default without looking
                for k0, res_ in list(issued[b]):
                    if k >= k0 + lag:
                        last_available[b] = res_
                        issued[b].remove((k0, res_))
            not_first[b] = True
            x_f_mpc = last_available[b]
            pos, vel, acc, _, _ = plan[b].feet()
            xw, qw, bv = g.wbc_inputs(x_f_mpc, xref, oRh, oTh, pos, vel, acc)
            wbc[b].compute(qw, bv, xw[12:], cgait[0, :], g.feet_p_cmd, g.feet_v_cmd, g.feet_a_cmd)
            P, D, q_des, v_des, t8 = g.result(wbc[b].tau_ff, wbc[b].qdes, wbc[b].vdes[:, 0], qf[b], vs[b])
            assert flags[b] == g.error_flag == 0, (k, b)
            for i, (ref, tol) in enumerate(((P, 0), (D, 0), (q_des, 1e-4), (v_des, 1e-4), (t8, 1e-4))):
                err = np.max(np.abs(got[b, i] - ref)) / max(1.0, np.max(np.abs(ref)))
                worst = max(worst, err)
                assert err <= tol, (k, b, i, err, "gpu", got[b, i].tolist(), "oracle", np.asarray(ref).tolist())
            nq[b, 7:], nv[b, 6:] = q_des, v_des  # perfect tracking of the oracle's targets
        qf, vf = nq, nv
    st = ctl.stats()
    assert np.all(st["mpc"]["status"] == 1)
    ctl.stop_parallel_loop()
    print("closed loop (%s) worst relative deviation %.2e" % (mode, worst))


def test_async_polling_mode_runs():
    """The production asynchronous mode adopts a result when its event has completed (timing dependent, like the
    reference's shared-flag polling): check it runs, stays in the nominal regime and ends up using MPC results."""
    import torch
    from Controller import Controller_batch

    B = 64
    ctl = Controller_batch(B, Q_INIT, multiprocessing=True)
    vref = _t(np.tile(np.array([0.3, 0.0, 0, 0, 0, 0.1]), (B, 1)))
    qf = np.zeros((B, 19))
    qf[:, 2], qf[:, 6], qf[:, 7:] = 0.2229, 1.0, Q_INIT
    qf, vf = _t(qf), _t(np.zeros((B, 18)))
    rpy, vs = _t(np.zeros((B, 3))), _t(np.zeros((B, 12)))
    for k in range(60):
        r = ctl.compute(vref, qf, vf, rpy, vs)
        qf[:, 7:].copy_(r.q_des)
        vf[:, 6:].copy_(r.v_des)
    torch.cuda.synchronize()
    assert ctl._adopted is not None and ctl._n_issued == 6
    assert int((ctl.error_flag != 0).sum().item()) == 0
    assert bool(torch.isfinite(ctl._res["result"]).all())
    ctl.stop_parallel_loop()


def test_async_free_running_startup_keeps_snapshots_intact():
    """Free-running asynchronous mode (mpc_lag=None) with a first solve that cannot finish in time (the MPC stream is
    held back): solves 0..2 are still queued when solve 3 is issued, so the loop stream must wait for solve n-3 before
    it overwrites that solve's snapshot / output buffers.  Each solve's inputs are cloned at issue and compared with
    what the solve was handed; every solve must end `solved` and the loop must end on an adopted MPC result."""
    import torch
    from Controller import Controller_batch

    B = 256
    with torch.cuda.stream(torch.cuda.Stream()):
        ctl = Controller_batch(B, Q_INIT, multiprocessing=True)
        with torch.cuda.stream(ctl._s_mpc.torch):
            torch.cuda._sleep(int(2.0e8))  # ~0.1 s at ~2 GHz: longer than the 50 loop iterations below take to issue
        vref = _t(np.tile(np.array([0.3, 0.0, 0, 0, 0, 0.1]), (B, 1)))
        qf = np.zeros((B, 19))
        qf[:, 2], qf[:, 6], qf[:, 7:] = 0.2229, 1.0, Q_INIT
        qf, vf = _t(qf), _t(np.zeros((B, 18)))
        rpy, vs = _t(np.zeros((B, 3))), _t(np.zeros((B, 12)))
        handed, orig_solve = [], ctl._b.mpc_solve

        def spy(xs, fs, k, out=None):
            handed.append((xs, fs, k))
            return orig_solve(xs, fs, k, out=out)

        ctl._b.mpc_solve = spy
        clones = []
        for k in range(50):
            r = ctl.compute(vref, qf, vf, rpy, vs)
            if k % 10 == 0:
                clones.append((ctl._pre["xref"].clone(), ctl._pre["fsteps"].clone()))
            qf[:, 7:].copy_(r.q_des)
            vf[:, 6:].copy_(r.v_des)
        torch.cuda.synchronize()
        assert ctl._n_issued == 5 and len(handed) == 5
        # the last three solves' snapshot buffers still hold exactly what the planners produced at their issue
        for n in (2, 3, 4):
            xs, fs, k = handed[n]
            assert k == 10 * n and torch.equal(xs, clones[n][0]) and torch.equal(fs, clones[n][1]), n
        st = ctl.stats()["mpc"]
        assert (st["status"] == 1).all()
        r = ctl.compute(vref, qf, vf, rpy, vs)  # k = 50: issues solve 5; solves 0..4 have finished, its poll adopts solve 4
        torch.cuda.synchronize()
        assert ctl._adopted == 4 % 3 and [n for n, _ in ctl._pending] == [5]
        assert int((ctl.error_flag != 0).sum().item()) == 0
        assert bool(torch.isfinite(ctl._res["result"]).all())
        ctl.stop_parallel_loop()


def test_control_pre_without_mpc_inputs_leaves_the_rest_unchanged():
    """qrw_control_pre on an iteration that does not solve (d_fsteps / d_gait NULL, x_f_mpc given): fsteps and gait are not
    written, of xref only column 0 and horizon step 1 are, and everything the WBC step reads is what the full call gives."""
    import torch

    import qrw_hip

    B, N = 70, 16
    rng = np.random.default_rng(21)
    engs = [qrw_hip.Batch(B, N) for _ in range(2)]
    for e in engs:
        e.planner_init()
        e.controller_init(_t(np.tile(Q_INIT, (B, 1))))
    vref = _t(rng.uniform(-0.4, 0.4, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.0]))
    qf = np.zeros((B, 19))
    qf[:, 2], qf[:, 6], qf[:, 7:] = 0.2229, 1.0, Q_INIT
    x_f = np.zeros((B, 24, N))
    x_f[:, 2, :], x_f[:, 14::3, :] = 0.2229, 6.0
    outs = [None, None]
    for k in range(1, 25):  # includes k = 10, 20: the lazy call is legal at any k, the caller decides
        vf = np.zeros((B, 18))
        vf[:, :6] = vref.cpu().numpy() + rng.uniform(-0.05, 0.05, (B, 6))
        rpy = rng.uniform(-0.02, 0.02, (B, 3))
        args = (k, vref, _t(qf), _t(vf), _t(rpy), 0)
        outs[0] = engs[0].control_pre(*args, x_f_mpc=_t(x_f), out=outs[0])
        if outs[1] is not None:
            outs[1]["fsteps"].fill_(-7.0)
            outs[1]["gait"].fill_(-7.0)
            outs[1]["xref"][:, :, 2:].fill_(-7.0)
        outs[1] = engs[1].control_pre(*args, x_f_mpc=_t(x_f), out=outs[1], mpc_inputs=False)
        torch.cuda.synchronize()
        full, lazy = outs
        for key in ("q", "v", "h_v", "v_ref", "oRh_oTh", "target", "feet_pva", "contacts", "x_f_wbc", "q_wbc", "b_v", "f_cmd", "feet_cmd"):
            assert torch.equal(full[key], lazy[key]), (k, key)
        assert torch.equal(full["xref"][:, :, :2], lazy["xref"][:, :, :2])
        if k > 1:
            assert bool((lazy["fsteps"] == -7.0).all()) and bool((lazy["gait"] == -7.0).all()) and bool((lazy["xref"][:, :, 2:] == -7.0).all())
        # the planner's own copy of the footstep table (item 14, [N_gait][12]): rows 0 and 1 are kept current by the lazy call
        # (row 1 is what updateNewContact takes at a gait change), the other rows wait for the next call with the MPC's inputs
        for b in (0, B // 2, B - 1):
            tf, tl = (e.planner_get(14, 12 * 20, b).reshape(20, 12) for e in engs)
            assert np.array_equal(tf[:2], tl[:2]), (k, b)
    # one call WITH the MPC's inputs brings the whole table (state copy and fsteps output) back to what the full sequence holds
    args = (25, vref, _t(qf), _t(np.zeros((B, 18))), _t(np.zeros((B, 3))), 0)
    outs[0] = engs[0].control_pre(*args, x_f_mpc=_t(x_f), out=outs[0])
    outs[1] = engs[1].control_pre(*args, x_f_mpc=_t(x_f), out=outs[1])
    torch.cuda.synchronize()
    assert torch.equal(outs[0]["fsteps"], outs[1]["fsteps"]) and torch.equal(outs[0]["xref"], outs[1]["xref"])
    assert torch.equal(outs[0]["gait"], outs[1]["gait"])
    for b in (0, B // 2, B - 1):
        assert np.array_equal(engs[0].planner_get(14, 12 * 20, b), engs[1].planner_get(14, 12 * 20, b)), b


def test_bound_iteration_equals_the_two_separate_calls():
    """qrw_iteration_bind / qrw_iteration_step (an iteration that does not solve on buffers bound once: one foreign call) against
    qrw_control_pre (without the MPC's inputs) + qrw_wbc_compute_result with every pointer passed per call: every output of both
    launches bit for bit over 24 iterations, with a per-robot joystick-code tensor, the MPC result alternating between two
    buffers (the loop adopts results into different ones), and a re-bind half-way (other input tensors)."""
    import torch

    import qrw_hip

    B, N = 70, 16
    rng = np.random.default_rng(33)
    engs = [qrw_hip.Batch(B, N) for _ in range(2)]
    for e in engs:
        e.planner_init()
        e.controller_init(_t(np.tile(Q_INIT, (B, 1))))
    vref = _t(rng.uniform(-0.4, 0.4, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.0]))
    code = torch.zeros((B,), dtype=torch.int32, device="cuda")
    x_fs = []
    for _ in range(2):
        x = np.zeros((B, 24, N))
        x[:, 2, :], x[:, 14::3, :] = 0.2229, 6.0 + rng.uniform(-0.5, 0.5, (B, 4, 1)).repeat(N, axis=2)
        x_fs.append(_t(x))
    qf_np = np.zeros((B, 19))
    qf_np[:, 2], qf_np[:, 6], qf_np[:, 7:] = 0.2229, 1.0, Q_INIT
    # the bound side's fixed input tensors (two sets: the second half of the test binds again)
    sets = [tuple(torch.zeros(shape, dtype=torch.float64, device="cuda") for shape in ((B, 19), (B, 18), (B, 3), (B, 12))) for _ in range(2)]
    pre = [None, None]
    post = [None, None]
    # one ordinary call on each side creates the output buffers (k = 1: does not solve)
    for i, e in enumerate(engs):
        qf, vf, rpy, vs = sets[0]
        qf.copy_(_t(qf_np))
        pre[i] = e.control_pre(1, vref, qf, vf, rpy, code, x_f_mpc=x_fs[0], mpc_inputs=False)
        fc = pre[i]["feet_cmd"]
        post[i] = e.wbc_compute_result(pre[i]["q_wbc"], pre[i]["b_v"], pre[i]["f_cmd"], pre[i]["contacts"], fc[0], fc[1], fc[2], qf, vs)
    step = None
    for k in range(2, 26):
        qf, vf, rpy, vs = sets[0 if k < 14 else 1]
        qf.copy_(_t(qf_np + np.pad(rng.uniform(-0.02, 0.02, (B, 12)), ((0, 0), (7, 0)))))
        vf.zero_()
        vf[:, :6] = vref + _t(rng.uniform(-0.05, 0.05, (B, 6)))
        rpy.copy_(_t(rng.uniform(-0.02, 0.02, (B, 3))))
        vs.copy_(_t(rng.uniform(-1, 1, (B, 12))))
        if k == 12:
            code[::3] = 2  # some robots are told to change gait
        x_f = x_fs[k & 1]
        # side 0: the two calls, every pointer per call
        pre[0] = engs[0].control_pre(k, vref, qf, vf, rpy, code, x_f_mpc=x_f, out=pre[0], mpc_inputs=False)
        fc = pre[0]["feet_cmd"]
        post[0] = engs[0].wbc_compute_result(pre[0]["q_wbc"], pre[0]["b_v"], pre[0]["f_cmd"], pre[0]["contacts"], fc[0], fc[1], fc[2],
                                             qf, vs, out=post[0])
        # side 1: bound once per set of input tensors, then one foreign call per iteration
        if step is None or k == 14:
            step = engs[1].bind_iteration(pre[1], post[1], (vref, qf, vf, rpy, vs, code))
        step(k, x_f)
        torch.cuda.synchronize()
        for key in ("q", "v", "h_v", "v_ref", "oRh_oTh", "target", "feet_pva", "contacts", "x_f_wbc", "q_wbc", "b_v", "f_cmd", "feet_cmd"):
            assert torch.equal(pre[0][key], pre[1][key]), (k, key)
        assert torch.equal(pre[0]["xref"][:, :, :2], pre[1]["xref"][:, :, :2])
        for key in ("tau_ff", "qdes", "vdes", "f_with_delta", "ddq_res", "feet", "result", "error_flag"):
            assert torch.equal(post[0][key], post[1][key]), (k, key)
    with pytest.raises(qrw_hip.QrwError):  # a handle without bound buffers refuses the step
        fresh = qrw_hip.Batch(B, N)
        fresh.planner_init()
        qrw_hip._check(fresh._lib.qrw_iteration_step(fresh._handle, 3, x_fs[0].data_ptr(), None), "qrw_iteration_step")


def test_two_bound_iterations_of_one_handle_keep_their_own_buffers():
    """the library keeps ONE binding per handle, Batch.bind_iteration returns independent-looking callables.  An earlier
    callable must not run on the buffers of a later bind: each callable re-installs its own binding when another bind of the handle
    came in between -- and a tensor OBJECT that changes storage (`t.data = ...`) is noticed by Controller_batch's bound path.
    Two callables of one handle bound to DIFFERENT input tensors, stepped alternately, against a second handle stepped with every
    pointer per call."""
    import torch

    import qrw_hip
    from Controller import Controller_batch

    B, N = 40, 16
    rng = np.random.default_rng(44)
    a, ref = qrw_hip.Batch(B, N), qrw_hip.Batch(B, N)
    for e in (a, ref):
        e.planner_init()
        e.controller_init(_t(np.tile(Q_INIT, (B, 1))))
    vref = _t(rng.uniform(-0.3, 0.3, (B, 6)) * np.array([1.0, 0.5, 0, 0, 0, 1.0]))
    x = np.zeros((B, 24, N))
    x[:, 2, :], x[:, 14::3, :] = 0.2229, 6.0
    x_f = _t(x)
    qf_np = np.zeros((B, 19))
    qf_np[:, 2], qf_np[:, 6], qf_np[:, 7:] = 0.2229, 1.0, Q_INIT

    def inputs(seed):
        r = np.random.default_rng(seed)
        qf = _t(qf_np + np.pad(r.uniform(-0.02, 0.02, (B, 12)), ((0, 0), (7, 0))))
        vf = torch.zeros((B, 18), dtype=torch.float64, device="cuda")
        vf[:, :6] = vref
        return qf, vf, _t(r.uniform(-0.02, 0.02, (B, 3))), _t(r.uniform(-1, 1, (B, 12)))

    sets = [inputs(1), inputs(2)]
    pre0 = a.control_pre(1, vref, *sets[0][:3], 0, x_f_mpc=x_f, mpc_inputs=False)
    fc = pre0["feet_cmd"]
    post0 = a.wbc_compute_result(pre0["q_wbc"], pre0["b_v"], pre0["f_cmd"], pre0["contacts"], fc[0], fc[1], fc[2], sets[0][0], sets[0][3])
    rp = ref.control_pre(1, vref, *sets[0][:3], 0, x_f_mpc=x_f, mpc_inputs=False)
    fc = rp["feet_cmd"]
    rw = ref.wbc_compute_result(rp["q_wbc"], rp["b_v"], rp["f_cmd"], rp["contacts"], fc[0], fc[1], fc[2], sets[0][0], sets[0][3])
    steps = [a.bind_iteration(pre0, post0, (vref, *sets[i][:3], sets[i][3], 0)) for i in range(2)]  # the second bind replaces the first in the library
    for k in range(2, 12):
        i = (k // 2) & 1  # 2,3 -> set 1; 4,5 -> set 0; ...: each callable is used after the other one has been
        steps[i](k, x_f)
        rp = ref.control_pre(k, vref, *sets[i][:3], 0, x_f_mpc=x_f, out=rp, mpc_inputs=False)
        fc = rp["feet_cmd"]
        rw = ref.wbc_compute_result(rp["q_wbc"], rp["b_v"], rp["f_cmd"], rp["contacts"], fc[0], fc[1], fc[2], sets[i][0], sets[i][3], out=rw)
        torch.cuda.synchronize()
        assert torch.equal(post0["result"], rw["result"]) and torch.equal(pre0["q_wbc"], rp["q_wbc"]), k
    # Controller_batch's bound path: the same tensor object on other storage must be re-bound, not run on the stale address
    c1, c2 = Controller_batch(B, Q_INIT), Controller_batch(B, Q_INIT)
    qf, vf, rpy, vs = inputs(3)
    qf2 = qf.clone()
    for k in range(6):
        if k == 4:  # from here on the caller's q tensor lives elsewhere (and the old storage holds garbage)
            old = qf.data
            qf.data = qf.data.clone()
            old.fill_(float("nan"))
        r1 = c1.compute(vref, qf, vf, rpy, vs)
        r2 = c2.compute(vref, qf2, vf, rpy, vs)
        torch.cuda.synchronize()
        assert torch.equal(r1.q_des, r2.q_des) and torch.equal(r1.tau_ff, r2.tau_ff), k
        assert bool(torch.isfinite(r1.tau_ff).all())


@pytest.mark.parametrize("mode", ["sync", "async_lag2"])
def test_stream_groups_controller_equals_the_single_handle(mode):
    """Controller_batch(..., groups=2) (Controller_groups: the fleet as two independent stream groups, opt-in) against the
    single handle: the robots are independent, so Result and error flag must be
    equal bit for bit over a closed loop that crosses several MPC iterations -- through compute() (joined every iteration)
    and through compute_group() on the groups' own streams (never joined).  Synchronous mode and the asynchronous MPC mode
    with a deterministic adoption lag."""
    import torch
    from Controller import Controller_batch, Controller_groups

    B, iters = 64, 27
    kw = dict(multiprocessing=(mode != "sync"), mpc_lag=(None if mode == "sync" else 2))
    rng = np.random.default_rng(5)
    qi = Q_INIT + rng.uniform(-0.03, 0.03, (B, 12))
    vref = _t(rng.uniform(-0.4, 0.4, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.0]))

    def run(ctl, free):
        qf = torch.zeros((B, 19), dtype=torch.float64, device="cuda")
        qf[:, 2], qf[:, 6] = 0.2229, 1.0
        qf[:, 7:] = _t(qi)
        vf = torch.zeros((B, 18), dtype=torch.float64, device="cuda")
        vf[:, :6] = vref
        rpy = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        vs = torch.zeros((B, 12), dtype=torch.float64, device="cuda")
        hist = []
        with torch.cuda.stream(torch.cuda.Stream()):
            for it in range(iters):
                if free:
                    for g in range(ctl.G):
                        sl = ctl.slice_of(g)
                        with torch.cuda.stream(ctl.stream_of(g)):
                            r = ctl.compute_group(g, vref[sl], qf[sl], vf[sl], rpy[sl], vs[sl])
                            qf[sl, 7:].copy_(r.q_des)
                            vf[sl, 6:].copy_(r.v_des)
                    for g in range(ctl.G):
                        ctl.stream_of(g).synchronize()
                    hist.append(ctl._fleet_result.clone())
                else:
                    r = ctl.compute(vref, qf, vf, rpy, vs)
                    qf[:, 7:].copy_(r.q_des)
                    vf[:, 6:].copy_(r.v_des)
                    torch.cuda.current_stream().synchronize()
                    hist.append(torch.stack([r.P, r.D, r.q_des, r.v_des, r.tau_ff], 1).clone())
            flag = ctl.error_flag.clone()
        ctl.stop_parallel_loop()
        return torch.stack(hist), flag

    one = Controller_batch(B, qi, groups=1, **kw)
    assert not isinstance(one, Controller_groups)
    ref, ref_flag = run(one, False)
    two = Controller_batch(B, qi, groups=2, **kw)
    assert isinstance(two, Controller_groups) and not isinstance(Controller_batch(8, Q_INIT), Controller_groups)
    got, flag = run(two, False)
    assert torch.equal(got, ref) and torch.equal(flag, ref_flag)
    got, flag = run(Controller_batch(B, qi, groups=2, **kw), True)
    assert torch.equal(got, ref) and torch.equal(flag, ref_flag)
    with pytest.raises(Exception):
        Controller_batch(7, Q_INIT, groups=2)


@pytest.mark.parametrize("free,defaults", [(False, False), (True, False), (False, True)], ids=["joined", "free", "for_deadline"])
def test_staggered_stream_groups_equal_single_handles_started_late(free, defaults):
    """Controller_batch(..., groups=2, stagger=True): group 1 starts k_mpc / 2 fleet ticks late, so the two groups' MPC solves fall
    on different ticks (the reference solves on k % k_mpc == 0 of the robot's own clock, scripts/Controller.py:246-253).  Every
    robot must see exactly a single-handle controller started that many ticks later: Result and error flag of group g at fleet
    tick t equal, bit for bit, those of a single handle over the group's robots at its tick t - lag_of(g); before its start a
    group's robots are commanded to hold q_init (P 3, D 0.2, zero v_des / tau_ff).  Joined (compute) and never joined
    (compute_group on the groups' streams).  for_deadline: what Controller_batch.for_deadline builds for 2048 robots and the
    reference's 2 ms slot is exactly that object; WITHOUT arguments a fleet of any size is one handle (the
    staggered form changes what half the robots do, so it is never chosen silently)."""
    import torch
    from Controller import Controller_batch, Controller_groups

    B, iters, k_mpc = (2048, 23, 10) if defaults else (32, 27, 10)
    rng = np.random.default_rng(11)
    qi = Q_INIT + rng.uniform(-0.03, 0.03, (B, 12))
    vref = _t(rng.uniform(-0.4, 0.4, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.0]))
    code = torch.zeros((B,), dtype=torch.int32, device="cuda")  # a per-robot joystick code tensor goes through both entry points

    def state(n, q0):
        qf = torch.zeros((n, 19), dtype=torch.float64, device="cuda")
        qf[:, 2], qf[:, 6] = 0.2229, 1.0
        qf[:, 7:] = _t(q0)
        vf = torch.zeros((n, 18), dtype=torch.float64, device="cuda")
        return qf, vf, torch.zeros((n, 3), dtype=torch.float64, device="cuda"), torch.zeros((n, 12), dtype=torch.float64, device="cuda")

    if defaults:
        ctl = Controller_batch.for_deadline(B, qi)
        assert ctl.deadline == 0.002
        for n in (2046, 2048, 4096):
            assert not isinstance(Controller_batch(n, Q_INIT), Controller_groups)       # no arguments: one handle at every size
        assert not Controller_batch(B, Q_INIT, groups=2).stagger                         # explicit groups: not staggered unless asked
        small = Controller_batch.for_deadline(512, Q_INIT)                               # fits the slot synchronously
        assert not isinstance(small, Controller_groups) and not small.multiprocessing
        mp = Controller_batch.for_deadline(4096, Q_INIT)                                 # only the asynchronous mode fits
        assert not isinstance(mp, Controller_groups) and mp.multiprocessing
        mp.stop_parallel_loop()
    else:
        ctl = Controller_batch(B, qi, groups=2, stagger=True, k_mpc=k_mpc)
    assert ctl.lag == (0, k_mpc // 2) and ctl.lag_of(1) == k_mpc // 2
    assert isinstance(ctl, Controller_groups) and ctl.G == 2 and ctl.stagger and ctl._delay == [0, k_mpc // 2]
    qf, vf, rpy, vs = state(B, qi)
    vf[:, :6] = vref
    hist = []
    with torch.cuda.stream(torch.cuda.Stream()):
        for t in range(iters):
            if free:
                for g in range(2):
                    sl = ctl.slice_of(g)
                    with torch.cuda.stream(ctl.stream_of(g)):
                        r = ctl.compute_group(g, vref[sl], qf[sl], vf[sl], rpy[sl], vs[sl], code[sl])
                        if ctl.group_started(g):
                            qf[sl, 7:].copy_(r.q_des)
                            vf[sl, 6:].copy_(r.v_des)
                for g in range(2):
                    ctl.stream_of(g).synchronize()
            else:
                r = ctl.compute(vref, qf, vf, rpy, vs, code)
                for g in range(2):
                    if ctl.group_started(g):
                        sl = ctl.slice_of(g)
                        qf[sl, 7:].copy_(r.q_des[sl])
                        vf[sl, 6:].copy_(r.v_des[sl])
                torch.cuda.current_stream().synchronize()
            hist.append(ctl._fleet_result.clone())
        flags = ctl.error_flag.clone()
    assert ctl.k == iters
    for g in range(2):
        sl, d = ctl.slice_of(g), ctl._delay[g]
        one = Controller_batch(B // 2, qi[sl], k_mpc=k_mpc)
        q1, v1, rpy1, vs1 = state(B // 2, qi[sl])
        v1[:, :6] = vref[sl]
        for t in range(iters):
            if t < d:  # not started: hold q_init with the controller's own PD gains
                hold = hist[t][sl]
                assert bool((hold[:, 0] == 3.0).all()) and bool((hold[:, 1] == 0.2).all()) and torch.equal(hold[:, 2], _t(qi[sl]))
                assert float(hold[:, 3:].abs().max()) == 0.0
                continue
            r = one.compute(vref[sl], q1, v1, rpy1, vs1, code[sl])
            q1[:, 7:].copy_(r.q_des)
            v1[:, 6:].copy_(r.v_des)
            torch.cuda.synchronize()
            assert torch.equal(hist[t][sl], one._res["result"]), (g, t)
        assert torch.equal(flags[sl], one.error_flag)


@pytest.mark.parametrize("cfg", ["default", "alt", "short_period"])
def test_control_pre_quad_form_equals_the_thread_form_through_gait_changes(cfg, monkeypatch):
    """qrw_control_pre's quad kernel (one quad per robot: touch-downs of the footstep table computed wave-together, mask arithmetic
    for getPhaseDuration) against the one-thread-per-robot form (QRW_PRE_QUAD=0, the kernel the planner parity tests compare with
    the CPU oracle) over 450 iterations in which every robot changes gait several times (pacing, bounding, walk, trot, static and
    the 4-beat gait of code 5, per-robot codes), with and without the MPC's inputs: every output and the planner's persistent state."""
    import torch

    import qrw_hip

    if cfg == "default":
        B, N, kw, k_mpc = 21, 16, dict(), 10
    elif cfg == "alt":  # 0.40 s gait on a 0.24 s horizon, 26 gait rows, 1 kHz loop
        B, N, kw, k_mpc = 21, 12, dict(N_gait=26, T_gait=0.40, dt_wbc=0.001), 20
    else:  # a 0.08 s gait period in 24 rows: up to six touch-downs per foot in the table (the quad form's row-wise fallback)
        B, N, kw, k_mpc = 21, 4, dict(N_gait=24, T_gait=0.08), 10
    Ng = kw.get("N_gait", 20)
    rng = np.random.default_rng(77)
    engs = [qrw_hip.Batch(B, N, **kw) for _ in range(2)]
    for e in engs:
        e.planner_init(k_mpc=k_mpc)
        e.controller_init(_t(np.tile(Q_INIT, (B, 1))))
    vref = rng.uniform(-0.5, 0.5, (B, 6)) * np.array([1.5, 0.8, 0, 0, 0, 1.2])
    vref[0, 5] = 0.0
    qf = np.zeros((B, 19))
    qf[:, 2], qf[:, 6], qf[:, 7:] = 0.2229, 1.0, Q_INIT
    x_f = np.zeros((B, 24, N))
    x_f[:, 2, :], x_f[:, 14::3, :] = 0.2229, 6.0
    outs = [None, None]
    keys = ("q", "v", "h_v", "v_ref", "oRh_oTh", "xref", "fsteps", "gait", "target", "feet_pva", "contacts", "x_f_wbc", "q_wbc", "b_v",
            "f_cmd", "feet_cmd")
    worst = 0.0
    for k in range(450):
        vf = np.zeros((B, 18))
        vf[:, :6] = vref + rng.uniform(-0.05, 0.05, (B, 6))
        rpy = rng.uniform(-0.02, 0.02, (B, 3))
        qf[:, 0:2] += 0.002 * vref[:, 0:2]
        code = np.zeros(B, np.int32)
        if k in (37, 111, 180, 262, 333, 401):  # Gait::changeGait acts at k % k_mpc == 0 of a new phase; codes at any k are legal
            code[:] = rng.integers(0, 6, B)
        if k == 222:
            code[:] = 4  # everybody static at once, then away from it again at 262
        solve = (k % k_mpc == 0)
        for e, eng in enumerate(engs):
            monkeypatch.setenv("QRW_PRE_QUAD", "1" if e == 0 else "0")
            c = _t(code).to(torch.int32) if code.any() else 0
            outs[e] = eng.control_pre(k, _t(vref), _t(qf), _t(vf), _t(rpy), c, x_f_mpc=(None if solve else _t(x_f)), out=outs[e],
                                      mpc_inputs=solve)
        torch.cuda.synchronize()
        for key in keys:
            if not solve and key in ("fsteps", "gait"):
                continue
            if solve and key in ("x_f_wbc", "q_wbc", "b_v", "f_cmd", "feet_cmd"):
                continue
            a, b_ = outs[0][key].cpu().numpy(), outs[1][key].cpu().numpy()
            if key == "xref" and not solve:
                a, b_ = a[:, :, :2], b_[:, :, :2]
            # (the two forms contract their fused multiply-adds differently; the swing polynomials amplify the last bit, as in the
            # oracle comparison of tests/test_gpu_planner.py -- and the trajectory state is a recursion over hundreds of iterations;
            # a wrong row or phase would show as 1e-2 and more)
            loose = key in ("feet_pva", "feet_cmd")
            assert np.allclose(a, b_, rtol=(1e-7 if loose else 1e-11), atol=(1e-6 if loose else 1e-13)), (k, key, np.abs(a - b_).max())
            if not loose:
                worst = max(worst, float(np.abs(a - b_).max()))
        if k % 25 == 3 or solve:
            for b in (0, 7, B - 1):
                for item, n in ((0, 4 * Ng), (1, 4 * Ng), (2, 4 * Ng), (3, 1), (4, 1), (5, 1), (7, 12), (8, 12), (9, 12), (10, 12), (11, 12),
                                (12, 4), (13, 4), (15, 12), (17, 12), (18, 2)):
                    sa, sb = engs[0].planner_get(item, n, b), engs[1].planner_get(item, n, b)
                    assert np.allclose(sa, sb, rtol=1e-7, atol=(1e-6 if item in (9, 10, 11) else 1e-12)), (k, b, item)
                ta, tb = (e.planner_get(14, 12 * Ng, b).reshape(Ng, 12) for e in engs)
                rows = Ng if solve else 2  # (rows >= 2 of the quad form's table wait for the next solving iteration)
                assert np.allclose(ta[:rows], tb[:rows], rtol=1e-11, atol=1e-13), (k, b)
    assert worst < 1e-9


@pytest.mark.parametrize("mode", ["sync", "async_lag3"])
def test_closed_loop_does_not_depend_on_uninitialised_buffers(oracle_mod, monkeypatch, mode):
    """Every torch.empty of the run (the controller's intermediate and output buffers on the device) comes back filled with NaN
    (floating point) or -1 (integers) instead of whatever the allocator hands out: the closed loop must still match the chained
    CPU oracles -- no kernel may read a buffer entry before something has written it (0 x leftover is only 0 while the leftover
    is finite; see the LDS case of round 4, profiles/r4_lds_poison.txt)."""
    import torch

    real_empty = torch.empty

    def poisoned_empty(*args, **kwargs):
        t = real_empty(*args, **kwargs)
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64):
            t.fill_(-1)
        return t

    monkeypatch.setattr(torch, "empty", poisoned_empty)
    _closed_loop(oracle_mod, mode, True, DEFAULT_CFG, 45)


def test_deadline_monitor_reports_the_overrun_and_for_deadline_avoids_it():
    """the reference runs its loop in a slot of dt_wbc = 2 ms (/root/reference/src/config_solo12.yaml:6;
    scripts/Controller.py:246-253: every k_mpc-th iteration carries the MPC solve).  The default object (one handle, synchronous)
    at 4096 robots cannot keep that: with deadline=dt_wbc it must say so -- once, as a RuntimeWarning that names the mode to use,
    and in ctl.overrun --; the object Controller_batch.for_deadline builds for the same fleet (asynchronous MPC) paced at 2 ms
    must not overrun."""
    import time
    import warnings

    import torch
    from Controller import Controller_batch, recommended_mode

    B = 4096
    rng = np.random.default_rng(5)
    vref = _t(rng.uniform(-0.3, 0.3, (B, 6)) * np.array([1.0, 0.5, 0, 0, 0, 1.0]))

    def loop(ctl, iters, paced):
        qf = torch.zeros((B, 19), dtype=torch.float64, device="cuda")
        qf[:, 2], qf[:, 6] = 0.2229, 1.0
        qf[:, 7:] = _t(np.broadcast_to(Q_INIT, (B, 12)).copy())
        vf = torch.zeros((B, 18), dtype=torch.float64, device="cuda")
        vf[:, :6] = vref
        rpy = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        vs = torch.zeros((B, 12), dtype=torch.float64, device="cuda")
        nxt = time.perf_counter()
        for _ in range(iters):
            r = ctl.compute(vref, qf, vf, rpy, vs)
            qf[:, 7:].copy_(r.q_des)
            vf[:, 6:].copy_(r.v_des)
            if paced:
                torch.cuda.current_stream().synchronize()
                nxt += 0.002
                while time.perf_counter() < nxt:
                    pass
        return ctl.deadline_flush()

    with torch.cuda.stream(torch.cuda.Stream()):
        ctl = Controller_batch(B, Q_INIT, deadline=0.002)
        assert ctl.deadline == 0.002 and ctl.overrun is None and not ctl.multiprocessing
        with pytest.warns(RuntimeWarning, match=r"deadline 2\.00 ms.*multiprocessing=True") as rec:
            worst = loop(ctl, 51, paced=False)
        assert len([w for w in rec if issubclass(w.category, RuntimeWarning)]) == 1  # said once
        assert ctl.overrun is not None and ctl.overrun[0] == worst > 0.002 and ctl.overrun[1] % ctl.k_mpc == 0
        assert recommended_mode(B, 0.002) == dict(multiprocessing=True)
        ok = Controller_batch.for_deadline(B, Q_INIT)
        assert ok.multiprocessing and ok.deadline == 0.002
        with torch.cuda.stream(ok.loop_stream):
            loop(ok, 30, paced=True)  # start-up: the first two solves set the QP up and start cold
            ok.worst_iteration, ok.overrun = 0.0, None
            with warnings.catch_warnings():
                warnings.simplefilter("error", RuntimeWarning)
                worst_ok = loop(ok, 60, paced=True)
        print("4096 robots, 2 ms slot: synchronous worst iteration %.2f ms (reported), asynchronous worst %.2f ms" % (worst * 1e3, worst_ok * 1e3))
        assert ok.overrun is None and 0.0 < worst_ok < 0.002
        ok.stop_parallel_loop()
