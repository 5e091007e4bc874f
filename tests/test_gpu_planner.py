"""GPU parity tests of the batched planners (planner_kernel.hip through the C ABI) vs the oracle, and of the
fully device-resident control step planners -> MPC -> WBC."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(rng, B, k):
    q7 = np.zeros((B, 7))
    q7[:, 2] = 0.2229 + rng.uniform(-0.01, 0.01, B)
    quat = np.stack([rng.uniform(-0.03, 0.03, B), rng.uniform(-0.03, 0.03, B), rng.uniform(-0.2, 0.2, B), np.ones(B)], 1)
    q7[:, 3:] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    q7[:, :2] = rng.uniform(-1, 1, (B, 2))
    return q7


def test_planner_step_matches_oracle(oracle_mod):
    import qrw_hip

    B, N = 9, 16
    rng = np.random.default_rng(4)
    eng = qrw_hip.Batch(B, N)
    eng.planner_init()
    refs = [oracle_mod.Planner() for _ in range(B)]
    vref = rng.uniform(-0.5, 0.5, (B, 6)) * np.array([2, 1, 0, 0, 0, 1.4])
    vref[0, 5] = 0.0  # exercises the vref(5) == 0 branches
    for k in range(0, 260):
        q7 = _inputs(rng, B, k)
        hv = vref + rng.uniform(-0.1, 0.1, (B, 6))
        code = np.zeros(B, np.int32)
        if k == 57:
            code[:] = [0, 1, 2, 3, 4, 5, 0, 2, 1]
        if k == 140:
            code[:] = 3
        # per-instance codes go through the host path one value at a time: use the scalar for uniform steps
        if len(set(code)) == 1:
            o = eng.planner_call_host(2 | 4 | 8 | 16, k=k, k_footsteps=10 - k % 10, refresh=(k % 10 == 0 and k != 0),
                                      q7=q7, v6=hv, vref6=vref, code=int(code[0]))
        else:
            import torch

            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
            od = eng.planner_step(k, t(q7), t(hv), t(vref), t(code))
            torch.cuda.synchronize()
            o = {kk: v.cpu().numpy() for kk, v in od.items()}
        for b in range(B):
            r = refs[b]
            r.step(k, q7[b], hv[b], vref[b], int(code[b]))
            f, tg, otg = r.footsteps()
            pos, vel, acc, t0s, tsw = r.feet()
            assert np.array_equal(o["gait"][b], r.gaits()[1]), (k, b)
            assert np.allclose(o["xref"][b], r.xref(), rtol=1e-12, atol=1e-13), (k, b)
            assert np.allclose(o["fsteps"][b], f, rtol=1e-11, atol=1e-13), (k, b)
            assert np.allclose(o["target"][b], otg, rtol=1e-11, atol=1e-13), (k, b)
            assert np.allclose(o["feet_pva"][b, 0], pos, rtol=1e-9, atol=1e-11), (k, b)
            assert np.allclose(o["feet_pva"][b, 1], vel, rtol=1e-9, atol=1e-10), (k, b)
            assert np.allclose(o["feet_pva"][b, 2], acc, rtol=1e-9, atol=1e-8), (k, b)
        if k % 50 == 7:
            b = 5
            past, cur, des = refs[b].gaits()
            Ng = 20
            assert np.array_equal(eng.planner_get(0, Ng * 4, b).reshape(Ng, 4), past)
            assert np.array_equal(eng.planner_get(2, Ng * 4, b).reshape(Ng, 4), des)
            fl = refs[b].flags()
            assert bool(eng.planner_get(3, 1, b)[0]) == fl["new_phase"] and bool(eng.planner_get(4, 1, b)[0]) == fl["is_static"]


def test_planner_long_run_with_random_gait_changes(oracle_mod):
    """1 500 iterations (three seconds of robot time) of the batched planners on the device path with per-robot joystick codes
    changing at random times (pacing, bounding, trot, static, walk and back), reference velocities re-drawn on the way and the
    yaw-rate == 0 branches kept in: gait matrices, reference states, footstep table, targets and foot trajectories against the
    oracle on every iteration (src/Gait.cpp:184-260, src/FootstepPlanner.cpp:51-221, src/FootTrajectoryGenerator.cpp:41-151,
    src/StatePlanner.cpp:21-61)."""
    import torch

    import qrw_hip

    B, N = 8, 16
    rng = np.random.default_rng(2026)
    eng = qrw_hip.Batch(B, N)
    eng.planner_init()
    refs = [oracle_mod.Planner() for _ in range(B)]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    vref = rng.uniform(-0.5, 0.5, (B, 6)) * np.array([2, 1, 0, 0, 0, 1.4])
    vref[0, 5] = 0.0
    out = None
    worst = [0.0, 0.0, 0.0]
    next_change = rng.integers(30, 200, B)
    for k in range(1500):
        if k % 400 == 399:
            vref = rng.uniform(-0.5, 0.5, (B, 6)) * np.array([2, 1, 0, 0, 0, 1.4])
            vref[1, 5] = 0.0
        q7 = _inputs(rng, B, k)
        hv = vref + rng.uniform(-0.1, 0.1, (B, 6))
        code = np.zeros(B, np.int32)
        for b in range(B):
            if k == next_change[b]:
                code[b] = rng.integers(1, 6)
                next_change[b] = k + rng.integers(25, 260)
        out = eng.planner_step(k, t(q7), t(hv), t(vref), t(code), out=out)
        torch.cuda.synchronize()
        o = {kk: v.cpu().numpy() for kk, v in out.items()}
        for b in range(B):
            r = refs[b]
            r.step(k, q7[b], hv[b], vref[b], int(code[b]))
            f, tg, otg = r.footsteps()
            pos, vel, acc, t0s, tsw = r.feet()
            assert np.array_equal(o["gait"][b], r.gaits()[1]), (k, b)
            assert np.array_equal(o["contacts"][b], r.gaits()[1][0]), (k, b)
            assert np.allclose(o["xref"][b], r.xref(), rtol=1e-12, atol=1e-13), (k, b)
            assert np.allclose(o["fsteps"][b], f, rtol=1e-11, atol=1e-13), (k, b)
            assert np.allclose(o["target"][b], otg, rtol=1e-11, atol=1e-13), (k, b)
            # (the polynomials' high powers cancel: the absolute error scales with the largest value of the trajectory, hundreds of
            # m/s^2 in the accelerations right after a gait change shortens a swing)
            for d, ref, tol in ((0, pos, 1e-7), (1, vel, 1e-7), (2, acc, 1e-6)):  # measured worst: 2.4e-9, 2.7e-9, 3.3e-8
                err = np.abs(o["feet_pva"][b, d] - ref).max() / max(1.0, np.abs(ref).max())
                worst[d] = max(worst[d], err)
                assert err < tol, (k, b, d, err)
    print("planner long run: worst scaled deviation of foot position / velocity / acceleration: %.1e %.1e %.1e" % tuple(worst))


def test_planner_dropin_classes_match_oracle(oracle_mod):
    """The reference-named classes called exactly as scripts/Controller.py:119-137,222-241 calls them."""
    import libquadruped_reactive_walking as lqrw

    dt_mpc, dt_wbc, T, N_gait, k_mpc, h_ref = 0.02, 0.002, 0.32, 20, 10, 0.2229
    shoulders = np.zeros((3, 4))
    shoulders[0, :] = [0.1946, 0.1946, -0.1946, -0.1946]
    shoulders[1, :] = [0.14695, -0.14695, 0.14695, -0.14695]
    fsteps_init = shoulders.copy()
    statePlanner = lqrw.StatePlanner()
    statePlanner.initialize(dt_mpc, T, h_ref)
    gait = lqrw.Gait()
    gait.initialize(dt_mpc, T, T, N_gait)
    footstepPlanner = lqrw.FootstepPlanner()
    footstepPlanner.initialize(dt_mpc, dt_wbc, T, h_ref, shoulders.copy(), gait, N_gait)
    ftg = lqrw.FootTrajectoryGenerator()
    ftg.initialize(0.05, 0.07, fsteps_init.copy(), shoulders.copy(), dt_wbc, k_mpc, gait)
    ref = oracle_mod.Planner(dt_mpc, dt_wbc, T, T, N_gait, k_mpc, h_ref, shoulders, 0.05, 0.07, fsteps_init, shoulders)
    with pytest.raises(ValueError):
        lqrw.Gait().initialize(dt_mpc, T, T, 10)
    rng = np.random.default_rng(8)
    q = np.zeros((19, 1))
    q[2, 0], q[6, 0] = h_ref, 1.0
    vref = np.array([0.5, 0.1, 0, 0, 0, 0.3])
    # FootstepPlanner::getRz (python/gepadd.cpp:123): zero but (2,2) = 1 before the first update (src/FootstepPlanner.cpp:10,48)
    assert np.array_equal(footstepPlanner.getRz(), np.diag([0.0, 0.0, 1.0])) and np.array_equal(ref.Rz(), np.diag([0.0, 0.0, 1.0]))
    assert gait.setGait(np.ones((N_gait, 4))) is False  # Gait::setGait prints and returns false (src/Gait.cpp:262-269)
    for k in range(0, 45):
        h_v = (vref + rng.uniform(-0.05, 0.05, 6)).reshape(6, 1)
        code = 2 if k == 23 else 0
        # a base that yaws (and rolls / pitches a little): quaternion x, y, z, w of rpy = (0.02, -0.03, 0.015 k)
        cr, sr, cp, sp, cy, sy = np.cos(0.01), np.sin(0.01), np.cos(-0.015), np.sin(-0.015), np.cos(0.0075 * k), np.sin(0.0075 * k)
        q[3:7, 0] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
        q[0, 0], q[1, 0] = 0.01 * k, -0.004 * k
        gait.updateGait(k, k_mpc, q[0:7, 0:1], code)
        o_target = footstepPlanner.updateFootsteps(k % k_mpc == 0 and k != 0, int(k_mpc - k % k_mpc), q[0:7, 0:1],
                                                   h_v.copy(), vref)
        ftg.update(k, o_target)
        statePlanner.computeReferenceStates(q[0:7, 0:1], h_v.copy(), vref.reshape(6, 1), 0.0)
        ref.step(k, q[:7, 0], h_v[:, 0], vref, code)
        f, tg, otg = ref.footsteps()
        pos, vel, acc, _, _ = ref.feet()
        assert statePlanner.getReferenceStates().shape == (12, 17) and footstepPlanner.getFootsteps().shape == (20, 12)
        assert np.allclose(statePlanner.getReferenceStates(), ref.xref(), rtol=1e-12, atol=1e-13)
        assert np.allclose(footstepPlanner.getFootsteps(), f, rtol=1e-11, atol=1e-13) and np.allclose(o_target, otg, atol=1e-13)
        assert np.array_equal(gait.getCurrentGait(), ref.gaits()[1]) and gait.getIsStatic() == ref.flags()["is_static"]
        Rz = footstepPlanner.getRz()
        assert Rz.shape == (3, 3) and np.allclose(Rz, ref.Rz(), rtol=0, atol=1e-14) and abs(Rz[1, 0] - np.sin(0.015 * k)) < 1e-12
        assert np.allclose(ftg.getFootPosition(), pos, atol=1e-11) and np.allclose(ftg.getFootVelocity(), vel, atol=1e-10)
        assert np.allclose(ftg.getFootAcceleration(), acc, atol=1e-8)


def test_device_resident_control_step_matches_oracle(oracle_mod):
    """planners -> MPC -> WBC with every intermediate left in HBM, against the same pipeline on the oracle."""
    import torch

    import qrw_hip

    B, N, k_mpc = 6, 16, 10
    rng = np.random.default_rng(15)
    eng = qrw_hip.Batch(B, N)
    eng.planner_init(k_mpc=k_mpc)
    planners = [oracle_mod.Planner() for _ in range(B)]
    mpcs = [oracle_mod.MPC(0.02, N, 0.32, 20) for _ in range(B)]
    wbcs = [oracle_mod.WbcController(0.002) for _ in range(B)]
    vref = rng.uniform(-0.4, 0.4, (B, 6)) * np.array([2, 1, 0, 0, 0, 1.5])
    q7 = np.zeros((B, 7))
    q7[:, 2], q7[:, 6] = 0.2229, 1.0
    q19 = np.zeros((B, 19))
    q19[:, 2], q19[:, 6] = 0.2229, 1.0
    q19[:, 7:] = [0.0, 0.7, -1.4, 0.0, 0.7, -1.4, 0.0, -0.7, 1.4, 0.0, -0.7, 1.4]
    dq = np.zeros((B, 18))
    dq[:, :6] = vref
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    mpc_out = None
    x_f = [np.zeros((24, N)) for _ in range(B)]
    for k in range(0, 32):
        hv = vref + rng.uniform(-0.05, 0.05, (B, 6))
        plan = eng.planner_step(k, t(q7), t(hv), t(vref), 0)
        if k % k_mpc == 0:
            mpc_out = eng.mpc_solve(plan["xref"], plan["fsteps"], k)
        contacts = plan["contacts"]
        assert torch.equal(contacts, plan["gait"][:, 0, :])
        f_cmd = mpc_out[:, 12:, 0].contiguous()
        # feet goals: foot trajectory outputs expressed relative to the base height (scripts/Controller.py:294-296, yaw = 0)
        pg = plan["feet_pva"][:, 0].clone()
        pg[:, 2, :] -= 0.2229
        w = eng.wbc_compute(t(q19), t(dq), f_cmd, contacts, pg.contiguous(), plan["feet_pva"][:, 1].contiguous(),
                            plan["feet_pva"][:, 2].contiguous())
        torch.cuda.synchronize()
        tau = w["tau_ff"].cpu().numpy()
        for b in range(B):
            planners[b].step(k, q7[b], hv[b], vref[b], 0)
            if k % k_mpc == 0:
                f, _, _ = planners[b].footsteps()
                mpcs[b].run(k, planners[b].xref(), f)
                x_f[b] = mpcs[b].get_latest_result()
            pos, vel, acc, _, _ = planners[b].feet()
            pgb = pos.copy()
            pgb[2] -= 0.2229
            wbcs[b].compute(q19[b], dq[b], x_f[b][12:, 0], planners[b].gaits()[1][0], pgb, vel, acc)
            assert np.allclose(tau[b], wbcs[b].tau_ff, rtol=1e-4, atol=1e-6), (k, b)
    assert np.isfinite(tau).all()


def test_planner_on_wild_inputs_matches_oracle(oracle_mod):
    """The planners far outside the joystick's range (src/Gait.cpp, FootstepPlanner.cpp, StatePlanner.cpp, FootTrajectoryGenerator.cpp
    take any q / v / vref / code): arbitrary yaw over +-pi with roll / pitch up to +-0.5 rad, base positions metres from the origin,
    reference velocities up to 3 m/s and 3 rad/s (the Raibert terms run into their +-L clamp, src/FootstepPlanner.cpp:160-176), measured
    velocities a metre per second away from the reference, re-drawn every 40 iterations, yaw rate exactly 0 on some robots, and gait
    codes arriving in bursts (a new code on consecutive iterations, codes during a transition, code 0).  Every output of every
    iteration against the oracle, 24 robots x 600 iterations."""
    import torch

    import qrw_hip

    B, N = 24, 16
    rng = np.random.default_rng(777)
    eng = qrw_hip.Batch(B, N)
    eng.planner_init()
    refs = [oracle_mod.Planner() for _ in range(B)]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()

    def draw_vref():
        v = rng.uniform(-1, 1, (B, 6)) * np.array([3.0, 1.5, 0, 0, 0, 3.0])
        v[::5, 5] = 0.0
        v[3::7, :2] = 0.0
        return v

    vref = draw_vref()
    out = None
    worst = [0.0] * 6
    burst = np.zeros(B, np.int64)
    for k in range(600):
        if k % 40 == 39:
            vref = draw_vref()
        q7 = np.zeros((B, 7))
        q7[:, :2] = rng.uniform(-5, 5, (B, 2))
        q7[:, 2] = 0.2229 + rng.uniform(-0.05, 0.05, B)
        yaw, pitch, roll = rng.uniform(-np.pi, np.pi, B), rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)
        cy, sy, cp, sp, cr, sr = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(roll / 2), np.sin(roll / 2)
        q7[:, 3] = sr * cp * cy - cr * sp * sy
        q7[:, 4] = cr * sp * cy + sr * cp * sy
        q7[:, 5] = cr * cp * sy - sr * sp * cy
        q7[:, 6] = cr * cp * cy + sr * sp * sy
        hv = vref + rng.uniform(-1.0, 1.0, (B, 6)) * np.array([1, 1, 0.3, 0.5, 0.5, 1])
        code = np.zeros(B, np.int32)
        for b in range(B):
            if burst[b] > 0:  # a burst: a code on every iteration for a while
                code[b] = rng.integers(0, 6)
                burst[b] -= 1
            elif rng.random() < 0.02:
                code[b] = rng.integers(1, 6)
                burst[b] = rng.integers(0, 4)
        out = eng.planner_step(k, t(q7), t(hv), t(vref), t(code), out=out)
        torch.cuda.synchronize()
        o = {kk: v.cpu().numpy() for kk, v in out.items()}
        for b in range(B):
            r = refs[b]
            r.step(k, q7[b], hv[b], vref[b], int(code[b]))
            f, tg, otg = r.footsteps()
            pos, vel, acc, t0s, tsw = r.feet()
            assert np.array_equal(o["gait"][b], r.gaits()[1]), (k, b)
            assert np.array_equal(o["contacts"][b], r.gaits()[1][0]), (k, b)
            for i, (name, got, ref) in enumerate((("xref", o["xref"][b], r.xref()), ("fsteps", o["fsteps"][b], f), ("target", o["target"][b], otg),
                                                  ("pos", o["feet_pva"][b, 0], pos), ("vel", o["feet_pva"][b, 1], vel), ("acc", o["feet_pva"][b, 2], acc))):
                err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
                worst[i] = max(worst[i], err)
                assert err < (1e-9 if i < 3 else 1e-6), (k, b, name, err)
    print("planner on wild inputs: worst scaled deviation of xref / fsteps / target / foot position / velocity / acceleration: " + " ".join("%.1e" % w for w in worst))
