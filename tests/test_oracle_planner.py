"""Checks of the oracle's planner restatement (oracle/planner_oracle.c <- src/Gait.cpp, src/StatePlanner.cpp,
src/FootstepPlanner.cpp, src/FootTrajectoryGenerator.cpp).  The reference has no vectors for these either:
pinned by structural properties of the algorithms."""
import numpy as np
import pytest

Q7 = np.array([0.0, 0.0, 0.2229, 0.0, 0.0, 0.0, 1.0])


def test_gait_initialisation_and_roll(oracle_mod):
    p = oracle_mod.Planner()
    past, cur, des = p.gaits()
    # Gait::initialize -> create_trot + create_gait_f (src/Gait.cpp:19-36,110-139): T_mpc = T_gait fills 16 rows
    assert np.array_equal(cur[:8], np.tile([1, 0, 0, 1], (8, 1))) and np.array_equal(cur[8:16], np.tile([0, 1, 1, 0], (8, 1)))
    assert not cur[16:].any() and not past.any()
    for k in range(0, 160):
        p.gait_update(k, Q7, 0)
        past, cur, des = p.gaits()
        rolls = k // 10 + 1
        exp_row0 = [1, 0, 0, 1] if (rolls % 16) < 8 else [0, 1, 1, 0]
        assert np.array_equal(cur[0], exp_row0), k
        assert np.count_nonzero(cur.any(axis=1)) == 16  # the horizon keeps 16 live rows
        if k % 10 == 0 and rolls % 8 == 0:
            assert p.flags()["new_phase"]
    # a joystick code swaps the desired gait; the current gait follows row by row as it rolls
    p.gait_update(161, Q7, 2)  # bounding
    assert np.array_equal(p.gaits()[2][0], [1, 1, 0, 0])
    with pytest.raises(ValueError):
        oracle_mod.Planner(N_gait=10)  # Gait::initialize throws (src/Gait.cpp:30-31)


def test_state_planner_formulas(oracle_mod, synth_mod):
    p = oracle_mod.Planner()
    for wz in (0.0, 0.35):
        v = np.array([0.1, -0.05, 0.02, 0.01, -0.02, 0.1])
        vref = np.array([0.4, 0.1, 0, 0, 0, wz])
        q = Q7.copy()
        q[3:7] = [0.02, -0.03, 0.1, 0.99]
        q[3:7] /= np.linalg.norm(q[3:7])
        p.state_compute(q, v, vref)
        x = p.xref()
        x0 = x[:, 0].copy()
        assert x0[0] == 0 and x0[1] == 0 and x0[5] == 0 and x0[2] == q[2] and np.array_equal(x0[6:], v)
        ref = synth_mod.reference_states(x0, vref, 16, 0.02)[0]
        assert np.allclose(x[:, 1:], ref[:, 1:], rtol=1e-12, atol=1e-14)


def test_footsteps_and_trajectories(oracle_mod):
    p = oracle_mod.Planner()
    v = np.array([0.3, 0.05, 0, 0, 0, 0.2])
    vref = np.array([0.4, 0.0, 0, 0, 0, 0.3])
    touchdown_err = []
    for k in range(0, 400):
        p.step(k, Q7, v, vref, 0)
        f, t, ot = p.footsteps()
        cur = p.gaits()[1]
        # fsteps rows mirror the gait: non-zero x exactly for stance feet (what src/MPC.cpp:686-701 relies on)
        assert np.array_equal(f[:, 0::3] != 0, cur > 0), k
        pos, vel, acc, t0s, tsw = p.feet()
        swing = cur[0] == 0
        assert (pos[2, swing] >= -1e-12).all() and (pos[2] <= 0.05 + 1e-9).all()
        for j in np.where(swing)[0]:
            if t0s[j] + 0.002 >= tsw[j] - 1e-9:  # last sample of a swing: the foot is at its target on the ground
                touchdown_err.append(abs(pos[2, j]))
    assert touchdown_err and max(touchdown_err) < 1e-6
    # stance feet of row 0 sit at currentFootstep, future touch-downs ahead of the shoulders when walking forward
    assert (np.abs(f[0, 0::3][cur[0] > 0]) < 0.3).all()
