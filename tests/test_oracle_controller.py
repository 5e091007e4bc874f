"""CPU checks of oracle/controller_oracle.py (the glue restatement) on properties the reference's formulas imply."""
import numpy as np

import controller_oracle as co


def test_quaternion_and_rotation_agree():
    rng = np.random.default_rng(0)
    for _ in range(20):
        r, p, y = rng.uniform(-1, 1, 3)
        x, yq, z, w = co.euler_to_quaternion([r, p, y])
        R = np.array([[1 - 2 * (yq * yq + z * z), 2 * (x * yq - z * w), 2 * (x * z + yq * w)],
                      [2 * (x * yq + z * w), 1 - 2 * (x * x + z * z), 2 * (yq * z - x * w)],
                      [2 * (x * z - yq * w), 2 * (yq * z + x * w), 1 - 2 * (x * x + yq * yq)]])
        assert np.allclose(R, co.euler_to_rotation(r, p, y), atol=1e-14)


def test_update_state_integrates_reference_velocity():
    g = co.ControllerGlue(np.zeros(12), 0.2229, 0.002)
    jv = np.array([0.5, 0.0, 0, 0, 0, 0.0])
    qf, vf = np.zeros(19), np.zeros(18)
    for _ in range(500):
        oRh, oTh = g.update_state(jv, qf, vf, np.zeros(3))
    assert np.allclose(oTh[:, 0], [0.5, 0.0, 0.0], atol=1e-12)  # 1 s at 0.5 m/s straight ahead
    jv[5] = np.pi / 2
    for _ in range(500):
        oRh, oTh = g.update_state(jv, qf, vf, np.zeros(3))
    assert np.isclose(g.yaw_estim, np.pi / 2)
    assert np.allclose(oRh[:2, :2], [[0, -1], [1, 0]], atol=1e-12)


def test_foot_command_of_a_fixed_world_point():
    """A foot fixed in the world while the base translates at v_ref has base-frame velocity -v_ref."""
    g = co.ControllerGlue(np.zeros(12), 0.2229, 0.002)
    jv = np.array([0.3, -0.1, 0, 0, 0, 0.0])
    oRh, oTh = g.update_state(jv, np.zeros(19), np.zeros(18), np.zeros(3))
    pos = np.array([[0.19, 0.19, -0.19, -0.19], [0.15, -0.15, 0.15, -0.15], [0.0, 0.0, 0.0, 0.0]])
    xw, qw, bv = g.wbc_inputs(np.zeros((24, 16)), np.zeros((12, 17)), oRh, oTh, pos, np.zeros((3, 4)), np.zeros((3, 4)))
    assert np.allclose(g.feet_v_cmd, -np.tile(jv[:3, None], (1, 4)))
    assert np.allclose(g.feet_p_cmd[2], -0.2229)
    assert qw[6, 0] == 1.0 and qw[2, 0] == 0.2229 and np.array_equal(bv[:6, 0], jv)


def test_security_flags_are_sticky_and_ordered():
    g = co.ControllerGlue(np.zeros(12), 0.2229, 0.002)
    ok = g.result(np.ones(12), np.zeros(19), np.zeros(18), np.zeros(19), np.zeros(12))
    assert g.error_flag == 0 and np.allclose(ok[4], 0.8)
    tau = np.ones(12)
    tau[3] = 8.01
    vs = np.zeros(12)
    vs[0] = 51
    P, D, qd, vd, t = g.result(tau, np.zeros(19), np.zeros(18), np.zeros(19), vs)
    assert g.error_flag == 3 and np.all(P == 0) and np.all(D == 0.1) and np.all(t == 0)
    g.result(np.ones(12), np.zeros(19), np.zeros(18), np.zeros(19), np.zeros(12))
    assert g.error_flag == 3 and g.error
