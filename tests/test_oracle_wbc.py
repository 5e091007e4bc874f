"""Checks of the oracle's WBC restatement (oracle/wbc_oracle.c <- src/QPWBC.cpp, src/InvKin.cpp,
scripts/QP_WBC.py, scripts/solo12InvKin.py).  PARITY UNPINNED by the reference (no vectors):
checked through formula recomposition in numpy and an independent SLSQP solve of the box-QP."""
import numpy as np
import pytest
from scipy.optimize import minimize

MU = 0.9


def G_matrix():
    SC = np.array([[-1, 0, MU], [1, 0, MU], [0, -1, MU], [0, 1, MU], [0, 0, 1.0]])  # src/QPWBC.cpp:10-16
    G = np.zeros((20, 12))
    for i in range(4):
        G[5 * i:5 * i + 5, 3 * i:3 * i + 3] = SC
    return G


def test_qpwbc_matches_formulas_and_independent_solver(oracle_mod):
    rng = np.random.default_rng(0)
    q = np.zeros(19)
    q[2], q[6] = 0.2229, 1.0
    q[7:] = [0.0, 0.7, -1.4, 0.0, 0.7, -1.4, 0.0, -0.7, 1.4, 0.0, -0.7, 1.4]
    qn = np.zeros(19)
    qn[6] = 1.0
    M = oracle_mod.crba(qn)
    M[:6, :6] *= np.eye(6)  # scripts/QP_WBC.py:93
    Jc = oracle_mod.feet_jacobians(q)
    Jc[3:9] = 0.0  # FR, HL in swing
    qp = oracle_mod.QPWBC()
    G = G_matrix()
    for trial in range(4):
        f_cmd = np.array([0.5, -0.3, 12.0, 0, 0, 0, 0, 0, 0, -0.4, 0.2, 13.0]) + (0 if trial == 0 else rng.normal(size=12))
        f_cmd[3:9] = 0.0
        RNEA = rng.normal(size=6) * np.array([1, 1, 5, 0.3, 0.3, 0.3]) + np.array([0, 0, 24.5, 0, 0, 0])
        assert qp.run(M, Jc, f_cmd, RNEA, np.zeros(4)) == 0
        Y = M[:6, :6]
        X = Jc[:, :6].T
        Yinv = np.linalg.pinv(Y)
        A = Yinv @ X
        gamma = Yinv @ (X @ f_cmd - RNEA)
        H = A.T @ (0.1 * np.eye(6)) @ A + 5.0 * np.eye(12)
        g = A.T @ (0.1 * np.eye(6)) @ gamma
        assert np.allclose(qp.get_H(), H, rtol=1e-12, atol=1e-14)
        lo, up = -G @ f_cmd, -G @ f_cmd + 25.0
        res = minimize(lambda x: 0.5 * x @ H @ x + g @ x, np.zeros(12), jac=lambda x: H @ x + g, method="SLSQP",
                       constraints=[{"type": "ineq", "fun": lambda x: G @ x - lo, "jac": lambda x: G},
                                    {"type": "ineq", "fun": lambda x: up - G @ x, "jac": lambda x: -G}],
                       options={"ftol": 1e-14, "maxiter": 500})
        f = qp.get_f_res()
        assert qp.status == 1 and qp.iter % 25 == 0
        assert np.allclose(f - f_cmd, res.x, atol=2e-4), (trial, np.abs(f - f_cmd - res.x).max())
        assert np.allclose(qp.get_ddq_res(), A @ (f - f_cmd) + gamma, atol=1e-12)
        assert (G @ f >= -1e-4).all() and (G @ f <= 25 + 1e-4).all()


def test_wbc_compute_recomposes(oracle_mod, synth_mod):
    """scripts/QP_WBC.py:52-131 recomposed from the oracle's own pieces in numpy."""
    sb = synth_mod.SyntheticBatch(3, 16, seed0=21)
    for b in range(3):
        wbc = oracle_mod.WbcController(0.002)
        for s in range(3):
            d = sb.step(s)
            q, dq = d["q"][b], d["dq"][b]
            contacts = d["contacts"][b]
            f_cmd = np.zeros(12)
            f_cmd[2::3] = contacts * 24.5 / max(contacts.sum(), 1)
            assert wbc.compute(q, dq, f_cmd, contacts, d["pgoals"][b], d["vgoals"][b], d["agoals"][b]) == 0
            posf, vf, wf, af, Jf = oracle_mod.fixed_feet(q[7:], dq[6:])
            ik = oracle_mod.InvKin(0.002)
            ddq = np.zeros(18)
            ddq[6:] = ik.refreshAndCompute(contacts, d["pgoals"][b], d["vgoals"][b], d["agoals"][b], posf, vf, wf,
                                           af, Jf)
            assert np.allclose(wbc.qdes[7:], q[7:] + ik.get_q_step(), atol=1e-14) and np.all(wbc.qdes[:7] == 0)
            assert np.allclose(wbc.vdes[6:, 0], ik.get_dq_cmd(), atol=1e-14) and np.all(wbc.vdes[:6] == 0)
            Jc = oracle_mod.feet_jacobians(q)
            for i in range(4):
                if not contacts[i]:
                    Jc[3 * i:3 * i + 3] = 0
            ddq[:6] += wbc.ddq_res
            tau = oracle_mod.rnea(q, dq, ddq)[6:] - Jc[:, 6:].T @ wbc.f_with_delta.ravel()
            assert np.allclose(wbc.tau_ff, tau, atol=1e-12)
            assert wbc.f_with_delta.shape == (12, 1) and wbc.vdes.shape == (18, 1) and wbc.qdes.shape == (19,)
            p, e, v = wbc.feet()
            assert np.allclose(p, posf.T) and np.allclose(e, d["pgoals"][b] - posf.T) and np.allclose(v, vf.T)
            assert np.abs(wbc.f_with_delta.ravel()[np.repeat(contacts == 0, 3)]).max() < 1e-3
        assert np.array_equal(wbc.k_since_contact.ravel() > 0, contacts > 0)


def test_pinv_of_full_base_block(oracle_mod):
    """pseudoInverse<> (include/qrw/InvKin.hpp:60-66) on a non-diagonal symmetric Y."""
    rng = np.random.default_rng(2)
    q = np.zeros(19)
    q[3:7] = [0.1, -0.2, 0.05, 0.97]
    q[3:7] /= np.linalg.norm(q[3:7])
    q[7:] = rng.uniform(-0.5, 0.5, 12)
    M = oracle_mod.crba(q)  # unmasked: Y has off-diagonal blocks
    Jc = oracle_mod.feet_jacobians(q)
    qp = oracle_mod.QPWBC()
    f_cmd = np.tile([0.0, 0.0, 6.0], 4)
    RNEA = np.array([0.1, -0.2, 24.0, 0.05, 0.02, -0.01])
    qp.run(M, Jc, f_cmd, RNEA, np.zeros(4))
    A = np.linalg.pinv(M[:6, :6]) @ Jc[:, :6].T
    H = 0.1 * A.T @ A + 5.0 * np.eye(12)
    assert np.allclose(qp.get_H(), H, rtol=1e-9, atol=1e-10)
