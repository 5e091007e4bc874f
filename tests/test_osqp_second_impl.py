"""The oracle's OSQP restatement (oracle/osqp_restate.c, C, banded solve) against a second, separately written
implementation of the same published algorithm (tests/osqp_numpy.py, dense numpy) on the two QPs of the hot path.

OSQP's answer is path-dependent: what is compared is the ITERATE SEQUENCE — iteration count, rho and
solution over warm-started solve sequences — through exactly the update calls the reference makes:
  MPC  src/MPC.cpp:541-558       setup, then osqp_update_A + osqp_update_bounds, osqp_solve
  WBC  src/QPWBC.cpp:252-270     setup, then osqp_update_P -> osqp_update_lin_cost -> upper bound -> lower bound
The QP data are assembled in numpy from the reference's formulas, independently of the oracle's C assembly.
"""
import numpy as np
import pytest

from osqp_numpy import OSQPNumpy
from test_oracle_mpc import DT, StatefulDenseQP, dense_qp
from test_oracle_wbc import G_matrix

F32 = lambda v: float(np.float32(v))  # noqa: E731  (the reference writes several settings as float literals)


@pytest.mark.parametrize("gait,seed", [("trot", 20260000), ("walk", 31007), ("bounding", 31011)])
def test_mpc_iterate_sequence_matches_second_implementation(oracle_mod, synth_mod, gait, seed):
    N = 16
    sb = synth_mod.SyntheticBatch(1, N, gaits=(gait,), seed0=seed)
    m = oracle_mod.MPC(DT, N, 0.32, 20)
    second = None
    x0 = None
    iters = []
    for s in range(6):
        d = sb.step(s, x0)
        xref, fsteps = d["xref"][0], d["fsteps"][0]
        assert m.run(s, xref, fsteps) == 0
        A, lo, up, Pd = dense_qp(xref, fsteps, N, first_call=(s == 0))
        if second is None:  # src/MPC.cpp:527-541
            second = OSQPNumpy(np.diag(Pd), np.zeros(24 * N), A, lo, up, sigma=F32(1e-6), eps_abs=F32(1e-6),
                               eps_rel=F32(1e-6), eps_prim_inf=F32(1e-5), eps_dual_inf=F32(1e-4), alpha=F32(1.6),
                               adaptive_rho=True, adaptive_rho_interval=200, adaptive_rho_tolerance=F32(5.0))
        else:  # :548-549
            second.update_A(A)
            second.update_bounds(lo, up)
        sx, _ = second.solve()
        iters.append(second.iter)
        assert (second.iter, second.status) == (m.iter, m.status), (s, second.iter, m.iter, second.status, m.status)
        assert np.isclose(second.rho, m.rho, rtol=1e-9), (s, second.rho, m.rho)
        sol = m.solution()
        assert np.abs(sx - sol).max() <= 1e-9 * max(1.0, np.abs(sol).max()), (s, np.abs(sx - sol).max())
        xi, zi, yi = m.iterates()  # scaled warm-start iterates carried to the next call
        assert np.allclose(second.x, xi, rtol=1e-7, atol=1e-9) and np.allclose(second.y, yi, rtol=1e-7, atol=1e-9)
        x0 = m.get_latest_result()[:12, 0][None]
    assert max(iters) >= 200  # the sequence crossed an adaptive-rho test, not just first-check terminations


@pytest.mark.parametrize("N,seed", [(8, 5), (12, 11)])
def test_mpc_iterate_sequence_on_arbitrary_contact_tables(oracle_mod, synth_mod, N, seed):
    """The same comparison on ARBITRARY contact tables (synth.RandomContactTables: single-stance rows, ragged tables that change
    completely between warm-started calls, large state errors -- several adaptive-rho updates per solve): the oracle's solver
    against the second implementation fed by the second, stateful assembly (StatefulDenseQP), so that neither the matrices nor
    the ADMM of the wide GPU parity tests' checker stand alone.  rho to 1e-6 here: on these inputs the adapted rho of two builds
    of one source already differs by ~1e-7 (tests/test_gpu_mpc_random_tables.py)."""
    gen = synth_mod.RandomContactTables(1, N, seed0=seed)
    m = oracle_mod.MPC(DT, N, DT * N, gen.N_gait)
    ind = StatefulDenseQP(N, gen.N_gait)
    second = None
    iters = []
    for s in range(5):
        d = gen.step(s)
        xref, fsteps = d["xref"][0], d["fsteps"][0]
        assert m.run(s, xref, fsteps) == 0
        A, lo, up, Pd = ind.call(s, xref, fsteps)
        if second is None:
            second = OSQPNumpy(np.diag(Pd), np.zeros(24 * N), A, lo, up, sigma=F32(1e-6), eps_abs=F32(1e-6),
                               eps_rel=F32(1e-6), eps_prim_inf=F32(1e-5), eps_dual_inf=F32(1e-4), alpha=F32(1.6),
                               adaptive_rho=True, adaptive_rho_interval=200, adaptive_rho_tolerance=F32(5.0))
        else:
            second.update_A(A)
            second.update_bounds(lo, up)
        sx, _ = second.solve()
        iters.append(second.iter)
        assert (second.iter, second.status) == (m.iter, m.status), (s, second.iter, m.iter, second.status, m.status)
        assert np.isclose(second.rho, m.rho, rtol=1e-6), (s, second.rho, m.rho)
        sol = m.solution()
        assert np.abs(sx - sol).max() <= 1e-7 * max(1.0, np.abs(sol).max()), (s, np.abs(sx - sol).max())
    assert max(iters) >= 400


def _wbc_qp(M, Jc, f_cmd, RNEA):
    """src/QPWBC.cpp:481-498 (compute_matrices) and :337-343 (bounds), in numpy."""
    Y = M[:6, :6]
    X = Jc[:, :6].T
    Yinv = np.linalg.pinv(Y)
    A = Yinv @ X
    gamma = Yinv @ (X @ f_cmd - RNEA)
    H = A.T @ (0.1 * np.eye(6)) @ A + 5.0 * np.eye(12)
    g = A.T @ (0.1 * np.eye(6)) @ gamma
    G = G_matrix()
    return H, g, G, -G @ f_cmd, -G @ f_cmd + 25.0, A, gamma


def test_wbc_update_sequence_matches_second_implementation(oracle_mod):
    """Eight consecutive QPWBC::run calls with changing contacts, forces and base wrench: the cost scale of call k
    is computed inside osqp_update_P with the linear cost of call k-1 still installed (src/QPWBC.cpp:258-261)."""
    rng = np.random.default_rng(5)
    qn = np.zeros(19)
    qn[6] = 1.0
    M = oracle_mod.crba(qn)
    M[:6, :6] *= np.eye(6)  # scripts/QP_WBC.py:93
    qp = oracle_mod.QPWBC()
    second = None
    contact_sets = [(1, 0, 0, 1), (1, 0, 0, 1), (0, 1, 1, 0), (1, 1, 1, 1), (1, 1, 1, 0), (0, 1, 1, 0), (1, 0, 0, 1),
                    (1, 1, 1, 1)]
    seen_active = False
    for k, cs in enumerate(contact_sets):
        q = np.zeros(19)
        q[2], q[6] = 0.2229, 1.0
        q[7:] = np.array([0.0, 0.7, -1.4, 0.0, 0.7, -1.4, 0.0, -0.7, 1.4, 0.0, -0.7, 1.4]) + rng.uniform(-0.1, 0.1, 12)
        Jc = oracle_mod.feet_jacobians(q)
        f_cmd = np.zeros(12)
        for i, c in enumerate(cs):
            if c:
                f_cmd[3 * i:3 * i + 3] = [rng.normal() * 2.0, rng.normal() * 2.0, 24.5 / sum(cs) + rng.normal()]
            else:
                Jc[3 * i:3 * i + 3] = 0.0
        if k == 4:
            f_cmd[2] = 0.4  # nearly unloaded stance foot: the f_z >= 0 / friction rows become active
            f_cmd[0] = 1.5
        RNEA = rng.normal(size=6) * np.array([2, 2, 5, 0.5, 0.5, 0.5]) + np.array([0, 0, 24.5, 0, 0, 0])
        assert qp.run(M, Jc, f_cmd, RNEA, np.zeros(4)) == 0
        H, g, G, lo, up, A, gamma = _wbc_qp(M, Jc, f_cmd, RNEA)
        if second is None:  # src/QPWBC.cpp:239-252
            second = OSQPNumpy(H, g, G, lo, up, eps_abs=F32(1e-5), eps_rel=F32(1e-5), adaptive_rho=True,
                               adaptive_rho_interval=200, adaptive_rho_tolerance=F32(5.0))
        else:  # :258-265, in the reference's order
            second.update_P(H)
            second.update_lin_cost(g)
            second.update_upper_bound(up)
            second.update_lower_bound(lo)
        dx, _ = second.solve()
        assert (second.iter, second.status) == (qp.iter, qp.status), (k, second.iter, qp.iter)
        assert np.isclose(second.rho, qp.rho, rtol=1e-9)
        f = qp.get_f_res()
        assert np.abs((dx + f_cmd) - f).max() <= 1e-9 * max(1.0, np.abs(f).max()), (k, np.abs(dx + f_cmd - f).max())
        assert np.allclose(qp.get_ddq_res(), A @ dx + gamma, rtol=1e-9, atol=1e-11)
        Gf = G @ f
        seen_active |= bool((np.abs(Gf) < 1e-3).any())
    assert seen_active  # at least one call ended on an active cone / unilateral row
