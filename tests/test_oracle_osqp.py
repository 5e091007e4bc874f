"""Checks of the OSQP-0.6-style restatement (oracle/osqp_restate.c) in isolation.

OSQP is absent from this image and the reference pins nothing at this boundary
(PARITY UNPINNED), so the solver is checked through the optimality
conditions of the QP it solves and against an independent active-set solution.
"""
import numpy as np
import pytest
import scipy.sparse as sp


def random_qp(rng, n, m, n_eq):
    Pd = rng.normal(size=(n, n))
    P = Pd @ Pd.T / n + 0.1 * np.eye(n)
    q = rng.normal(size=n)
    A = rng.normal(size=(m, n)) * (rng.uniform(size=(m, n)) < 0.5)
    x_feas = rng.normal(size=n)
    Ax = A @ x_feas
    l = Ax - rng.uniform(0.1, 1.0, m)
    u = Ax + rng.uniform(0.1, 1.0, m)
    l[:n_eq] = u[:n_eq] = Ax[:n_eq]
    l[n_eq:n_eq + 2] = -np.inf
    return P, q, A, l, u


def kkt_residuals(P, q, A, l, u, x, y):
    stat = np.abs(P @ x + q + A.T @ y).max()
    Ax = A @ x
    prim = max(np.maximum(l - Ax, 0).max(), np.maximum(Ax - u, 0).max())
    # complementarity: y+ only where Ax = u, y- only where Ax = l
    comp = max((np.maximum(y, 0) * np.where(np.isfinite(u), u - Ax, 0)).max(),
               (np.maximum(-y, 0) * np.where(np.isfinite(l), Ax - l, 0)).max())
    return stat, prim, comp


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_qp_satisfies_kkt(oracle_mod, seed):
    rng = np.random.default_rng(seed)
    P, q, A, l, u = random_qp(rng, 12, 20, 3)
    s = oracle_mod.OSQP(sp.csc_matrix(np.triu(P)), q, sp.csc_matrix(A), l, u, eps_abs=1e-9, eps_rel=1e-9,
                        adaptive_rho_interval=50, max_iter=20000)
    x, y = s.solve()
    info = s.info()
    assert info["status"] == 1, info
    stat, prim, comp = kkt_residuals(P, q, A, l, u, x, y)
    assert stat < 1e-6 and prim < 1e-6 and comp < 1e-5, (stat, prim, comp, info)


def test_warm_start_and_updates(oracle_mod):
    """update_A / update_bounds / update_P keep the iterates (warm start) and re-scale the data."""
    rng = np.random.default_rng(7)
    P, q, A, l, u = random_qp(rng, 10, 16, 2)
    As = sp.csc_matrix(A)
    s = oracle_mod.OSQP(sp.csc_matrix(np.triu(P)), q, As, l, u, eps_abs=1e-8, eps_rel=1e-8,
                        adaptive_rho_interval=25, max_iter=20000)
    x1, _ = s.solve()
    it_cold = s.info()["iter"]
    # same problem again: warm start must terminate at the first check
    s.update_A(As.data)
    s.update_bounds(l, u)
    x2, _ = s.solve()
    assert s.info()["iter"] == 25 and it_cold > 25
    assert np.allclose(x1, x2, atol=1e-6)
    # perturbed data: still a KKT point of the NEW problem
    A2 = As.copy()
    A2.data = A2.data * (1 + 0.05 * rng.normal(size=A2.data.size))
    s.update_A(A2.data)
    s.update_bounds(l - 0.1, u + 0.1)
    x3, y3 = s.solve()
    stat, prim, comp = kkt_residuals(P, q, A2.toarray(), l - 0.1, u + 0.1, x3, y3)
    assert stat < 1e-5 and prim < 1e-5, (stat, prim)


def test_ruiz_scaling_equilibrates(oracle_mod):
    rng = np.random.default_rng(3)
    P, q, A, l, u = random_qp(rng, 8, 12, 2)
    A[:, 0] *= 100.0
    A[3, :] *= 1e-2
    s = oracle_mod.OSQP(sp.csc_matrix(np.triu(P)), q, sp.csc_matrix(A), l, u)
    D, E, c = s.scaling()
    K = np.block([[c * np.diag(D) @ P @ np.diag(D), (np.diag(E) @ A @ np.diag(D)).T],
                  [np.diag(E) @ A @ np.diag(D), np.zeros((12, 12))]])
    norms = np.abs(K).max(axis=0)
    norms = norms[norms > 0]
    # ten Ruiz passes bring every KKT column norm to O(1) (exactly 1 up to the interleaved cost scaling)
    assert norms.max() / norms.min() < 3.0, norms
    raw = np.abs(np.block([[P, A.T], [A, np.zeros((12, 12))]])).max(axis=0)
    assert raw.max() / raw[raw > 0].min() > 100.0
    assert (D > 0).all() and (E > 0).all() and c > 0


def test_primal_infeasible_detected(oracle_mod):
    # x <= -1 and x >= 1
    P = sp.csc_matrix(np.array([[1.0]]))
    A = sp.csc_matrix(np.array([[1.0], [1.0]]))
    s = oracle_mod.OSQP(P, np.zeros(1), A, np.array([-1e30, 1.0]), np.array([-1.0, 1e30]))
    x, _ = s.solve()
    assert s.info()["status"] == -3 and np.isnan(x).all()
