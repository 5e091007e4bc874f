"""The algebra of dissect.h (a measured-slower form of the N = 32 path, kept as scripts/experiments/slower_forms.patch: the
horizon dissected around step 16, both halves swept as 16-step twisted systems, the coupling carried as fill blocks E_k
and one root solve) restated in numpy on a random symmetric positive definite block-tridiagonal system and checked
against a dense solve.  CPU only (python -m pytest scripts/experiments/dissection_algebra_check.py); the HIP code is
checked on the GPU by the patched library's self-test (qrw_selftest_sweeps -> dissect_selftest) against a dense host solve."""
import numpy as np


def test_dissected_solve_equals_dense_solve():
    rng = np.random.default_rng(1)
    N, n, ROOT = 32, 12, 16
    C = [None] + [0.3 * rng.standard_normal((n, n)) for _ in range(1, N)]  # C[k] = K[k, k-1]
    T = []
    for _ in range(N):
        M = rng.standard_normal((n, n))
        T.append(M @ M.T + 6 * np.eye(n))
    K = np.zeros((N * n, N * n))
    for k in range(N):
        K[k * n:(k + 1) * n, k * n:(k + 1) * n] = T[k]
        if k > 0:
            K[k * n:(k + 1) * n, (k - 1) * n:k * n] = C[k]
            K[(k - 1) * n:k * n, k * n:(k + 1) * n] = C[k].T
    assert np.linalg.eigvalsh(K).min() > 0
    r = rng.standard_normal((N, n))
    x_ref = np.linalg.solve(K, r.ravel()).reshape(N, n)

    inv = np.linalg.inv
    Dinv, Mneg, E, schur = {}, {}, {}, []   # Delta_k^-1, negated chain matrices (from step, to step), fill, root's Schur terms

    def up_chain(k0, k1, G=None):       # eliminate k0..k1 upwards; G = fill block K[k0, ROOT] or None
        for k in range(k0, k1 + 1):
            D = T[k] if k == k0 else T[k] + Mneg[(k - 1, k)] @ C[k].T       # T_k - N_k C_k'
            Dinv[k] = inv(D)
            Mneg[(k, k + 1)] = -C[k + 1] @ Dinv[k]                           # -N_{k+1}
            if G is not None:
                E[k] = -Dinv[k] @ G
                schur.append(G.T @ E[k])
                G = Mneg[(k, k + 1)] @ G
        return G

    def down_chain(k0, k1, G=None):
        for k in range(k0, k1 - 1, -1):
            D = T[k] if k == k0 else T[k] + Mneg[(k + 1, k)] @ C[k + 1]      # T_k - Nt_k C_{k+1}
            Dinv[k] = inv(D)
            Mneg[(k, k - 1)] = -C[k].T @ Dinv[k]                             # -Nt_{k-1}
            if G is not None:
                E[k] = -Dinv[k] @ G
                schur.append(G.T @ E[k])
                G = Mneg[(k, k - 1)] @ G
        return G

    # left half: chain A 0..7 up, chain B 15..9 down with fill G_15 = K[15,16] = C_16', half root 8
    up_chain(0, 7)
    G8 = down_chain(15, 9, C[16].T)
    Dinv[8] = inv(T[8] + Mneg[(7, 8)] @ C[8].T + Mneg[(9, 8)] @ C[9])
    E[8] = -Dinv[8] @ G8
    schur.append(G8.T @ E[8])
    # right half: chain 31..25 down, chain 17..23 up with fill G_17 = K[17,16] = C_17, half root 24
    down_chain(31, 25)
    G24 = up_chain(17, 23, C[17])
    Dinv[24] = inv(T[24] + Mneg[(25, 24)] @ C[25] + Mneg[(23, 24)] @ C[24].T)
    E[24] = -Dinv[24] @ G24
    schur.append(G24.T @ E[24])
    Dinv[ROOT] = inv(T[ROOT] + sum(schur))
    assert sorted(E) == list(range(8, 16)) + list(range(17, 25))

    # forward sweeps of both halves
    u = r.copy()
    for k in range(1, 8):
        u[k] += Mneg[(k - 1, k)] @ u[k - 1]
    for k in range(14, 8, -1):
        u[k] += Mneg[(k + 1, k)] @ u[k + 1]
    u[8] += Mneg[(7, 8)] @ u[7] + Mneg[(9, 8)] @ u[9]
    for k in range(30, 24, -1):
        u[k] += Mneg[(k + 1, k)] @ u[k + 1]
    for k in range(18, 24):
        u[k] += Mneg[(k - 1, k)] @ u[k - 1]
    u[24] += Mneg[(25, 24)] @ u[25] + Mneg[(23, 24)] @ u[23]
    # phase 1: Delta^-1 products and fill contributions; phase 2: the root; phase 3: the fill steps take x_16
    v = np.zeros_like(u)
    for k in range(N):
        if k != ROOT:
            v[k] = Dinv[k] @ u[k]
    x = np.zeros_like(u)
    x[ROOT] = Dinv[ROOT] @ (u[ROOT] + sum(E[k].T @ u[k] for k in E))
    for k in E:
        v[k] = v[k] + E[k] @ x[ROOT]
    # backward sweeps of both halves
    x[8] = v[8]
    for k in range(7, -1, -1):
        x[k] = v[k] + Mneg[(k, k + 1)].T @ x[k + 1]
    for k in range(9, 16):
        x[k] = v[k] + Mneg[(k, k - 1)].T @ x[k - 1]
    x[24] = v[24]
    for k in range(25, 32):
        x[k] = v[k] + Mneg[(k, k - 1)].T @ x[k - 1]
    for k in range(23, 16, -1):
        x[k] = v[k] + Mneg[(k, k + 1)].T @ x[k + 1]
    assert np.abs(x - x_ref).max() < 1e-12
