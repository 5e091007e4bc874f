"""One-gait-period closed-loop map (monodromy) of the UNCONSTRAINED MPC law of the reference's two-stance trot scenario
(/root/reference/scripts/test_mpc.py:136-190): every call solves the QP of src/MPC.cpp with the current state in column 0
of xref and takes the first predicted state as the next current state (:186-187).  While no friction-cone row is active
the optimum is an equality-constrained QP, the first predicted state is an affine function of the current state, and the
product of the 16 Jacobians of one gait period decides whether a deviation from the reference grows or dies out.

Built from tests/test_oracle_mpc.dense_qp (an independent numpy assembly per SURVEY.md Appendix A.1) and one dense KKT
solve per call: no oracle code, no ADMM.  Shared by the CPU test (oracle) and the GPU test (HIP path)."""
import numpy as np

import trot_kat

N = trot_kat.N
# src/MPC.cpp:330 (the weights in use) and :329 (the set left in the comment above it), float literals promoted to double
W_330 = np.array([2.0, 2.0, 20.0, 0.25, 0.25, 10.0] + [np.float32(0.2)] * 3 + [0.0, 0.0, np.float32(0.3)], dtype=np.float64)
W_329 = np.array([2.0, 2.0, 20.0, 2.0, 2.0, 10.0] + [np.float32(0.2)] * 3 + [0.0, 0.0, 10.0], dtype=np.float64)


def first_predicted_state(xref, fsteps, w):
    """x_f[:12, 0] of the equality-constrained QP (dynamics rows + force-enable rows of the swing feet), weights w."""
    from test_oracle_mpc import dense_qp

    A, lo, up, Pd = dense_qp(xref, fsteps, N, first_call=False)
    Pd = np.concatenate([np.tile(w, N), Pd[12 * N:]])
    gait = fsteps[:N, 0::3] != 0
    rows = list(range(12 * N)) + [12 * N + 12 * k + 3 * f + c for k in range(N) for f in range(4) if not gait[k, f]
                                  for c in range(3)]
    Ae, be = A[rows], up[rows]
    n, me = 24 * N, len(rows)
    K = np.zeros((n + me, n + me))
    K[:n, :n] = np.diag(Pd)
    K[:n, n:] = Ae.T
    K[n:, :n] = Ae
    sol = np.linalg.solve(K, np.concatenate([np.zeros(n), be]))[:n]
    return sol[:12] + xref[:, 1]


def monodromy(w, eps=1e-7):
    """Product of the Jacobians d(first predicted state) / d(current state) over the 16 calls of one gait period, at the
    centred reference state (the lever arms of horizon step 0 depend on the current position, hence differences)."""
    plan = trot_kat.CompressedTrot()
    xc = np.zeros(12)
    xc[2] = trot_kat.H_REF
    Phi = np.eye(12)
    for _ in range(16):
        fs = plan.fsteps()
        xref = np.zeros((12, N + 1))
        xref[2, :] = trot_kat.H_REF
        xref[:, 0] = xc
        x1 = first_predicted_state(xref, fs, w)
        F = np.zeros((12, 12))
        for j in range(12):
            xr = xref.copy()
            xr[j, 0] += eps
            F[:, j] = (first_predicted_state(xr, fs, w) - x1) / eps
        Phi = F @ Phi
        plan.roll()
    return Phi


def spectral_radius(w):
    return float(np.abs(np.linalg.eigvals(monodromy(w))).max())
