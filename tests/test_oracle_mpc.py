"""Checks of the oracle's MPC restatement (oracle/mpc_oracle.c <- /root/reference/src/MPC.cpp).

Pins available: the analytic properties asserted by the reference's stale
scripts/test_mpc.py:54-110,136-190 (tests/trot_kat.py); the structural invariants of the QP; an independent dense
assembly of the QP from the documented equations (scripts/Documentation/Equations_MPC_22_02_2020.tex:348-616);
the optimality conditions of the solved QP.  No numeric vectors exist in the reference.
"""
import numpy as np
import pytest

DT = 0.02
MASS = np.float64(np.float32(2.50000279))
MU = np.float64(np.float32(0.9))
GI = np.array([[3.09249e-2, -8.00101e-7, 1.865287e-5], [-8.00101e-7, 5.106100e-2, 1.245813e-4],
               [1.865287e-5, 1.245813e-4, 6.939757e-2]])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def dense_qp(xref, fsteps, N, first_call=False):
    """Independent dense assembly of (A, l, u, Pdiag) per SURVEY.md Appendix A.1."""
    n, m = 24 * N, 44 * N
    A = np.zeros((m, n))
    Ad = np.eye(12)
    Ad[:6, 6:] = DT * np.eye(6)
    gait = (fsteps[:, 0::3] != 0).astype(float)
    u = np.zeros(m)
    l = np.zeros(m)
    for k in range(N):
        A[12 * k:12 * k + 12, 12 * k:12 * k + 12] = -np.eye(12)
        if k > 0:
            A[12 * k:12 * k + 12, 12 * (k - 1):12 * k] = Ad
        B = np.zeros((12, 12))
        for i in range(4):
            B[6:9, 3 * i:3 * i + 3] = DT / MASS * np.eye(3)
        c, s = np.cos(xref[5, k]), np.sin(xref[5, k])
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        Iinv = np.linalg.inv(R.T @ GI @ R)
        for i in range(4):
            if first_call:
                foot = np.array([[0.19, 0.19, -0.19, -0.19], [0.15005, -0.15005, 0.15005, -0.15005], [0, 0, 0, 0]])[:, i]
                lever = foot - xref[0:3, k]
            else:
                lever = fsteps[k, 3 * i:3 * i + 3] - (xref[0:3, k] + np.array([0, 0, -0.03]))
            B[9:12, 3 * i:3 * i + 3] = DT * (Iinv @ skew(lever))
        A[12 * k:12 * k + 12, 12 * (N + k):12 * (N + k) + 12] = B
        for i in range(4):
            for cc in range(3):
                A[12 * N + 12 * k + 3 * i + cc, 12 * (N + k) + 3 * i + cc] = 1.0 - gait[k, i]
            r0 = 24 * N + 20 * k + 5 * i
            c0 = 12 * (N + k) + 3 * i
            A[r0 + 0, c0 + 0], A[r0 + 0, c0 + 2] = 1.0, -MU
            A[r0 + 1, c0 + 0], A[r0 + 1, c0 + 2] = -1.0, -MU
            A[r0 + 2, c0 + 1], A[r0 + 2, c0 + 2] = 1.0, -MU
            A[r0 + 3, c0 + 1], A[r0 + 3, c0 + 2] = -1.0, -MU
            A[r0 + 4, c0 + 2] = -1.0
            l[r0:r0 + 4] = -np.inf
            l[r0 + 4] = -25.0
        rhs = np.zeros(12)
        rhs[8] = np.float64(np.float32(9.81)) * DT
        rhs += xref[:, k + 1] - Ad @ xref[:, k]
        u[12 * k:12 * k + 12] = rhs
        l[12 * k:12 * k + 12] = rhs
    w = np.array([2.0, 2.0, 20.0, 0.25, 0.25, 10.0] + [np.float32(0.2)] * 3 + [0.0, 0.0, np.float32(0.3)], dtype=np.float64)
    Pd = np.concatenate([np.tile(w, N), np.full(12 * N, np.float64(np.float32(5e-5)))])
    return A, l, u, Pd


def csc_to_dense(p, i, x, m, n):
    A = np.zeros((m, n))
    for j in range(n):
        A[i[p[j]:p[j + 1]], j] = x[p[j]:p[j + 1]]
    return A


@pytest.mark.parametrize("N", [16, 32])
def test_qp_assembly_matches_independent_dense_build(oracle_mod, synth_mod, N):
    sb = synth_mod.SyntheticBatch(2, N, gaits=("trot", "walk"), seed0=99)
    mpc = [oracle_mod.MPC(DT, N, DT * N, 40 if N > 20 else 20) for _ in range(2)]
    sb.N_gait = mpc[0].N_gait
    sb = synth_mod.SyntheticBatch(2, N, N_gait=mpc[0].N_gait, gaits=("trot", "walk"), seed0=99)
    for s in range(3):
        d = sb.step(s)
        for b in range(2):
            assert mpc[b].run(s, d["xref"][b], d["fsteps"][b]) == 0
            (p, i, x), (pp, pi, px), lo, up = mpc[b].qp()
            assert p[-1] == 126 * N - 18  # nnz invariant [SURVEY §8(c)(2)]
            A = csc_to_dense(p, i, x, 44 * N, 24 * N)
            A_ref, l_ref, u_ref, Pd = dense_qp(d["xref"][b], d["fsteps"][b], N, first_call=(s == 0))
            assert np.allclose(A, A_ref, rtol=1e-12, atol=1e-15), (s, b)
            assert np.allclose(up, u_ref, rtol=1e-12, atol=1e-14)
            assert np.array_equal(np.isinf(lo), np.isinf(l_ref))
            fin = np.isfinite(l_ref)
            assert np.allclose(lo[fin], l_ref[fin], rtol=1e-12, atol=1e-14)
            assert np.array_equal(px, Pd) and np.array_equal(pi, np.arange(24 * N))
            # column counts: state columns 2/3 (last block 1), force columns 7,7,10
            cnt = np.diff(p)
            assert list(cnt[:12]) == [2] * 6 + [3] * 6 and list(cnt[12 * (N - 1):12 * N]) == [1] * 12
            assert list(cnt[12 * N:12 * N + 3]) == [7, 7, 10]
            # gait / S getters (src/MPC.cpp:770-780)
            g = mpc[b].get_gait()
            assert np.array_equal(g[:N], (d["fsteps"][b][:N, 0::3] != 0).astype(float))
            assert np.array_equal(mpc[b].get_Sgait().reshape(N, 4, 3)[:, :, 0], 1.0 - g[:N])


def test_fourstance_immobile_properties(oracle_mod):
    """scripts/test_mpc.py:54-85: equal foot forces, state close to reference; sum f_z = m g."""
    N = 16
    m = oracle_mod.MPC(DT, N, 0.32, 20)
    xref = np.zeros((12, N + 1))
    xref[2, :] = 0.24474949993103629
    fsteps = np.zeros((20, 12))
    fsteps[:N, :] = [0.195, 0.147, 0., 0.195, -0.147, 0., -0.195, 0.147, 0., -0.195, -0.147, 0.]
    for i in range(100):
        m.run(i, xref, fsteps)
        r = m.get_latest_result()
        xref[:, 0] = r[:12, 0]
    assert r.shape == (24, N)
    assert np.allclose(r[12:, 0], np.tile(r[12:15, 0], 4), atol=1e-8)
    assert np.allclose(r[:12, 0], xref[:, 1], atol=1e-3)
    assert abs(r[14::3, 0].sum() - 9.81 * 2.50000279) < 1e-3  # the commented analytic value, test_mpc.py:85
    assert m.status == 1


def test_trot_solution_satisfies_optimality_conditions(oracle_mod, synth_mod):
    N = 16
    sb = synth_mod.SyntheticBatch(1, N, seed0=5)
    m = oracle_mod.MPC(DT, N, 0.32, 20)
    x0 = None
    for s in range(6):
        d = sb.step(s, x0)
        m.run(s, d["xref"][0], d["fsteps"][0])
        r = m.get_latest_result()
        x0 = r[:12, 0][None]
        assert m.status == 1 and m.iter % 25 == 0
        (p, i, x), (_, _, px), lo, up = m.qp()
        A = csc_to_dense(p, i, x, 44 * N, 24 * N)
        sol = m.solution()
        Ax = A @ sol
        tol = 1e-6 * (1 + np.abs(Ax).max())
        assert (Ax <= up + 5 * tol).all() and (Ax >= lo - 5 * tol).all()
        # swing feet carry no force, stance feet respect the friction pyramid
        f = r[12:, :].T.reshape(N, 4, 3)
        gait = d["gait"][0, :N]
        assert np.abs(f[gait == 0]).max() < 1e-4
        fs = f[gait == 1]
        assert (fs[:, 2] >= -1e-4).all() and (fs[:, 2] <= 25 + 1e-4).all()
        assert (np.abs(fs[:, 0]) <= MU * fs[:, 2] + 1e-4).all() and (np.abs(fs[:, 1]) <= MU * fs[:, 2] + 1e-4).all()
        # predicted states follow the dynamics rows exactly enough
        # dynamics rows hold to OSQP's primal tolerance eps_abs + eps_rel*max(|Ax|,|z|) (src/MPC.cpp:529-530)
        assert np.abs((Ax - up)[:12 * N]).max() <= 1.01e-6 * (1 + np.abs(Ax).max())


def test_first_call_uses_default_footholds_then_fsteps(oracle_mod, synth_mod):
    """src/MPC.cpp:223 vs :438 — B at num_iter==0 ignores fsteps and offset_CoM."""
    N = 16
    sb = synth_mod.SyntheticBatch(1, N, seed0=11)
    d = sb.step(0)
    m1, m2 = oracle_mod.MPC(DT, N, 0.32, 20), oracle_mod.MPC(DT, N, 0.32, 20)
    f2 = d["fsteps"][0].copy()
    f2[:, 0::3] = np.where(f2[:, 0::3] != 0, f2[:, 0::3] + 0.03, 0.0)  # same gait, other footholds
    m1.run(0, d["xref"][0], d["fsteps"][0])
    m2.run(0, d["xref"][0], f2)
    assert np.array_equal(m1.get_latest_result(), m2.get_latest_result())
    m1.run(1, d["xref"][0], d["fsteps"][0])
    m2.run(1, d["xref"][0], f2)
    assert not np.allclose(m1.get_latest_result(), m2.get_latest_result())
    # an un-setup object refuses num_iter != 0 (the reference would dereference a null workspace)
    assert oracle_mod.MPC(DT, N, 0.32, 20).run(3, d["xref"][0], d["fsteps"][0]) != 0


def test_fourstance_not_centered_known_answer(oracle_mod):
    """scripts/test_mpc.py:87-110: four-stance, non-centred start, 500 receding-horizon calls: equal foot forces and
    first predicted state within 1e-3 of the reference state."""
    import trot_kat

    N = trot_kat.N
    m = oracle_mod.MPC(DT, N, 0.32, trot_kat.N_GAIT)
    xref = np.zeros((12, N + 1))
    xref[2, :] = trot_kat.H_REF
    xref[:, 0] = trot_kat.NOT_CENTERED
    fsteps = np.zeros((trot_kat.N_GAIT, 12))
    fsteps[:N, :] = [0.195, 0.147, 0., 0.195, -0.147, 0., -0.195, 0.147, 0., -0.195, -0.147, 0.]
    for i in range(500):
        assert m.run(i, xref, fsteps) == 0
        r = m.get_latest_result()
        xref[:, 0] = r[:12, 0]
    assert np.allclose(r[12:, 0], np.tile(r[12:15, 0], 4))  # :109 (default tolerances, as in the reference)
    assert np.allclose(r[:12, 0], xref[:, 1], atol=1e-3)  # :110
    assert abs(r[14::3, 0].sum() - 9.81 * 2.50000279) < 1e-3


def test_twostance_centered_known_answer(oracle_mod):
    """scripts/test_mpc.py:136-160: two-stance trot (FL+HR / FR+HL, period 0.32 s) with the reference's
    receding-horizon roll (:96-133), centred start, 500 calls: first predicted state within 1e-2 of the reference
    state.  Inputs and criterion are the reference's own (tests/trot_kat.py restates them for today's API)."""
    import trot_kat

    m = oracle_mod.MPC(DT, trot_kat.N, 0.32, trot_kat.N_GAIT)
    stats = []

    def solve(i, xref, fsteps):
        assert m.run(i, xref, fsteps) == 0
        stats.append((m.iter, m.status))
        return m.get_latest_result()

    x_f, xref = trot_kat.run_twostance(solve, 500, centered=True)
    assert np.allclose(x_f[:12, 0], xref[:, 1], atol=1e-2), x_f[:12, 0] - xref[:, 1]  # :160
    fz = x_f[14::3, 0]
    assert abs(fz.sum() - 9.81 * 2.50000279) < 0.05 and (fz >= -1e-6).all()  # two stance feet carry the body
    assert (np.array(stats)[:, 1] == 1).all()  # OSQP_SOLVED on every call


def _exact_equality_qp(i, xref, fsteps, N):
    """The optimum of the MPC QP when no cone row is active: equality-constrained QP (dynamics rows + force-enable
    rows of the swing feet) solved through its dense KKT system.  No ADMM, no oracle code: numpy + dense_qp only."""
    A, lo, up, Pd = dense_qp(xref, fsteps, N, first_call=(i == 0))
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
    Ax = A @ sol
    cone = slice(24 * N, 44 * N)
    assert (Ax[cone] <= up[cone] + 1e-9).all() and (Ax[cone] >= lo[cone] - 1e-9).all()  # inactive, as assumed
    r = np.zeros((24, N))
    r[:12] = sol[:12 * N].reshape(N, 12).T + xref[:, 1:]
    r[12:] = sol[12 * N:].reshape(N, 12).T
    return r


def test_twostance_not_centered_criterion_is_not_met_by_the_qp_as_written(oracle_mod, monkeypatch):
    """scripts/test_mpc.py:162-190 (non-centred start, 2000 calls, within 1e-2) is STALE with respect to
    src/MPC.cpp: with the QP exactly as written there (weights :330: roll/pitch 0.25, roll/pitch rate 0) the closed
    loop `state := first predicted state` of this two-stance scenario is linearly unstable, so the criterion cannot be
    met by ANY exact solver of that QP.  Shown without the oracle's solver or assembly: for a start 1 % of the way to
    the reference's non-centred state no cone row is ever active, the optimum is an equality-constrained QP solved by
    one dense KKT system, and that closed loop grows by ~1.19 per gait period; the oracle follows it.  (The same holds
    for the older weight set left in the comment at :329.)  The reference's own start state then saturates f_z = 25
    within the first call and leaves the neighbourhood altogether."""
    import trot_kat

    N = trot_kat.N
    centre = np.zeros(12)
    centre[2] = trot_kat.H_REF
    monkeypatch.setattr(trot_kat, "NOT_CENTERED", centre + 0.01 * (trot_kat.NOT_CENTERED - centre))
    m = oracle_mod.MPC(DT, N, 0.32, trot_kat.N_GAIT)
    err_o, err_e = [], []

    def solve_oracle(i, xref, fsteps):
        assert m.run(i, xref, fsteps) == 0 and m.status == 1
        r = m.get_latest_result()
        err_o.append(r[:12, 0] - xref[:, 1])
        return r

    def solve_exact(i, xref, fsteps):
        r = _exact_equality_qp(i, xref, fsteps, N)
        err_e.append(r[:12, 0] - xref[:, 1])
        return r

    calls = 161
    trot_kat.run_twostance(solve_oracle, calls, centered=False)
    trot_kat.run_twostance(solve_exact, calls, centered=False)
    eo, ee = np.abs(np.array(err_o)).max(axis=1), np.abs(np.array(err_e)).max(axis=1)
    # same phase of the gait, one / ten periods apart: growth, not decay, in both
    assert ee[160] > 2.0 * ee[32] and eo[160] > 2.0 * eo[32], (eo[::16], ee[::16])
    # the oracle (eps 1e-6 ADMM, warm-started) tracks the exact optimum's closed loop while the deviation is small
    assert np.abs(np.array(err_o)[:64] - np.array(err_e)[:64]).max() < 1e-3
    # and from the reference's own start the criterion of :190 is far from met already after 400 calls
    monkeypatch.undo()
    m2 = oracle_mod.MPC(DT, N, 0.32, trot_kat.N_GAIT)

    def solve2(i, xref, fsteps):
        assert m2.run(i, xref, fsteps) == 0
        return m2.get_latest_result()

    x_f, xref = trot_kat.run_twostance(solve2, 400, centered=False)
    assert np.abs(x_f[:12, 0] - xref[:, 1]).max() > 1.0


def test_twostance_monodromy_explains_the_red_known_answer(oracle_mod, monkeypatch):
    """the one reference-held pin this build does not meet (scripts/test_mpc.py:162-190: two-stance trot
    from a non-centred start, 2000 calls, within 1e-2), turned from an argument into checked statements.
    (a) The one-gait-period closed-loop map of the unconstrained MPC law (tests/monodromy.py: independent assembly, dense
        KKT solves) has spectral radius 1.20 for the weights at src/MPC.cpp:330 and 1.45 for the set commented at :329:
        both > 1, the scenario's closed loop is linearly unstable for the QP as written.
    (b) The oracle's closed loop from a start 1 % of the way to the reference's non-centred state grows at exactly that
        rate (four periods apart: rho^4 within 15 %).
    (c) Which single change of w[] makes the reference's criterion reachable: roll weight 0.25 -> 0.01 (spectral radius
        0.79; lateral-velocity weight 0.2 -> 0 gives 0.80); raising the roll / pitch RATE weights from 0, the obvious
        candidate, makes it worse (1.63 at 0.3).  With the roll weight at 0.01 the full constrained QP -- the oracle's OSQP
        restatement on the independent assembly, from the reference's own start state, cone rows active in the first
        calls -- is `solved` on every call and within 1e-2 of the reference after 400 calls (2.0e-3; the remaining 1600
        calls of the reference's loop only contract further: spectral radius < 1).
    The HIP path has the reference's weights compiled in (mpc_kernel.hip w_all, as MPC.cpp:330): its growth rate is checked
    against (a) in tests/test_gpu_mpc.py; the modified weight set is a statement about the reference's test, not a
    product option."""
    import scipy.sparse as sp

    import monodromy as mono
    import trot_kat

    N = trot_kat.N
    rho_330, rho_329 = mono.spectral_radius(mono.W_330), mono.spectral_radius(mono.W_329)
    assert 1.17 < rho_330 < 1.23 and rho_329 > 1.3, (rho_330, rho_329)
    # (b) the oracle's own closed loop grows at that rate
    centre = np.zeros(12)
    centre[2] = trot_kat.H_REF
    monkeypatch.setattr(trot_kat, "NOT_CENTERED", centre + 0.01 * (trot_kat.NOT_CENTERED - centre))
    m = oracle_mod.MPC(DT, N, 0.32, trot_kat.N_GAIT)
    err = []

    def solve_oracle(i, xref, fsteps):
        assert m.run(i, xref, fsteps) == 0 and m.status == 1
        r = m.get_latest_result()
        err.append(np.abs(r[:12, 0] - xref[:, 1]).max())
        return r

    trot_kat.run_twostance(solve_oracle, 161, centered=False)
    growth = err[160] / err[96]
    assert 0.85 * rho_330 ** 4 < growth < 1.15 * rho_330 ** 4, (growth, rho_330 ** 4)
    monkeypatch.undo()
    # (c) single changes of w[]
    w_rate = mono.W_330.copy()
    w_rate[9] = w_rate[10] = 0.3
    assert mono.spectral_radius(w_rate) > rho_330
    w_fix = mono.W_330.copy()
    w_fix[3] = 0.01
    assert mono.spectral_radius(w_fix) < 0.9
    # the reference's loop with w_fix, full QP (cone rows included) through the oracle's OSQP restatement
    state = {}
    statuses = []

    def solve_fixed(i, xref, fsteps):
        A, lo, up, Pd = dense_qp(xref, fsteps, N, first_call=(i == 0))
        Pd = np.concatenate([np.tile(w_fix, N), Pd[12 * N:]])
        lo = np.where(np.isinf(lo), -1e30, lo)
        if not state:
            pat = A != 0
            for k in range(N):  # structural pattern: the B blocks' angular rows and the force-enable diagonal always stored
                pat[12 * k + 6:12 * k + 12, 12 * (N + k):12 * (N + k) + 12] = True
                for e in range(12):
                    pat[12 * N + 12 * k + e, 12 * (N + k) + e] = True
            rows, cols = np.nonzero(pat)
            order = np.lexsort((rows, cols))
            state["rc"] = (rows[order], cols[order])
            Asp = sp.csc_matrix((np.ones(len(rows)), state["rc"]), shape=A.shape)
            Asp.data[:] = A[state["rc"]]
            state["s"] = oracle_mod.OSQP(sp.diags(Pd).tocsc(), np.zeros(24 * N), Asp, lo, up, sigma=1e-6, eps_abs=1e-6,
                                         eps_rel=1e-6, eps_prim_inf=1e-5, eps_dual_inf=1e-4, alpha=1.6, adaptive_rho=1,
                                         adaptive_rho_interval=200, adaptive_rho_tolerance=5.0)
        else:
            state["s"].update_A(A[state["rc"]])
            state["s"].update_bounds(lo, up)
        s = state["s"]
        x, _ = s.solve()
        statuses.append(s.info()["status"])
        r = np.zeros((24, N))
        r[:12] = x[:12 * N].reshape(N, 12).T + xref[:, 1:]
        r[12:] = x[12 * N:].reshape(N, 12).T
        return r

    x_f, xref = trot_kat.run_twostance(solve_fixed, 400, centered=False)
    assert set(statuses) == {1}
    assert np.allclose(x_f[:12, 0], xref[:, 1], atol=1e-2), np.abs(x_f[:12, 0] - xref[:, 1]).max()  # scripts/test_mpc.py:190


def test_full_gait_table_without_a_zero_row_is_defined_behaviour(oracle_mod, synth_mod):
    """The oracle's ONE deliberate departure from the reference: `MPC::construct_gait` walks `fsteps` until its first
    all-zero row and then writes a zero row at that index (/root/reference/src/MPC.cpp:686-701) -- with N_gait == n_steps and
    every horizon step planned the table has no zero row, and the reference reads row N_gait of an N_gait-row matrix and writes
    `gait.row(N_gait)`: undefined behaviour there (Eigen without bounds checks).  Oracle and HIP kernel stop at the table's end.
    The agreed behaviour, pinned here: a full table gives exactly the result of the same rows followed by zero rows in a larger
    table (which IS defined in the reference), gait matrix and iteration counts included."""
    for N in (16, 8):
        full = synth_mod.SyntheticBatch(2, N, N_gait=N, gaits=("trot", "walk"), seed0=5000)
        padded = synth_mod.SyntheticBatch(2, N, N_gait=N + 4, gaits=("trot", "walk"), seed0=5000)
        ma = [oracle_mod.MPC(0.02, N, 0.02 * N, N) for _ in range(2)]
        mb = [oracle_mod.MPC(0.02, N, 0.02 * N, N + 4) for _ in range(2)]
        for s in range(3):
            da, db = full.step(s), padded.step(s)
            assert np.array_equal(da["fsteps"], db["fsteps"][:, :N]) and not db["fsteps"][:, N:].any()
            assert (np.abs(da["fsteps"]).sum(axis=2) > 0).all()  # the full table really has no zero row
            for b in range(2):
                assert ma[b].run(s, da["xref"][b], da["fsteps"][b]) == 0 and mb[b].run(s, db["xref"][b], db["fsteps"][b]) == 0
                assert np.array_equal(ma[b].get_latest_result(), mb[b].get_latest_result()), (N, s, b)
                assert ma[b].iter == mb[b].iter and ma[b].status == mb[b].status == 1
                assert np.array_equal(ma[b].get_gait(), mb[b].get_gait()[:N])
                assert np.array_equal(ma[b].get_Sgait(), mb[b].get_Sgait())


class StatefulDenseQP:
    """A second, independently written restatement of how the constraint matrix EVOLVES over warm-started calls
    (/root/reference/src/MPC.cpp:626-649 run, :686-701 construct_gait, :665-681 construct_S, :213-256 create_ML's value part,
    :418-464 update_ML): plain numpy, dense, stateful -- the B blocks and S flags of rows the current table does not reach keep
    whatever earlier calls left there.  Written from the reference's text, not from oracle/mpc_oracle.c."""

    def __init__(self, N, N_gait):
        self.N, self.N_gait = N, N_gait
        self.gait = np.zeros((N_gait + 1, 4))      # (+ a guard row: the reference's matrix has N_gait rows, tests/DESIGN.md 2)
        self.Bang = np.zeros((N, 3, 12))           # rows 9..11 of B_k
        self.S = np.zeros(12 * N)

    def _bang(self, yaw, levers):
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        Iinv = np.linalg.inv(R.T @ GI @ R)
        return np.hstack([DT * (Iinv @ skew(levers[:, i])) for i in range(4)])

    def call(self, num_iter, xref, fsteps):
        N = self.N
        k = 0                                       # construct_gait
        while k < self.N_gait and np.any(fsteps[k] != 0.0):
            self.gait[k] = (fsteps[k, 0::3] != 0.0)
            k += 1
        if k < self.N_gait:
            self.gait[k] = 0
        if num_iter == 0:                           # create_ML: default footholds, no CoM offset, every step
            foot = np.array([[0.19, 0.19, -0.19, -0.19], [0.15005, -0.15005, 0.15005, -0.15005], [0, 0, 0, 0]])
            for j in range(N):
                self.Bang[j] = self._bang(xref[5, j], foot - xref[0:3, j:j + 1])
            self.S[:] = 0.0
        else:                                       # update_ML: rows until the first all-zero GAIT row
            j = 0
            while np.any(self.gait[j]):
                levers = fsteps[j].reshape(4, 3).T - (xref[0:3, j:j + 1] + np.array([[0], [0], [-0.03]]))
                self.Bang[j] = self._bang(xref[5, j], levers)
                j += 1
        i = 0                                       # construct_S
        while np.any(self.gait[i]):
            self.S[12 * i:12 * i + 12] = np.repeat(1.0 - self.gait[i], 3)
            i += 1
        # the dense matrix: the constant part from dense_qp (full-stance dummy table), then the stateful values
        dummy = np.ones((self.N_gait, 12))
        A, l, u, Pd = dense_qp(xref, dummy, N, first_call=True)
        for j in range(N):
            A[12 * j + 9:12 * j + 12, 12 * (N + j):12 * (N + j) + 12] = self.Bang[j]
            for e in range(12):
                A[12 * N + 12 * j + e, 12 * (N + j) + e] = self.S[12 * j + e]
        return A, l, u, Pd


@pytest.mark.parametrize("N,full", [(5, False), (16, False), (24, False), (16, True), (12, False)])
def test_qp_assembly_on_arbitrary_contact_tables(oracle_mod, synth_mod, N, full):
    """The oracle's matrix values, bounds and getters over warm-started calls on ARBITRARY contact tables
    (synth.RandomContactTables: tables of any length that change completely between calls, so stale rows matter) and on the
    hand-made decoding corners of tests/test_gpu_mpc_random_tables.py (a footstep row whose x entries are all 0 but which is
    not all zero ends update_ML / construct_S and not construct_gait; a stance foot with x = 0 reads as swing; z-only rows),
    against StatefulDenseQP -- the pin that the wide GPU parity tests' oracle is right about them."""
    import test_gpu_mpc_random_tables as wide

    N_gait = N if full else max(20, N + 4)
    Bn = 6
    gen = synth_mod.RandomContactTables(Bn, N, N_gait=N_gait, seed0=4242 + N)
    hand = wide._hand_tables(N, N_gait)
    mpc = [oracle_mod.MPC(DT, N, DT * N, N_gait) for _ in range(Bn)]
    ind = [StatefulDenseQP(N, N_gait) for _ in range(Bn)]
    stale_seen = 0
    for c in range(7):
        d = gen.step(c)
        for b in range(Bn):
            fsteps = d["fsteps"][b]
            if not full and c in (2, 5):
                fsteps = hand[(b + c) % hand.shape[0]]
            xref = d["xref"][b]
            assert mpc[b].run(c, xref, fsteps) == 0
            (p, i, x), _, lo, up = mpc[b].qp()
            A = csc_to_dense(p, i, x, 44 * N, 24 * N)
            A_ref, l_ref, u_ref, _ = ind[b].call(c, xref, fsteps)
            assert np.allclose(A, A_ref, rtol=1e-12, atol=1e-15), (c, b, np.abs(A - A_ref).max())
            assert np.allclose(up, u_ref, rtol=1e-12, atol=1e-14)
            fin = np.isfinite(l_ref)
            assert np.array_equal(np.isinf(lo), ~fin) and np.allclose(lo[fin], l_ref[fin], rtol=1e-12, atol=1e-14)
            assert np.array_equal(mpc[b].get_gait(), ind[b].gait[:N_gait]), (c, b)
            assert np.array_equal(mpc[b].get_Sgait().ravel(), ind[b].S), (c, b)
            # did this call leave rows stale that a plain rewrite from the table would have changed?
            fresh = np.repeat(1.0 - (fsteps[:N, 0::3] != 0), 3, axis=1).ravel()
            stale_seen += int((fresh != ind[b].S).any())
    assert stale_seen == 0 if full else stale_seen > 5  # (a full table rewrites every row; the ragged ones really leave rows stale)
