"""A SECOND, independent implementation of the OSQP 0.6.x algorithm — dense numpy, test infrastructure only.

Why it exists: OSQP is a third-party dependency of the reference (README.md:21-31, call sites src/MPC.cpp:527-558,
src/QPWBC.cpp:239-270) that is absent from this image, and its returned iterate is path-dependent (warm start,
adaptive rho, termination only every 25 iterations).  oracle/osqp_restate.c restates the published algorithm in C with
a banded/sparse linear solve; this file restates it a second time, written separately from that C file, with dense
matrices and a dense LU of the quasi-definite KKT matrix, so that a transcription slip in either one shows up as a
different iteration count, rho or solution (tests/test_osqp_second_impl.py).  It follows the published algorithm
(Stellato et al., "OSQP: an operator splitting solver for quadratic programs", and the 0.6.x C sources' structure:
scale_data / set_rho_vec / update_xz_tilde / update_x / update_z / update_y / update_info / check_termination /
adapt_rho / osqp_update_*), not the reference repository.

Nothing outside tests/ may import this module.
"""
import numpy as np
import scipy.linalg as sla

OSQP_INFTY = 1e30
RHO_MIN, RHO_MAX = 1e-6, 1e6
RHO_EQ_OVER_RHO_INEQ = 1e3
RHO_TOL = 1e-4
MIN_SCALING, MAX_SCALING = 1e-4, 1e4

SOLVED, SOLVED_INACCURATE, MAX_ITER, PRIM_INF, PRIM_INF_INACC, DUAL_INF, DUAL_INF_INACC, NON_CVX, UNSOLVED = (
    1, 2, -2, -3, 3, -4, 4, -7, -10)


def _ninf(v):
    return np.abs(v).max() if v.size else 0.0


def _limit(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.where(v > MAX_SCALING, MAX_SCALING, v)


class OSQPNumpy:
    def __init__(self, P, q, A, l, u, rho=0.1, sigma=1e-6, max_iter=4000, eps_abs=1e-3, eps_rel=1e-3,
                 eps_prim_inf=1e-4, eps_dual_inf=1e-4, alpha=1.6, scaling=10, adaptive_rho=True,
                 adaptive_rho_interval=0, adaptive_rho_tolerance=5.0, check_termination=25, warm_start=True):
        """P: full symmetric n x n; A: m x n; l, u may hold +-inf (the C API does not clip them)."""
        self.P0, self.q0, self.A0 = np.array(P, float), np.array(q, float), np.array(A, float)
        self.l0, self.u0 = np.array(l, float), np.array(u, float)
        self.n, self.m = self.A0.shape[1], self.A0.shape[0]
        self.rho, self.sigma, self.max_iter = float(rho), float(sigma), int(max_iter)
        self.eps_abs, self.eps_rel = float(eps_abs), float(eps_rel)
        self.eps_prim_inf, self.eps_dual_inf = float(eps_prim_inf), float(eps_dual_inf)
        self.alpha, self.n_scaling = float(alpha), int(scaling)
        self.adaptive_rho, self.interval, self.tol = bool(adaptive_rho), int(adaptive_rho_interval), float(
            adaptive_rho_tolerance)
        assert self.interval > 0, "time-based adaptive-rho interval (0) is not reproducible; the reference sets 200"
        self.check_every, self.warm_start = int(check_termination), bool(warm_start)
        self._scale()
        self.ctype = np.full(self.m, 99)
        self.rho_vec = np.zeros(self.m)
        self._classify(force=True)
        self._factor()
        self.x, self.z, self.y = np.zeros(self.n), np.zeros(self.m), np.zeros(self.m)
        self.iter, self.status, self.rho_updates = 0, UNSOLVED, 0

    # ---- data scaling (Ruiz equilibration + cost normalisation), restarted from identity each time ----
    def _scale(self):
        P, q, A = self.P0.copy(), self.q0.copy(), self.A0.copy()
        D, E, c = np.ones(self.n), np.ones(self.m), 1.0
        for _ in range(self.n_scaling):
            dcol = np.maximum(np.abs(P).max(axis=0), np.abs(A).max(axis=0) if self.m else 0.0)
            erow = np.abs(A).max(axis=1) if self.m else np.zeros(0)
            d = 1.0 / np.sqrt(_limit(dcol))
            e = 1.0 / np.sqrt(_limit(erow))
            P = d[:, None] * P * d[None, :]
            A = e[:, None] * A * d[None, :]
            q = d * q
            D, E = D * d, E * e
            cost = max(np.abs(P).max(axis=0).mean(), float(_limit(np.array([_ninf(q)]))[0]))
            ct = 1.0 / float(_limit(np.array([cost]))[0])
            P, q, c = P * ct, q * ct, c * ct
        self.P, self.q, self.A, self.D, self.E, self.c = P, q, A, D, E, c
        self.l, self.u = E * self.l0, E * self.u0

    def _classify(self, force=False):
        """rho per constraint from the SCALED bounds; returns True when a class changed."""
        free = (self.l < -OSQP_INFTY * MIN_SCALING) & (self.u > OSQP_INFTY * MIN_SCALING)
        with np.errstate(invalid="ignore"):
            eq = ~free & (self.u - self.l < RHO_TOL)
        new = np.where(free, -1, np.where(eq, 1, 0))
        changed = force or bool((new != self.ctype).any())
        if changed:
            self.ctype = new
            self._rho_from_classes()
        return changed

    def _rho_from_classes(self):
        self.rho_vec = np.where(self.ctype == -1, RHO_MIN,
                                np.where(self.ctype == 1, RHO_EQ_OVER_RHO_INEQ * self.rho, self.rho))

    def _factor(self):
        K = np.zeros((self.n + self.m, self.n + self.m))
        K[:self.n, :self.n] = self.P + self.sigma * np.eye(self.n)
        K[:self.n, self.n:] = self.A.T
        K[self.n:, :self.n] = self.A
        K[self.n:, self.n:] = -np.diag(1.0 / self.rho_vec)
        self._lu = sla.lu_factor(K)

    # ---- osqp_update_* ----
    def update_A(self, A):
        self.A0 = np.array(A, float)  # the unscaled data are kept, so "unscale, overwrite, rescale" is a fresh _scale()
        self._scale()
        self._factor()

    def update_P(self, P):
        self.P0 = np.array(P, float)
        self._scale()  # NB: q0 here is whatever linear cost was installed last (the cost scale sees the OLD q)
        self._factor()

    def update_lin_cost(self, q):
        self.q0 = np.array(q, float)
        self.q = self.c * self.D * self.q0

    def update_bounds(self, l, u):
        self.l0, self.u0 = np.array(l, float), np.array(u, float)
        self.l, self.u = self.E * self.l0, self.E * self.u0
        if self._classify():
            self._factor()

    def update_upper_bound(self, u):
        self.u0 = np.array(u, float)
        self.u = self.E * self.u0
        if (self.u < self.l).any():
            return 1  # OSQP reports the error and returns before touching rho (the data stay overwritten)
        if self._classify():
            self._factor()
        return 0

    def update_lower_bound(self, l):
        self.l0 = np.array(l, float)
        self.l = self.E * self.l0
        if (self.u < self.l).any():
            return 1
        if self._classify():
            self._factor()
        return 0

    # ---- residuals / termination ----
    def _info(self):
        self.Ax, self.Px, self.Aty = self.A @ self.x, self.P @ self.x, self.A.T @ self.y
        self.pri_res = _ninf((self.Ax - self.z) / self.E)
        self.dua_res = _ninf((self.Px + self.q + self.Aty) / self.D) / self.c

    def _rho_estimate(self):
        pri = _ninf(self.Ax - self.z) / (max(_ninf(self.z), _ninf(self.Ax)) + 1e-10)
        dua = _ninf(self.Px + self.q + self.Aty) / (max(_ninf(self.q), _ninf(self.Aty), _ninf(self.Px)) + 1e-10)
        return min(max(self.rho * np.sqrt(pri / (dua + 1e-10)), RHO_MIN), RHO_MAX)

    def _primal_infeasible(self, eps):
        dy = self.delta_y
        nrm = _ninf(self.E * dy)
        if nrm > eps:
            with np.errstate(invalid="ignore"):
                lhs = float(np.sum(self.u * np.maximum(dy, 0) + self.l * np.minimum(dy, 0)))
            if lhs < -eps * nrm:  # False when lhs is NaN (an infinite bound times a zero)
                return _ninf((self.A.T @ dy) / self.D) < eps * nrm
        return False

    def _dual_infeasible(self, eps):
        dx = self.delta_x
        nrm = _ninf(self.D * dx)
        if nrm > eps:
            if float(self.q @ dx) < -self.c * eps * nrm:
                if _ninf((self.P @ dx) / self.D) < self.c * eps * nrm:
                    Adx = (self.A @ dx) / self.E
                    bad = ((self.u < OSQP_INFTY * MIN_SCALING) & (Adx > eps * nrm)) | (
                        (self.l > -OSQP_INFTY * MIN_SCALING) & (Adx < -eps * nrm))
                    return not bool(bad.any())
        return False

    def _check(self, approximate):
        k = 10.0 if approximate else 1.0
        ea, er, epi, edi = k * self.eps_abs, k * self.eps_rel, k * self.eps_prim_inf, k * self.eps_dual_inf
        if self.pri_res > OSQP_INFTY or self.dua_res > OSQP_INFTY or not np.isfinite(self.pri_res + self.dua_res):
            self.status = NON_CVX
            return True
        prim_ok = prim_inf = dual_ok = dual_inf = False
        if self.m == 0:
            prim_ok = True
        else:
            eps_prim = ea + er * max(_ninf(self.z / self.E), _ninf(self.Ax / self.E))
            if self.pri_res < eps_prim:
                prim_ok = True
            else:
                prim_inf = self._primal_infeasible(epi)
        eps_dual = ea + er * max(_ninf(self.q / self.D), _ninf(self.Aty / self.D), _ninf(self.Px / self.D)) / self.c
        if self.dua_res < eps_dual:
            dual_ok = True
        else:
            dual_inf = self._dual_infeasible(edi)
        if prim_ok and dual_ok:
            self.status = SOLVED_INACCURATE if approximate else SOLVED
            return True
        if prim_inf:
            self.status = PRIM_INF_INACC if approximate else PRIM_INF
            return True
        if dual_inf:
            self.status = DUAL_INF_INACC if approximate else DUAL_INF
            return True
        return False

    # ---- osqp_solve ----
    def solve(self):
        n, m, a = self.n, self.m, self.alpha
        if not self.warm_start:
            self.x[:], self.z[:], self.y[:] = 0, 0, 0
        self.status, self.rho_updates = UNSOLVED, 0
        it = 0
        done = False
        for it in range(1, self.max_iter + 1):
            xp, zp = self.x, self.z
            sol = sla.lu_solve(self._lu, np.concatenate([self.sigma * xp - self.q, zp - self.y / self.rho_vec]))
            xt = sol[:n]
            zt = zp + (sol[n:] - self.y) / self.rho_vec
            self.x = a * xt + (1 - a) * xp
            self.delta_x = self.x - xp
            zr = a * zt + (1 - a) * zp
            self.z = np.minimum(np.maximum(zr + self.y / self.rho_vec, self.l), self.u)
            self.delta_y = self.rho_vec * (zr - self.z)
            self.y = self.y + self.delta_y
            can_check = self.check_every and it % self.check_every == 0
            if can_check:
                self._info()
                if self._check(False):
                    done = True
                    break
            if self.adaptive_rho and it % self.interval == 0:
                if not can_check:
                    self._info()
                est = self._rho_estimate()
                if est > self.rho * self.tol or est < self.rho / self.tol:
                    self.rho = min(max(est, RHO_MIN), RHO_MAX)
                    self._rho_from_classes()
                    self._factor()
                    self.rho_updates += 1
        self.iter = it
        if not done:
            if not (self.check_every and it % self.check_every == 0):
                self._info()
            if not self._check(True):
                self.status = MAX_ITER
        if self.status in (PRIM_INF, PRIM_INF_INACC, DUAL_INF, DUAL_INF_INACC, NON_CVX):
            sx, sy = np.full(n, np.nan), np.full(m, np.nan)
            self.x, self.z, self.y = np.zeros(n), np.zeros(m), np.zeros(m)  # iterates cannot seed the next run
        else:
            sx, sy = self.D * self.x, self.E * self.y / self.c
        return sx, sy
