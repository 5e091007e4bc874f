"""ctypes loader for the CPU oracle (oracle/libqrw_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path
(quadruped-reactive-walking_amd/) never does.  PARITY UNPINNED: see oracle/qrw_oracle.h.

The classes mirror the reference binding surface (python/gepadd.cpp:22-31,186-195,217-224)
so that parity tests read like calls into the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(fast=False, out_dir=None):
    """Compile the oracle with gcc (make). Returns the path of the shared library."""
    target = "libqrw_oracle_fast.so" if fast else "libqrw_oracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _arr(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


_lib_cache = {}


def load(fast=False):
    key = bool(fast)
    if key in _lib_cache:
        return _lib_cache[key]
    path = os.path.join(_HERE, "libqrw_oracle_fast.so" if fast else "libqrw_oracle.so")
    if not fast and os.environ.get("QRW_ORACLE_LIB"):  # e.g. the sanitizer build (make -C oracle san)
        path = os.environ["QRW_ORACLE_LIB"]
    elif not os.path.exists(path):
        build(fast)
    lib = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "mpc_oracle_create": (vp, [C.c_double, C.c_int, C.c_double, C.c_int]),
        "mpc_oracle_destroy": (None, [vp]),
        "mpc_oracle_run": (C.c_int, [vp, C.c_int, _dp, _dp]),
        "mpc_oracle_get_latest_result": (None, [vp, _dp]),
        "mpc_oracle_get_gait": (None, [vp, _dp]),
        "mpc_oracle_get_Sgait": (None, [vp, _dp]),
        "mpc_oracle_restart": (C.c_int, [vp, C.c_double]),
        "mpc_oracle_iter": (C.c_int, [vp]),
        "mpc_oracle_status": (C.c_int, [vp]),
        "mpc_oracle_rho": (C.c_double, [vp]),
        "mpc_oracle_pri_res": (C.c_double, [vp]),
        "mpc_oracle_check_ratios": (None, [vp, _dp]),
        "mpc_oracle_dua_res": (C.c_double, [vp]),
        "mpc_oracle_nnz_ML": (C.c_int, [vp]),
        "mpc_oracle_get_ML": (None, [vp, _ip, _ip, _dp]),
        "mpc_oracle_get_P": (None, [vp, _ip, _ip, _dp]),
        "mpc_oracle_get_bounds": (None, [vp, _dp, _dp]),
        "mpc_oracle_get_solution": (None, [vp, _dp]),
        "mpc_oracle_get_iterates": (None, [vp, _dp, _dp, _dp]),
        "rbd_oracle_fixed_feet": (None, [_dp] * 7),
        "rbd_oracle_rnea": (None, [_dp] * 4),
        "rbd_oracle_crba_base_block": (None, [_dp, _dp]),
        "rbd_oracle_crba": (None, [_dp, _dp]),
        "rbd_oracle_feet_jacobians": (None, [_dp, _dp]),
        "invkin_oracle_refresh_and_compute": (None, [_dp] * 12),
        "qpwbc_oracle_create": (vp, []),
        "qpwbc_oracle_destroy": (None, [vp]),
        "qpwbc_oracle_run": (C.c_int, [vp, _dp, _dp, _dp, _dp, _dp]),
        "qpwbc_oracle_get_f_res": (None, [vp, _dp]),
        "qpwbc_oracle_get_ddq_res": (None, [vp, _dp]),
        "qpwbc_oracle_get_H": (None, [vp, _dp]),
        "qpwbc_oracle_iter": (C.c_int, [vp]),
        "qpwbc_oracle_status": (C.c_int, [vp]),
        "qpwbc_oracle_rho": (C.c_double, [vp]),
        "wbc_oracle_create": (vp, [C.c_double]),
        "wbc_oracle_destroy": (None, [vp]),
        "wbc_oracle_compute": (C.c_int, [vp] + [_dp] * 12),
        "wbc_oracle_qp_iter": (C.c_int, [vp]),
        "wbc_oracle_qp_status": (C.c_int, [vp]),
        "wbc_oracle_qp_rho": (C.c_double, [vp]),
        "wbc_oracle_get_feet": (None, [vp, _dp, _dp, _dp]),
        "wbc_oracle_get_k_since_contact": (None, [vp, _dp]),
        "planner_oracle_create": (vp, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, _dp,
                                       C.c_double, C.c_double, _dp, _dp]),
        "planner_oracle_destroy": (None, [vp]),
        "planner_oracle_gait_update": (None, [vp, C.c_int, _dp, C.c_int]),
        "planner_oracle_footsteps_update": (None, [vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp]),
        "planner_oracle_traj_update": (None, [vp, C.c_int, _dp]),
        "planner_oracle_state_compute": (None, [vp, _dp, _dp, _dp, C.c_double]),
        "planner_oracle_step": (None, [vp, C.c_int, _dp, _dp, _dp, C.c_int]),
        "planner_oracle_phase_duration": (C.c_double, [vp, C.c_int, C.c_int, C.c_double]),
        "planner_oracle_get_gaits": (None, [vp, _dp, _dp, _dp]),
        "planner_oracle_get_flags": (None, [vp, _dp]),
        "planner_oracle_get_xref": (None, [vp, _dp]),
        "planner_oracle_get_footsteps": (None, [vp, _dp, _dp, _dp]),
        "planner_oracle_get_feet": (None, [vp, _dp, _dp, _dp, _dp, _dp]),
        "planner_oracle_get_Rz": (None, [vp, _dp]),
        "mpc_oracle_run_batch": (C.c_int, [C.POINTER(vp), C.c_int, _ip, _dp, _dp, _dp, C.c_int]),
        "wbc_oracle_compute_batch": (C.c_int, [C.POINTER(vp), C.c_int] + [_dp] * 11 + [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib_cache[key] = lib
    return lib


class MPC:
    """Oracle counterpart of the bound class MPC (python/gepadd.cpp:22-31; src/MPC.cpp)."""

    def __init__(self, dt, n_steps, T_gait, N_gait, fast=False):
        self._lib = load(fast)
        self.n_steps, self.N_gait = int(n_steps), int(N_gait)
        self._h = self._lib.mpc_oracle_create(float(dt), int(n_steps), float(T_gait), int(N_gait))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.mpc_oracle_destroy(self._h)
            self._h = None

    def run(self, num_iter, xref_in, fsteps_in):
        xref = _arr(xref_in, (12, self.n_steps + 1))
        fsteps = _arr(fsteps_in, (self.N_gait, 12))
        return self._lib.mpc_oracle_run(self._h, int(num_iter), _ptr(xref), _ptr(fsteps))

    def cold_start(self, rho=0.1):
        """Test hook (no reference counterpart): osqp_update_rho(rho) + cold start of the workspace, what restarts a solver
        whose last solve was abandoned half-way (the HIP path's aborted time-sliced launch)."""
        return self._lib.mpc_oracle_restart(self._h, float(rho))

    def get_latest_result(self):
        out = np.zeros((24, self.n_steps))
        self._lib.mpc_oracle_get_latest_result(self._h, _ptr(out))
        return out

    def get_gait(self):
        out = np.zeros((self.N_gait, 4))
        self._lib.mpc_oracle_get_gait(self._h, _ptr(out))
        return out

    def get_Sgait(self):
        out = np.zeros((12 * self.n_steps, 1))
        self._lib.mpc_oracle_get_Sgait(self._h, _ptr(out))
        return out

    # --- introspection (tests only) ---
    @property
    def iter(self):
        return self._lib.mpc_oracle_iter(self._h)

    @property
    def status(self):
        return self._lib.mpc_oracle_status(self._h)

    @property
    def rho(self):
        return self._lib.mpc_oracle_rho(self._h)

    @property
    def residuals(self):
        return self._lib.mpc_oracle_pri_res(self._h), self._lib.mpc_oracle_dua_res(self._h)

    def qp(self):
        """(ML as (p,i,x), P diag as (p,i,x), l, u) exactly as handed to the solver."""
        N = self.n_steps
        nnz = self._lib.mpc_oracle_nnz_ML(self._h)
        p = np.zeros(24 * N + 1, np.int32)
        i = np.zeros(nnz, np.int32)
        x = np.zeros(nnz)
        self._lib.mpc_oracle_get_ML(self._h, p.ctypes.data_as(_ip), i.ctypes.data_as(_ip), _ptr(x))
        pp = np.zeros(24 * N + 1, np.int32)
        pi = np.zeros(24 * N, np.int32)
        px = np.zeros(24 * N)
        self._lib.mpc_oracle_get_P(self._h, pp.ctypes.data_as(_ip), pi.ctypes.data_as(_ip), _ptr(px))
        lo, up = np.zeros(44 * N), np.zeros(44 * N)
        self._lib.mpc_oracle_get_bounds(self._h, _ptr(lo), _ptr(up))
        return (p, i, x), (pp, pi, px), lo, up

    def solution(self):
        x = np.zeros(24 * self.n_steps)
        self._lib.mpc_oracle_get_solution(self._h, _ptr(x))
        return x

    def iterates(self):
        N = self.n_steps
        x, z, y = np.zeros(24 * N), np.zeros(44 * N), np.zeros(44 * N)
        self._lib.mpc_oracle_get_iterates(self._h, _ptr(x), _ptr(z), _ptr(y))
        return x, z, y


class InvKin:
    """Oracle counterpart of the bound class InvKin (python/gepadd.cpp:186-195; src/InvKin.cpp)."""

    def __init__(self, dt):
        self._lib = load()
        self.dt = dt
        self._q_step = np.zeros(12)
        self._dq_cmd = np.zeros(12)

    def refreshAndCompute(self, contacts, goals, vgoals, agoals, posf, vf, wf, af, Jf):
        ddq = np.zeros(12)
        args = [_arr(contacts, (4,)), _arr(goals, (3, 4)), _arr(vgoals, (3, 4)), _arr(agoals, (3, 4)),
                _arr(posf, (4, 3)), _arr(vf, (4, 3)), _arr(wf, (4, 3)), _arr(af, (4, 3)), _arr(Jf, (12, 12))]
        self._lib.invkin_oracle_refresh_and_compute(*[_ptr(a) for a in args], _ptr(ddq), _ptr(self._dq_cmd),
                                                    _ptr(self._q_step))
        return ddq

    def get_q_step(self):
        return self._q_step.copy()

    def get_dq_cmd(self):
        return self._dq_cmd.copy()


class QPWBC:
    """Oracle counterpart of the bound class QPWBC (python/gepadd.cpp:217-224; src/QPWBC.cpp)."""

    def __init__(self):
        self._lib = load()
        self._h = self._lib.qpwbc_oracle_create()

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.qpwbc_oracle_destroy(self._h)
            self._h = None

    def run(self, M, Jc, f_cmd, RNEA, k_contact):
        a = [_arr(M, (18, 18)), _arr(Jc, (12, 18)), _arr(f_cmd, (12,)), _arr(RNEA, (6,)), _arr(k_contact, (4,))]
        return self._lib.qpwbc_oracle_run(self._h, *[_ptr(x) for x in a])

    def get_f_res(self):
        o = np.zeros(12)
        self._lib.qpwbc_oracle_get_f_res(self._h, _ptr(o))
        return o

    def get_ddq_res(self):
        o = np.zeros(6)
        self._lib.qpwbc_oracle_get_ddq_res(self._h, _ptr(o))
        return o

    def get_H(self):
        o = np.zeros((12, 12))
        self._lib.qpwbc_oracle_get_H(self._h, _ptr(o))
        return o

    @property
    def iter(self):
        return self._lib.qpwbc_oracle_iter(self._h)

    @property
    def status(self):
        return self._lib.qpwbc_oracle_status(self._h)

    @property
    def rho(self):
        return self._lib.qpwbc_oracle_rho(self._h)


class WbcController:
    """Oracle counterpart of scripts/QP_WBC.py wbc_controller (compute only)."""

    def __init__(self, dt, fast=False):
        self._lib = load(fast)
        self.dt = dt
        self._h = self._lib.wbc_oracle_create(float(dt))
        self.tau_ff = np.zeros(12)
        self.qdes = np.zeros(19)
        self.vdes = np.zeros((18, 1))
        self.f_with_delta = np.zeros((12, 1))
        self.ddq_res = np.zeros(6)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.wbc_oracle_destroy(self._h)
            self._h = None

    def compute(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals):
        a = [_arr(q, (19,)), _arr(dq, (18,)), _arr(f_cmd, (12,)), _arr(contacts, (4,)), _arr(pgoals, (3, 4)),
             _arr(vgoals, (3, 4)), _arr(agoals, (3, 4))]
        vdes = np.zeros(18)
        f = np.zeros(12)
        rc = self._lib.wbc_oracle_compute(self._h, *[_ptr(x) for x in a], _ptr(self.tau_ff), _ptr(self.qdes),
                                          _ptr(vdes), _ptr(f), _ptr(self.ddq_res))
        self.vdes[:, 0] = vdes
        self.f_with_delta = f.reshape((-1, 1))
        return rc

    @property
    def qp_iter(self):
        return self._lib.wbc_oracle_qp_iter(self._h)

    def feet(self):
        p, e, v = np.zeros((3, 4)), np.zeros((3, 4)), np.zeros((3, 4))
        self._lib.wbc_oracle_get_feet(self._h, _ptr(p), _ptr(e), _ptr(v))
        return p, e, v

    @property
    def k_since_contact(self):
        k = np.zeros(4)
        self._lib.wbc_oracle_get_k_since_contact(self._h, _ptr(k))
        return k.reshape((1, 4))


# ---- free functions: rigid-body slices ----
def fixed_feet(q12, dq12):
    lib = load()
    posf, vf, wf, af, Jf = np.zeros((4, 3)), np.zeros((4, 3)), np.zeros((4, 3)), np.zeros((4, 3)), np.zeros((12, 12))
    lib.rbd_oracle_fixed_feet(_ptr(_arr(q12, (12,))), _ptr(_arr(dq12, (12,))), _ptr(posf), _ptr(vf), _ptr(wf),
                              _ptr(af), _ptr(Jf))
    return posf, vf, wf, af, Jf


def rnea(q19, v18, a18):
    lib = load()
    tau = np.zeros(18)
    lib.rbd_oracle_rnea(_ptr(_arr(q19, (19,))), _ptr(_arr(v18, (18,))), _ptr(_arr(a18, (18,))), _ptr(tau))
    return tau


def crba(q19):
    lib = load()
    M = np.zeros((18, 18))
    lib.rbd_oracle_crba(_ptr(_arr(q19, (19,))), _ptr(M))
    return M


def crba_base_block(q19):
    lib = load()
    M = np.zeros((6, 6))
    lib.rbd_oracle_crba_base_block(_ptr(_arr(q19, (19,))), _ptr(M))
    return M


def feet_jacobians(q19):
    lib = load()
    J = np.zeros((12, 18))
    lib.rbd_oracle_feet_jacobians(_ptr(_arr(q19, (19,))), _ptr(J))
    return J


class MPCBatch:
    """B independent oracle MPC objects stepped with OpenMP threads (bench cpu_baseline leg)."""

    def __init__(self, B, dt, n_steps, T_gait, N_gait, fast=True):
        self._lib = load(fast)
        self.B, self.n_steps, self.N_gait = B, n_steps, N_gait
        self._hs = (C.c_void_p * B)(*[self._lib.mpc_oracle_create(dt, n_steps, T_gait, N_gait) for _ in range(B)])

    def __del__(self):
        for h in getattr(self, "_hs", []):
            self._lib.mpc_oracle_destroy(h)
        self._hs = []

    def iters(self):
        """ADMM iteration counts / statuses of the last run, per instance (tests only)."""
        it = np.array([self._lib.mpc_oracle_iter(h) for h in self._hs], dtype=np.int32)
        st = np.array([self._lib.mpc_oracle_status(h) for h in self._hs], dtype=np.int32)
        return it, st

    def check_ratios(self):
        """(B,4): residual / tolerance of the LAST full termination check of every instance's last solve ({primal, dual}) and of
        the check BEFORE it -- how close to OSQP's threshold a termination decision was (tests: a solve that another arithmetic
        ends one check earlier or later is a rounding matter exactly when the deciding ratio is within ~1e-3 of 1)."""
        out = np.zeros((self.B, 4))
        for b, h in enumerate(self._hs):
            self._lib.mpc_oracle_check_ratios(h, _ptr(out[b]))
        return out

    def run(self, num_iter, xref, fsteps, threads):
        ni = np.ascontiguousarray(np.broadcast_to(np.asarray(num_iter, np.int32), (self.B,)))
        xref = _arr(xref, (self.B, 12, self.n_steps + 1))
        fsteps = _arr(fsteps, (self.B, self.N_gait, 12))
        out = np.zeros((self.B, 24, self.n_steps))
        self._lib.mpc_oracle_run_batch(self._hs, self.B, ni.ctypes.data_as(_ip), _ptr(xref), _ptr(fsteps), _ptr(out),
                                       int(threads))
        return out


class WbcBatch:
    def __init__(self, B, dt, fast=True):
        self._lib = load(fast)
        self.B = B
        self._hs = (C.c_void_p * B)(*[self._lib.wbc_oracle_create(dt) for _ in range(B)])

    def __del__(self):
        for h in getattr(self, "_hs", []):
            self._lib.wbc_oracle_destroy(h)
        self._hs = []

    def qp_stats(self):
        """(ADMM iteration counts, statuses, rho) of the box-QPs of the last compute, per instance (tests only)."""
        it = np.array([self._lib.wbc_oracle_qp_iter(h) for h in self._hs], dtype=np.int32)
        st = np.array([self._lib.wbc_oracle_qp_status(h) for h in self._hs], dtype=np.int32)
        rho = np.array([self._lib.wbc_oracle_qp_rho(h) for h in self._hs])
        return it, st, rho

    def compute(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals, threads):
        B = self.B
        a = [_arr(q, (B, 19)), _arr(dq, (B, 18)), _arr(f_cmd, (B, 12)), _arr(contacts, (B, 4)), _arr(pgoals, (B, 3, 4)),
             _arr(vgoals, (B, 3, 4)), _arr(agoals, (B, 3, 4))]
        tau, qdes, vdes, f = np.zeros((B, 12)), np.zeros((B, 19)), np.zeros((B, 18)), np.zeros((B, 12))
        self._lib.wbc_oracle_compute_batch(self._hs, B, *[_ptr(x) for x in a], _ptr(tau), _ptr(qdes), _ptr(vdes),
                                           _ptr(f), int(threads))
        return tau, qdes, vdes, f


# ---- raw access to the OSQP restatement (tests of the solver in isolation) ----
class _OqCsc(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("p", _ip), ("i", _ip), ("x", _dp)]


class _OqSettings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("eps_abs", C.c_double),
                ("eps_rel", C.c_double), ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("adaptive_rho_tolerance", C.c_double), ("max_iter", C.c_int), ("scaling", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int), ("check_termination", C.c_int),
                ("warm_start", C.c_int), ("scaled_termination", C.c_int)]


class OSQP:
    """Thin handle on oracle/osqp_restate.c. P (upper triangle) and A are scipy.sparse CSC."""

    def __init__(self, P, q, A, l, u, perm=None, **settings):
        lib = self._lib = load()
        vp = C.c_void_p
        lib.oq_set_default_settings.argtypes = [C.POINTER(_OqSettings)]
        lib.oq_setup.restype = vp
        lib.oq_setup.argtypes = [C.POINTER(_OqCsc), C.POINTER(_OqCsc), _dp, _dp, _dp, C.POINTER(_OqSettings), _ip]
        lib.oq_cleanup.argtypes = [vp]
        for f in ("oq_update_A", "oq_update_P", "oq_update_lin_cost", "oq_update_lower_bound",
                  "oq_update_upper_bound"):
            getattr(lib, f).argtypes = [vp, _dp]
            getattr(lib, f).restype = C.c_int
        lib.oq_update_bounds.argtypes = [vp, _dp, _dp]
        lib.oq_solve.argtypes = [vp]
        for f in ("oq_solution_x", "oq_solution_y", "oq_iter_x", "oq_iter_y", "oq_iter_z", "oq_scaling_D",
                  "oq_scaling_E"):
            getattr(lib, f).argtypes = [vp]
            getattr(lib, f).restype = _dp
        for f in ("oq_info_iter", "oq_info_status", "oq_info_rho_updates"):
            getattr(lib, f).argtypes = [vp]
            getattr(lib, f).restype = C.c_int
        for f in ("oq_info_pri_res", "oq_info_dua_res", "oq_info_rho", "oq_scaling_c"):
            getattr(lib, f).argtypes = [vp]
            getattr(lib, f).restype = C.c_double
        s = _OqSettings()
        lib.oq_set_default_settings(C.byref(s))
        for k, v in settings.items():
            setattr(s, k, v)
        self.n, self.m = A.shape[1], A.shape[0]
        keep = []

        def csc(M):
            M = M.tocsc()
            M.sort_indices()
            p = np.ascontiguousarray(M.indptr, np.int32)
            i = np.ascontiguousarray(M.indices, np.int32)
            x = np.ascontiguousarray(M.data, np.float64)
            keep.extend([p, i, x])
            return _OqCsc(M.shape[0], M.shape[1], p.ctypes.data_as(_ip), i.ctypes.data_as(_ip), _ptr(x))

        Pc, Ac = csc(P), csc(A)
        q, l, u = _arr(q), _arr(l), _arr(u)
        pp = None if perm is None else np.ascontiguousarray(perm, np.int32).ctypes.data_as(_ip)
        self._h = lib.oq_setup(C.byref(Pc), C.byref(Ac), _ptr(q), _ptr(l), _ptr(u), C.byref(s), pp)
        if not self._h:
            raise RuntimeError("oq_setup failed")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.oq_cleanup(self._h)
            self._h = None

    def update_A(self, x):
        return self._lib.oq_update_A(self._h, _ptr(_arr(x)))

    def update_P(self, x):
        return self._lib.oq_update_P(self._h, _ptr(_arr(x)))

    def update_lin_cost(self, q):
        return self._lib.oq_update_lin_cost(self._h, _ptr(_arr(q)))

    def update_bounds(self, l, u):
        return self._lib.oq_update_bounds(self._h, _ptr(_arr(l)), _ptr(_arr(u)))

    def solve(self):
        self._lib.oq_solve(self._h)
        x = np.ctypeslib.as_array(self._lib.oq_solution_x(self._h), (self.n,)).copy()
        y = np.ctypeslib.as_array(self._lib.oq_solution_y(self._h), (self.m,)).copy()
        return x, y

    def info(self):
        L = self._lib
        return dict(iter=L.oq_info_iter(self._h), status=L.oq_info_status(self._h), rho=L.oq_info_rho(self._h),
                    pri_res=L.oq_info_pri_res(self._h), dua_res=L.oq_info_dua_res(self._h),
                    rho_updates=L.oq_info_rho_updates(self._h), c=L.oq_scaling_c(self._h))

    def scaling(self):
        D = np.ctypeslib.as_array(self._lib.oq_scaling_D(self._h), (self.n,)).copy()
        E = np.ctypeslib.as_array(self._lib.oq_scaling_E(self._h), (self.m,)).copy()
        return D, E, self._lib.oq_scaling_c(self._h)


class Planner:
    """Oracle counterpart of the reference planner objects (Gait, StatePlanner, FootstepPlanner,
    FootTrajectoryGenerator; python/gepadd.cpp:44-181) wired as scripts/Controller.py:119-137 wires them."""

    def __init__(self, dt_mpc=0.02, dt_wbc=0.002, T_gait=0.32, T_mpc=0.32, N_gait=20, k_mpc=10, h_ref=0.2229,
                 shoulders=None, max_height=0.05, lock_time=0.07, init_target=None, init_foot_pos=None):
        self._lib = load()
        if shoulders is None:
            shoulders = np.array([[0.1946, 0.1946, -0.1946, -0.1946], [0.14695, -0.14695, 0.14695, -0.14695],
                                  [0.0, 0.0, 0.0, 0.0]])
        sh = _arr(shoulders, (3, 4))
        it = _arr(init_target if init_target is not None else sh, (3, 4))
        ip = _arr(init_foot_pos if init_foot_pos is not None else sh, (3, 4))
        self.N_gait, self.n_steps = int(N_gait), int(round(T_mpc / dt_mpc))
        self._h = self._lib.planner_oracle_create(dt_mpc, dt_wbc, T_gait, T_mpc, int(N_gait), int(k_mpc), h_ref,
                                                  _ptr(sh), max_height, lock_time, _ptr(it), _ptr(ip))
        if not self._h:
            raise ValueError("Sizes of matrices are too small for considered durations. Increase N_gait in config file.")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.planner_oracle_destroy(self._h)
            self._h = None

    def step(self, k, q7, h_v, vref, code=0):
        self._lib.planner_oracle_step(self._h, int(k), _ptr(_arr(q7, (7,))), _ptr(_arr(h_v, (6,))),
                                      _ptr(_arr(vref, (6,))), int(code))

    def gait_update(self, k, q7, code):
        self._lib.planner_oracle_gait_update(self._h, int(k), _ptr(_arr(q7, (7,))), int(code))

    def footsteps_update(self, refresh, k, q7, b_v, b_vref):
        out = np.zeros((3, 4))
        self._lib.planner_oracle_footsteps_update(self._h, int(bool(refresh)), int(k), _ptr(_arr(q7, (7,))),
                                                  _ptr(_arr(b_v, (6,))), _ptr(_arr(b_vref, (6,))), _ptr(out))
        return out

    def traj_update(self, k, target):
        self._lib.planner_oracle_traj_update(self._h, int(k), _ptr(_arr(target, (3, 4))))

    def state_compute(self, q7, v, vref, z_average=0.0):
        self._lib.planner_oracle_state_compute(self._h, _ptr(_arr(q7, (7,))), _ptr(_arr(v, (6,))),
                                               _ptr(_arr(vref, (6,))), float(z_average))

    def phase_duration(self, i, j, value):
        return self._lib.planner_oracle_phase_duration(self._h, int(i), int(j), float(value))

    def gaits(self):
        a, b, c = (np.zeros((self.N_gait, 4)) for _ in range(3))
        self._lib.planner_oracle_get_gaits(self._h, _ptr(a), _ptr(b), _ptr(c))
        return a, b, c

    def flags(self):
        f = np.zeros(4)
        self._lib.planner_oracle_get_flags(self._h, _ptr(f))
        return dict(new_phase=bool(f[0]), is_static=bool(f[1]), remaining_time=f[2], n_swing=int(f[3]))

    def xref(self):
        x = np.zeros((12, self.n_steps + 1))
        self._lib.planner_oracle_get_xref(self._h, _ptr(x))
        return x

    def Rz(self):
        """FootstepPlanner::getRz (src/FootstepPlanner.cpp:236)."""
        r = np.zeros((3, 3))
        self._lib.planner_oracle_get_Rz(self._h, _ptr(r))
        return r

    def footsteps(self):
        f, t, ot = np.zeros((self.N_gait, 12)), np.zeros((3, 4)), np.zeros((3, 4))
        self._lib.planner_oracle_get_footsteps(self._h, _ptr(f), _ptr(t), _ptr(ot))
        return f, t, ot

    def feet(self):
        p, v, a, t0, ts = np.zeros((3, 4)), np.zeros((3, 4)), np.zeros((3, 4)), np.zeros(4), np.zeros(4)
        self._lib.planner_oracle_get_feet(self._h, _ptr(p), _ptr(v), _ptr(a), _ptr(t0), _ptr(ts))
        return p, v, a, t0, ts
