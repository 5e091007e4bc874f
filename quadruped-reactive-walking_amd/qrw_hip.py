"""ctypes binding of libqrw_hip.so (C ABI: include/qrw_hip.h) — the only way Python reaches the kernels.

Replaces the Boost.Python/eigenpy module of the reference (python/gepadd.cpp). The library is
gfx950-only and has NO CPU fallback: loading works anywhere (so symbols can be checked without a
GPU), but creating a handle without a HIP device raises.

`Batch` is the batched, device-resident API (torch tensors in, torch tensors out, caller's
stream); the *_host methods take/return numpy arrays and are what the single-robot drop-in
classes (libquadruped_reactive_walking.py) use.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("QRW_HIP_LIB", os.path.join(_HERE, "libqrw_hip.so"))  # override: diagnostic builds
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class QrwError(RuntimeError):
    pass


class _PlannerConfig(C.Structure):
    _fields_ = [("k_mpc", C.c_int32), ("h_ref", C.c_double), ("shoulders", C.c_double * 12), ("max_height", C.c_double),
                ("lock_time", C.c_double), ("init_target", C.c_double * 12), ("init_foot_pos", C.c_double * 12)]


class _IterationBuffers(C.Structure):
    """qrw_iteration_buffers (include/qrw_hip.h), field for field."""
    _fields_ = ([(n, C.c_void_p) for n in ("d_joy_vref", "d_q_filt", "d_v_filt", "d_rpy", "d_v_secu", "d_code")]
                + [("code_scalar", C.c_int32)]
                + [(n, C.c_void_p) for n in ("d_q", "d_v", "d_hv", "d_vref", "d_oRh_oTh", "d_xref", "d_target", "d_feet_pva",
                                              "d_contacts", "d_x_f_wbc", "d_q_wbc", "d_b_v", "d_f_cmd", "d_feet_cmd",
                                              "d_tau_ff", "d_qdes", "d_vdes", "d_f_with_delta", "d_ddq_res", "d_feet", "d_result",
                                              "d_error_flag")])


PLAN_GAIT, PLAN_FOOTSTEPS, PLAN_TRAJ, PLAN_STATE, PLAN_OUTPUTS = 2, 4, 8, 16, 32
SHOULDERS = np.array([[0.1946, 0.1946, -0.1946, -0.1946], [0.14695, -0.14695, 0.14695, -0.14695], [0.0, 0.0, 0.0, 0.0]])


class _Config(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n_steps", C.c_int32), ("N_gait", C.c_int32), ("device", C.c_int32),
                ("dt_mpc", C.c_double), ("T_gait", C.c_double), ("dt_wbc", C.c_double)]


# every symbol include/qrw_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
SIGNATURES = {
    "qrw_create": (C.c_int, [C.POINTER(_Config), C.POINTER(_vp)]),
    "qrw_destroy": (C.c_int, [_vp]),
    "qrw_last_error": (C.c_char_p, []),
    "qrw_mpc_solve": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "qrw_mpc_solve_host": (C.c_int, [_vp, _dp, _dp, _ip, C.c_int32, _dp]),
    "qrw_mpc_get_gait": (C.c_int, [_vp, C.c_int32, _dp, _dp]),
    "qrw_mpc_copy_iters": (C.c_int, [_vp, _vp, _vp]),
    "qrw_mpc_solve_sequence": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "qrw_mpc_sequence_error": (C.c_int, [_vp, _ip]),
    "qrw_mpc_get_stats": (C.c_int, [_vp, _ip, _ip, _dp, _dp, _dp]),
    "qrw_mpc_get_state": (C.c_int, [_vp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]),
    "qrw_mpc_get_order": (C.c_int, [_vp, _vp, _vp, _vp]),
    "qrw_mpc_get_slice_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "qrw_wbc_compute": (C.c_int, [_vp] + [_vp] * 13 + [_vp]),
    "qrw_wbc_compute_host": (C.c_int, [_vp] + [_dp] * 13),
    "qrw_wbc_get_stats": (C.c_int, [_vp, _ip, _ip, _dp, _dp]),
    "qrw_wbc_set_lanes": (C.c_int, [_vp, C.c_int32]),
    "qrw_invkin_host": (C.c_int, [_vp] + [_dp] * 12),
    "qrw_qpwbc_host": (C.c_int, [_vp] + [_dp] * 7),
    "qrw_fixed_feet_host": (C.c_int, [_vp] + [_dp] * 7),
    "qrw_get_base_inertia_diag": (C.c_int, [_vp, _dp]),
    "qrw_planner_init": (C.c_int, [_vp, _vp, _vp]),
    "qrw_planner_step": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp]),
    "qrw_planner_call_host": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_int32, _dp,
                                        C.c_double, _dp, _dp, _dp, _dp, _dp]),
    "qrw_planner_get_host": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "qrw_controller_init": (C.c_int, [_vp, _vp, C.c_double, _vp]),
    "qrw_controller_update_state": (C.c_int, [_vp] + [_vp] * 9 + [_vp]),
    "qrw_controller_wbc_inputs": (C.c_int, [_vp] + [_vp] * 9 + [_vp]),
    "qrw_controller_result": (C.c_int, [_vp] + [_vp] * 7 + [_vp]),
    "qrw_mpc_result_shift": (C.c_int, [_vp, _vp, _vp, _vp]),
    "qrw_control_pre": (C.c_int, [_vp, C.c_int32] + [_vp] * 4 + [_vp, C.c_int32, _vp] + [_vp] * 16 + [_vp]),
    "qrw_wbc_compute_result": (C.c_int, [_vp] + [_vp] * 13 + [_vp] * 4 + [_vp]),
    "qrw_iteration_bind": (C.c_int, [_vp, C.POINTER(_IterationBuffers)]),
    "qrw_iteration_step": (C.c_int, [_vp, C.c_int32, _vp, _vp]),
    "qrw_stream_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "qrw_stream_destroy": (C.c_int, [_vp]),
    "qrw_device_cu_count": (C.c_int, [C.c_int32, _ip]),
    "qrw_stream_wait_stream": (C.c_int, [_vp, _vp, _vp]),
    "qrw_selftest_sweeps": (C.c_int, [_dp]),
    "qrw_state_bytes": (C.c_int64, [_vp]),
}

# test-suite-only entry points (include/qrw_hip_test.h): fault injection and LDS / register poisoning
TEST_SIGNATURES = {
    "qrw_test_poke_aborted": (C.c_int, [_vp, C.c_int32]),
    "qrw_test_poison_probe": (C.c_int, [C.POINTER(C.c_uint32)]),
    "qrw_test_known_answer": (C.c_int, [C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load_library():
    """dlopen libqrw_hip.so and attach the signatures. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise QrwError("libqrw_hip.so is not built (run `make -C quadruped-reactive-walking_amd/csrc` or "
                       "__graft_entry__.build()); there is no CPU fallback")
    # more hardware queues than HIP's default four, for the multi-stream modes (StreamGroups, the asynchronous MPC mode):
    # streams that share a queue serialise.  Only effective if nothing in the process has touched the GPU yet.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the host API
        pass
    if os.path.basename(_LIB_PATH).startswith("WRONG_RESULTS") and os.environ.get("QRW_ALLOW_WRONG_RESULTS") != "1":
        raise QrwError("%s is a timing-experiment build that computes wrong results on purpose (scripts/experiments/); "
                       "set QRW_ALLOW_WRONG_RESULTS=1 to load it for a timing run" % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(TEST_SIGNATURES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().qrw_last_error()
        raise QrwError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def _h(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if a.shape != tuple(shape):
        a = a.reshape(shape)
    return a


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def device_cu_count(device=0):
    n = C.c_int32(0)
    _check(load_library().qrw_device_cu_count(int(device), C.byref(n)), "qrw_device_cu_count")
    return n.value


class CuStream:
    """A HIP stream restricted to compute units [first_cu, first_cu + n_cus) (all units if n_cus <= 0), wrapped as a
    torch.cuda.ExternalStream in `.torch` so that torch tensors / events and `with torch.cuda.stream(...)` work on it."""

    def __init__(self, device=0, first_cu=0, n_cus=0):
        import torch

        self._lib = load_library()
        p = _vp()
        _check(self._lib.qrw_stream_create(int(device), int(first_cu), int(n_cus), C.byref(p)), "qrw_stream_create")
        self.ptr = p.value
        self.torch = torch.cuda.ExternalStream(self.ptr, device=torch.device("cuda:%d" % device))

    def close(self):
        if getattr(self, "ptr", None):
            self.torch.synchronize()
            self._lib.qrw_stream_destroy(_vp(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def selftest_sweeps():
    err = C.c_double(0.0)
    rc = load_library().qrw_selftest_sweeps(C.byref(err))
    return rc, err.value


class Batch:
    """B robot instances resident on one GPU: persistent MPC / WBC solver state + kernels."""

    def __init__(self, batch, n_steps=16, N_gait=20, dt_mpc=0.02, T_gait=0.32, dt_wbc=0.002, device=0):
        self._lib = load_library()
        self.B, self.N, self.N_gait = int(batch), int(n_steps), int(N_gait)
        self.dt_mpc, self.dt_wbc, self.T_gait, self.device = float(dt_mpc), float(dt_wbc), float(T_gait), int(device)
        cfg = _Config(self.B, self.N, self.N_gait, self.device, self.dt_mpc, self.T_gait, self.dt_wbc)
        self._handle = _vp()
        _check(self._lib.qrw_create(C.byref(cfg), C.byref(self._handle)), "qrw_create")

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.qrw_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------ device-resident API (torch tensors)
    def _dev(self, t, shape):
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise QrwError("expected a contiguous float64 CUDA tensor")
        if t.device.index != self.device:
            raise QrwError("tensor lives on cuda:%s, this handle on cuda:%d" % (t.device.index, self.device))
        if tuple(t.shape) != tuple(shape):
            raise QrwError("bad shape %s, expected %s" % (tuple(t.shape), tuple(shape)))
        return _vp(t.data_ptr())

    def _stream(self):
        """The caller's current stream ON THIS HANDLE'S DEVICE (not on whatever device is current)."""
        import torch

        return _vp(torch.cuda.current_stream(self.device).cuda_stream)

    def mpc_solve(self, xref, fsteps, num_iter, out=None):
        """xref (B,12,N+1), fsteps (B,N_gait,12) CUDA float64; num_iter int or CUDA int32 (B,). Returns (B,24,N)."""
        import torch

        if out is None:
            out = torch.empty((self.B, 24, self.N), dtype=torch.float64, device=xref.device)
        ni_ptr, ni = _vp(0), 0
        if isinstance(num_iter, torch.Tensor):
            if num_iter.dtype != torch.int32 or not num_iter.is_cuda or tuple(num_iter.shape) != (self.B,):
                raise QrwError("num_iter tensor must be CUDA int32 of shape (B,)")
            ni_ptr = _vp(num_iter.data_ptr())
        else:
            ni = int(num_iter)
        _check(self._lib.qrw_mpc_solve(self._handle, self._dev(xref, (self.B, 12, self.N + 1)),
                                       self._dev(fsteps, (self.B, self.N_gait, 12)), ni_ptr, ni,
                                       self._dev(out, (self.B, 24, self.N)), self._stream()), "qrw_mpc_solve")
        return out

    def wbc_compute(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals, out=None):
        """All inputs CUDA float64 with leading dim B. Returns dict of CUDA tensors."""
        import torch

        B = self.B
        if out is None:
            dev = q.device
            out = dict(tau_ff=torch.empty((B, 12), dtype=torch.float64, device=dev),
                       qdes=torch.empty((B, 19), dtype=torch.float64, device=dev),
                       vdes=torch.empty((B, 18), dtype=torch.float64, device=dev),
                       f_with_delta=torch.empty((B, 12), dtype=torch.float64, device=dev),
                       ddq_res=torch.empty((B, 6), dtype=torch.float64, device=dev),
                       feet=torch.empty((B, 3, 3, 4), dtype=torch.float64, device=dev))
        _check(self._lib.qrw_wbc_compute(
            self._handle, self._dev(q, (B, 19)), self._dev(dq, (B, 18)), self._dev(f_cmd, (B, 12)),
            self._dev(contacts, (B, 4)), self._dev(pgoals, (B, 3, 4)), self._dev(vgoals, (B, 3, 4)),
            self._dev(agoals, (B, 3, 4)), self._dev(out["tau_ff"], (B, 12)), self._dev(out["qdes"], (B, 19)),
            self._dev(out["vdes"], (B, 18)), self._dev(out["f_with_delta"], (B, 12)),
            self._dev(out["ddq_res"], (B, 6)), self._dev(out["feet"], (B, 3, 3, 4)), self._stream()),
            "qrw_wbc_compute")
        return out

    def stream_wait_stream(self, waiter, signaller):
        """`waiter` (a torch stream) waits for what is on `signaller` now; no host synchronisation (qrw_stream_wait_stream)."""
        _check(self._lib.qrw_stream_wait_stream(self._handle, _vp(waiter.cuda_stream), _vp(signaller.cuda_stream)),
               "qrw_stream_wait_stream")

    def wbc_set_lanes(self, lanes):
        """Lanes per robot instance of the full WBC step: 16 (default, whole-chip streams) or 4 (streams that own few compute
        units); scheduling only, see qrw_wbc_set_lanes."""
        _check(self._lib.qrw_wbc_set_lanes(self._handle, int(lanes)), "qrw_wbc_set_lanes")

    def mpc_solve_sequence(self, xref, fsteps, first_num_iter=0, out=None, iters=None):
        """K consecutive MPC calls of every instance in one launch, ordered per instance only (qrw_mpc_solve_sequence):
        xref (K,B,12,N+1), fsteps (K,B,N_gait,12) CUDA float64 -> out (K,B,24,N); iters: optional CUDA int32 (K,B)."""
        import torch

        K = int(xref.shape[0])
        if out is None:
            out = torch.empty((K, self.B, 24, self.N), dtype=torch.float64, device=xref.device)
        ip = _vp(0)
        if iters is not None:
            if not (iters.is_cuda and iters.dtype == torch.int32 and iters.is_contiguous() and tuple(iters.shape) == (K, self.B)):
                raise QrwError("iters must be a contiguous CUDA int32 tensor of shape (K, B)")
            ip = _vp(iters.data_ptr())
        _check(self._lib.qrw_mpc_solve_sequence(self._handle, K, self._dev(xref, (K, self.B, 12, self.N + 1)),
                                                self._dev(fsteps, (K, self.B, self.N_gait, 12)), int(first_num_iter),
                                                self._dev(out, (K, self.B, 24, self.N)), ip, self._stream()),
               "qrw_mpc_solve_sequence")
        return out

    def mpc_sequence_timed_out(self):
        """Synchronises the device; True if the sequence kernel's task queue ever gave up waiting (never expected; the
        calls that did not run are NaN in `out` and -1 in `iters`)."""
        v = C.c_int32(0)
        _check(self._lib.qrw_mpc_sequence_error(self._handle, C.byref(v)), "qrw_mpc_sequence_error")
        return bool(v.value)

    def copy_mpc_iters(self, dst):
        """Last solve's ADMM iteration counts -> CUDA int32 tensor (B,), device to device on the current stream."""
        import torch

        if not (isinstance(dst, torch.Tensor) and dst.is_cuda and dst.dtype == torch.int32 and dst.is_contiguous()
                and tuple(dst.shape) == (self.B,) and dst.device.index == self.device):
            raise QrwError("copy_mpc_iters: expected a contiguous CUDA int32 tensor of shape (%d,) on cuda:%d" % (self.B, self.device))
        _check(self._lib.qrw_mpc_copy_iters(self._handle, _vp(dst.data_ptr()), self._stream()), "qrw_mpc_copy_iters")
        return dst

    # ------------------------------------------------ host-buffer API (numpy)
    def mpc_solve_host(self, xref, fsteps, num_iter):
        xref = _h(xref, (self.B, 12, self.N + 1))
        fsteps = _h(fsteps, (self.B, self.N_gait, 12))
        out = np.empty((self.B, 24, self.N))
        ni_arr, ni = None, 0
        if np.ndim(num_iter) > 0:
            ni_arr = np.ascontiguousarray(num_iter, dtype=np.int32).reshape(self.B)
        else:
            ni = int(num_iter)
        _check(self._lib.qrw_mpc_solve_host(self._handle, _p(xref), _p(fsteps),
                                            ni_arr.ctypes.data_as(_ip) if ni_arr is not None else None, ni, _p(out)),
               "qrw_mpc_solve_host")
        return out

    def wbc_compute_host(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals):
        B = self.B
        a = [_h(q, (B, 19)), _h(dq, (B, 18)), _h(f_cmd, (B, 12)), _h(contacts, (B, 4)), _h(pgoals, (B, 3, 4)),
             _h(vgoals, (B, 3, 4)), _h(agoals, (B, 3, 4))]
        o = dict(tau_ff=np.empty((B, 12)), qdes=np.empty((B, 19)), vdes=np.empty((B, 18)),
                 f_with_delta=np.empty((B, 12)), ddq_res=np.empty((B, 6)), feet=np.empty((B, 3, 3, 4)))
        _check(self._lib.qrw_wbc_compute_host(self._handle, *[_p(x) for x in a], _p(o["tau_ff"]), _p(o["qdes"]),
                                              _p(o["vdes"]), _p(o["f_with_delta"]), _p(o["ddq_res"]), _p(o["feet"])),
               "qrw_wbc_compute_host")
        return o

    def fixed_feet_host(self, q12, dq12):
        B = self.B
        posf, vf, wf, af = (np.empty((B, 4, 3)) for _ in range(4))
        Jf = np.empty((B, 12, 12))
        _check(self._lib.qrw_fixed_feet_host(self._handle, _p(_h(q12, (B, 12))), _p(_h(dq12, (B, 12))), _p(posf),
                                             _p(vf), _p(wf), _p(af), _p(Jf)), "qrw_fixed_feet_host")
        return posf, vf, wf, af, Jf

    def invkin_host(self, contacts, goals, vgoals, agoals, posf, vf, wf, af, Jf):
        B = self.B
        a = [_h(contacts, (B, 4)), _h(goals, (B, 3, 4)), _h(vgoals, (B, 3, 4)), _h(agoals, (B, 3, 4)),
             _h(posf, (B, 4, 3)), _h(vf, (B, 4, 3)), _h(wf, (B, 4, 3)), _h(af, (B, 4, 3)), _h(Jf, (B, 12, 12))]
        ddq, dq_cmd, q_step = np.empty((B, 12)), np.empty((B, 12)), np.empty((B, 12))
        _check(self._lib.qrw_invkin_host(self._handle, *[_p(x) for x in a], _p(ddq), _p(dq_cmd), _p(q_step)),
               "qrw_invkin_host")
        return ddq, dq_cmd, q_step

    def qpwbc_host(self, M, Jc, f_cmd, RNEA, want_H=True):
        B = self.B
        f_res, ddq_res = np.empty((B, 12)), np.empty((B, 6))
        H = np.empty((B, 12, 12)) if want_H else None
        _check(self._lib.qrw_qpwbc_host(self._handle, _p(_h(M, (B, 18, 18))), _p(_h(Jc, (B, 12, 18))),
                                        _p(_h(f_cmd, (B, 12))), _p(_h(RNEA, (B, 6))), _p(f_res), _p(ddq_res), _p(H)),
               "qrw_qpwbc_host")
        return f_res, ddq_res, H

    # ------------------------------------------------ planners
    def planner_init(self, k_mpc=10, h_ref=0.2229, shoulders=None, max_height=0.05, lock_time=0.07, init_target=None,
                     init_foot_pos=None):
        sh = _h(SHOULDERS if shoulders is None else shoulders, (3, 4))
        it = _h(sh if init_target is None else init_target, (3, 4))
        ip = _h(sh if init_foot_pos is None else init_foot_pos, (3, 4))
        pc = _PlannerConfig()
        pc.k_mpc, pc.h_ref, pc.max_height, pc.lock_time = int(k_mpc), float(h_ref), float(max_height), float(lock_time)
        for i in range(12):
            pc.shoulders[i], pc.init_target[i], pc.init_foot_pos[i] = sh.ravel()[i], it.ravel()[i], ip.ravel()[i]
        rc = self._lib.qrw_planner_init(self._handle, C.cast(C.byref(pc), _vp), _vp(0))
        if rc == -3:  # what Gait::initialize throws (src/Gait.cpp:30-31; std::invalid_argument -> ValueError in Python)
            raise ValueError(self._lib.qrw_last_error().decode())
        _check(rc, "qrw_planner_init")
        self.k_mpc = int(k_mpc)

    def planner_step(self, k, q7, hv, vref, code=0, out=None):
        """Device API: q7 (B,7) or the full q (B,19) of controller_update_state, hv (B,6), vref (B,6) CUDA float64; code int
        or CUDA int32 (B,). Returns dict (xref, fsteps, gait, target, feet_pva, contacts)."""
        import torch

        B, N, Ng = self.B, self.N, self.N_gait
        if out is None:
            dev = q7.device
            out = dict(xref=torch.empty((B, 12, N + 1), dtype=torch.float64, device=dev),
                       fsteps=torch.empty((B, Ng, 12), dtype=torch.float64, device=dev),
                       gait=torch.empty((B, Ng, 4), dtype=torch.float64, device=dev),
                       target=torch.empty((B, 3, 4), dtype=torch.float64, device=dev),
                       feet_pva=torch.empty((B, 3, 3, 4), dtype=torch.float64, device=dev),
                       contacts=torch.empty((B, 4), dtype=torch.float64, device=dev))
        cptr, cs = _vp(0), 0
        if isinstance(code, torch.Tensor):
            cptr = _vp(code.data_ptr())
        else:
            cs = int(code)
        q_ld = int(q7.shape[-1]) if q7.dim() == 2 else 7
        if q_ld not in (7, 19):
            raise QrwError("q7 must have shape (B,7) or (B,19)")
        _check(self._lib.qrw_planner_step(self._handle, int(k), self._dev(q7, (B, q_ld)), q_ld, self._dev(hv, (B, 6)),
                                          self._dev(vref, (B, 6)), cptr, cs, self._dev(out["xref"], (B, 12, N + 1)),
                                          self._dev(out["fsteps"], (B, Ng, 12)), self._dev(out["gait"], (B, Ng, 4)),
                                          self._dev(out["target"], (B, 3, 4)), self._dev(out["feet_pva"], (B, 3, 3, 4)),
                                          self._dev(out["contacts"], (B, 4)), self._stream()), "qrw_planner_step")
        return out

    def planner_call_host(self, mode, k=0, k_footsteps=0, refresh=False, q7=None, v6=None, vref6=None, code=0,
                          target_in=None, z_average=0.0, want=("xref", "fsteps", "gait", "target", "feet_pva")):
        B, N, Ng = self.B, self.N, self.N_gait
        o = dict(xref=np.empty((B, 12, N + 1)) if "xref" in want else None,
                 fsteps=np.empty((B, Ng, 12)) if "fsteps" in want else None,
                 gait=np.empty((B, Ng, 4)) if "gait" in want else None,
                 target=np.empty((B, 3, 4)) if "target" in want else None,
                 feet_pva=np.empty((B, 3, 3, 4)) if "feet_pva" in want else None)
        a = [None if x is None else _h(x, shp) for x, shp in ((q7, (B, 7)), (v6, (B, 6)), (vref6, (B, 6)))]
        t = None if target_in is None else _h(target_in, (B, 3, 4))
        _check(self._lib.qrw_planner_call_host(self._handle, int(mode), int(k), int(k_footsteps), int(bool(refresh)),
                                               _p(a[0]), _p(a[1]), _p(a[2]), int(code), _p(t), float(z_average),
                                               _p(o["xref"]), _p(o["fsteps"]), _p(o["gait"]), _p(o["target"]),
                                               _p(o["feet_pva"])), "qrw_planner_call_host")
        return o

    def planner_get(self, which, count, b=0):
        out = np.empty(int(count))
        _check(self._lib.qrw_planner_get_host(self._handle, int(which), int(b), int(count), _p(out)),
               "qrw_planner_get_host")
        return out

    # ------------------------------------------------ controller glue
    def controller_init(self, q_init12=None, h_ref=0.2229):
        """Controller.__init__ state (scripts/Controller.py:119-123,154). q_init12: CUDA (B,12) or None."""
        ptr = _vp(0) if q_init12 is None else self._dev(q_init12, (self.B, 12))
        _check(self._lib.qrw_controller_init(self._handle, ptr, float(h_ref), self._stream()), "qrw_controller_init")

    def controller_update_state(self, joy_v_ref, q_filt, v_filt, rpy, out=None):
        """Controller.updateState (scripts/Controller.py:381-426). CUDA float64 in, dict of CUDA tensors out."""
        import torch

        B = self.B
        if out is None:
            dev = q_filt.device
            out = dict(q=torch.empty((B, 19), dtype=torch.float64, device=dev),
                       v=torch.empty((B, 18), dtype=torch.float64, device=dev),
                       h_v=torch.empty((B, 6), dtype=torch.float64, device=dev),
                       v_ref=torch.empty((B, 6), dtype=torch.float64, device=dev),
                       oRh_oTh=torch.empty((B, 12), dtype=torch.float64, device=dev))
        _check(self._lib.qrw_controller_update_state(
            self._handle, self._dev(joy_v_ref, (B, 6)), self._dev(q_filt, (B, 19)), self._dev(v_filt, (B, 18)),
            self._dev(rpy, (B, 3)), self._dev(out["q"], (B, 19)), self._dev(out["v"], (B, 18)),
            self._dev(out["h_v"], (B, 6)), self._dev(out["v_ref"], (B, 6)), self._dev(out["oRh_oTh"], (B, 12)),
            self._stream()), "qrw_controller_update_state")
        return out

    def controller_wbc_inputs(self, x_f_mpc, xref, feet_pva, v, out=None):
        """WBC target assembly (scripts/Controller.py:258-296)."""
        import torch

        B, N = self.B, self.N
        if out is None:
            dev = xref.device
            out = dict(x_f_wbc=torch.empty((B, 24), dtype=torch.float64, device=dev),
                       q_wbc=torch.empty((B, 19), dtype=torch.float64, device=dev),
                       b_v=torch.empty((B, 18), dtype=torch.float64, device=dev),
                       f_cmd=torch.empty((B, 12), dtype=torch.float64, device=dev),
                       feet_cmd=torch.empty((3, B, 3, 4), dtype=torch.float64, device=dev))
        _check(self._lib.qrw_controller_wbc_inputs(
            self._handle, self._dev(x_f_mpc, (B, 24, N)), self._dev(xref, (B, 12, N + 1)),
            self._dev(feet_pva, (B, 3, 3, 4)), self._dev(v, (B, 18)), self._dev(out["x_f_wbc"], (B, 24)),
            self._dev(out["q_wbc"], (B, 19)), self._dev(out["b_v"], (B, 18)), self._dev(out["f_cmd"], (B, 12)),
            self._dev(out["feet_cmd"], (3, B, 3, 4)), self._stream()), "qrw_controller_wbc_inputs")
        return out

    def controller_result(self, tau_ff, qdes, vdes, q_filt, v_secu, out=None):
        """Result + security_check (scripts/Controller.py:306-310,341-365): (B,5,12) P,D,q_des,v_des,tau_ff and flag."""
        import torch

        B = self.B
        if out is None:
            dev = tau_ff.device
            out = dict(result=torch.empty((B, 5, 12), dtype=torch.float64, device=dev),
                       error_flag=torch.empty((B,), dtype=torch.int32, device=dev))
        if not (out["error_flag"].is_cuda and out["error_flag"].dtype == torch.int32 and out["error_flag"].is_contiguous()
                and tuple(out["error_flag"].shape) == (B,)):
            raise ValueError("error_flag must be a contiguous CUDA int32 tensor of shape (%d,)" % B)
        _check(self._lib.qrw_controller_result(
            self._handle, self._dev(tau_ff, (B, 12)), self._dev(qdes, (B, 19)), self._dev(vdes, (B, 18)),
            self._dev(q_filt, (B, 19)), self._dev(v_secu, (B, 12)), self._dev(out["result"], (B, 5, 12)),
            _vp(out["error_flag"].data_ptr()), self._stream()), "qrw_controller_result")
        return out

    def mpc_result_shift(self, gait, x_f_mpc):
        """MPC_Wrapper.solve bookkeeping (scripts/MPC_Wrapper.py:89-102) in place on x_f_mpc (B,24,N); gait (B,N_gait,4)."""
        _check(self._lib.qrw_mpc_result_shift(self._handle, self._dev(gait, (self.B, self.N_gait, 4)),
                                              self._dev(x_f_mpc, (self.B, 24, self.N)), self._stream()),
               "qrw_mpc_result_shift")
        return x_f_mpc

    # ------------------------------------------------ fused control iteration (two launches + the MPC solve)
    def control_pre(self, k, joy_v_ref, q_filt, v_filt, rpy, code=0, x_f_mpc=None, out=None, mpc_inputs=True):
        """update_state + planner_step (+ controller_wbc_inputs when x_f_mpc is given) in one launch.
        Returns one dict with the outputs of the three separate calls.  mpc_inputs=False (an iteration that does not
        solve, x_f_mpc given): `fsteps` and `gait` are not written and of `xref` only columns 0 and 1 are."""
        import torch

        B, N, Ng = self.B, self.N, self.N_gait
        if out is None:
            dev = q_filt.device
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
            out = dict(q=mk(B, 19), v=mk(B, 18), h_v=mk(B, 6), v_ref=mk(B, 6), oRh_oTh=mk(B, 12),
                       xref=mk(B, 12, N + 1), fsteps=mk(B, Ng, 12), gait=mk(B, Ng, 4), target=mk(B, 3, 4),
                       feet_pva=mk(B, 3, 3, 4), contacts=mk(B, 4),
                       x_f_wbc=mk(B, 24), q_wbc=mk(B, 19), b_v=mk(B, 18), f_cmd=mk(B, 12), feet_cmd=mk(3, B, 3, 4))
        cptr, cs = _vp(0), 0
        if isinstance(code, torch.Tensor):
            cptr = _vp(code.data_ptr())
        else:
            cs = int(code)
        xf = _vp(0) if x_f_mpc is None else self._dev(x_f_mpc, (B, 24, N))
        d = self._dev
        _check(self._lib.qrw_control_pre(
            self._handle, int(k), d(joy_v_ref, (B, 6)), d(q_filt, (B, 19)), d(v_filt, (B, 18)), d(rpy, (B, 3)), cptr, cs, xf,
            d(out["q"], (B, 19)), d(out["v"], (B, 18)), d(out["h_v"], (B, 6)), d(out["v_ref"], (B, 6)),
            d(out["oRh_oTh"], (B, 12)), d(out["xref"], (B, 12, N + 1)),
            d(out["fsteps"], (B, Ng, 12)) if mpc_inputs else _vp(0), d(out["gait"], (B, Ng, 4)) if mpc_inputs else _vp(0),
            d(out["target"], (B, 3, 4)), d(out["feet_pva"], (B, 3, 3, 4)),
            d(out["contacts"], (B, 4)), d(out["x_f_wbc"], (B, 24)), d(out["q_wbc"], (B, 19)), d(out["b_v"], (B, 18)),
            d(out["f_cmd"], (B, 12)), d(out["feet_cmd"], (3, B, 3, 4)), self._stream()), "qrw_control_pre")
        return out

    def wbc_compute_result(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals, q_filt, v_secu, out=None):
        """wbc_compute + controller_result in one launch; the dict also holds `result` (B,5,12) and `error_flag` (B,)."""
        import torch

        B = self.B
        if out is None:
            dev = q.device
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
            out = dict(tau_ff=mk(B, 12), qdes=mk(B, 19), vdes=mk(B, 18), f_with_delta=mk(B, 12), ddq_res=mk(B, 6),
                       feet=mk(B, 3, 3, 4), result=mk(B, 5, 12),
                       error_flag=torch.empty((B,), dtype=torch.int32, device=dev))
        d = self._dev
        _check(self._lib.qrw_wbc_compute_result(
            self._handle, d(q, (B, 19)), d(dq, (B, 18)), d(f_cmd, (B, 12)), d(contacts, (B, 4)), d(pgoals, (B, 3, 4)),
            d(vgoals, (B, 3, 4)), d(agoals, (B, 3, 4)), d(out["tau_ff"], (B, 12)), d(out["qdes"], (B, 19)),
            d(out["vdes"], (B, 18)), d(out["f_with_delta"], (B, 12)), d(out["ddq_res"], (B, 6)), d(out["feet"], (B, 3, 3, 4)),
            d(q_filt, (B, 19)), d(v_secu, (B, 12)), d(out["result"], (B, 5, 12)), _vp(out["error_flag"].data_ptr()),
            self._stream()), "qrw_wbc_compute_result")
        return out

    def bind_iteration(self, pre, post, inputs, stream=None):
        """An iteration that does not solve (control_pre without MPC inputs + wbc_compute_result) on FIXED buffers as one callable:
        `pre` / `post` are the dicts those two calls returned, `inputs` = (joy_v_ref, q_filt, v_filt, rpy, v_secu, code) the loop's
        input tensors (code: int or CUDA int32 (B,)).  Everything is validated ONCE and bound in the library (qrw_iteration_bind);
        step(k, x_f_mpc) then is one foreign call with three arguments (qrw_iteration_step: the two launches) -- a control loop
        passes the same buffers on every tick, and marshalling their ~50 pointers again was most of compute()'s host time.
        stream: a torch stream to launch on whatever the current stream is (None: the caller's current stream, looked up per call)."""
        import torch

        B, N = self.B, self.N
        d = self._dev
        joy_v_ref, q_filt, v_filt, rpy, v_secu, code = inputs
        ef = post["error_flag"]
        if not (ef.is_cuda and ef.dtype == torch.int32 and ef.is_contiguous() and tuple(ef.shape) == (B,)):
            raise QrwError("error_flag: expected a contiguous int32 CUDA tensor of shape (B,)")
        b = _IterationBuffers()
        b.d_joy_vref, b.d_q_filt, b.d_v_filt = d(joy_v_ref, (B, 6)), d(q_filt, (B, 19)), d(v_filt, (B, 18))
        b.d_rpy, b.d_v_secu = d(rpy, (B, 3)), d(v_secu, (B, 12))
        if isinstance(code, torch.Tensor):
            if not (code.is_cuda and code.dtype == torch.int32 and code.is_contiguous() and tuple(code.shape) == (B,)):
                raise QrwError("joystick code: expected a contiguous int32 CUDA tensor of shape (B,)")
            b.d_code, b.code_scalar = code.data_ptr(), 0
        else:
            b.d_code, b.code_scalar = None, int(code)
        for name, shape in (("q", (B, 19)), ("v", (B, 18)), ("hv", (B, 6)), ("vref", (B, 6)), ("oRh_oTh", (B, 12)),
                            ("xref", (B, 12, N + 1)), ("target", (B, 3, 4)), ("feet_pva", (B, 3, 3, 4)), ("contacts", (B, 4)),
                            ("x_f_wbc", (B, 24)), ("q_wbc", (B, 19)), ("b_v", (B, 18)), ("f_cmd", (B, 12)),
                            ("feet_cmd", (3, B, 3, 4))):
            setattr(b, "d_" + name, d(pre[{"hv": "h_v", "vref": "v_ref"}.get(name, name)], shape))
        for name, shape in (("tau_ff", (B, 12)), ("qdes", (B, 19)), ("vdes", (B, 18)), ("f_with_delta", (B, 12)),
                            ("ddq_res", (B, 6)), ("feet", (B, 3, 3, 4)), ("result", (B, 5, 12))):
            setattr(b, "d_" + name, d(post[name], shape))
        b.d_error_flag = ef.data_ptr()
        f_bind = self._lib.qrw_iteration_bind
        _check(f_bind(self._handle, C.byref(b)), "qrw_iteration_bind")
        # the library keeps ONE binding per handle (binding again replaces it): every callable remembers which binding is its
        # own and puts it back before it steps if another bind_iteration of this handle came in between (an earlier
        # callable must not run on the buffers of a later bind)
        self._bind_gen = gen = getattr(self, "_bind_gen", 0) + 1
        keep = (pre, post, inputs, b)  # the buffers stay alive as long as the callable does
        f_step, h, cur_stream = self._lib.qrw_iteration_step, self._handle, self._stream
        fixed = None if stream is None else _vp(stream.cuda_stream)
        last = [None, None, 0]  # the MPC result tensor validated last, its pointer argument and address (the loop alternates between a few buffers)

        def step(k, x_f_mpc):
            if self._bind_gen != gen:
                _check(f_bind(h, C.byref(keep[3])), "qrw_iteration_bind")
                self._bind_gen = gen
            if x_f_mpc is not last[0] or x_f_mpc.data_ptr() != last[2]:  # (same object, other storage: t.data = ..., set_())
                last[0], last[1], last[2] = x_f_mpc, d(x_f_mpc, (B, 24, N)), x_f_mpc.data_ptr()
            rc = f_step(h, k, last[1], cur_stream() if fixed is None else fixed)
            if rc:
                _check(rc, "qrw_iteration_step")

        return step

    # ------------------------------------------------ getters / diagnostics
    def mpc_gait(self, b=0):
        gait, S = np.empty((self.N_gait, 4)), np.empty(12 * self.N)
        _check(self._lib.qrw_mpc_get_gait(self._handle, int(b), _p(gait), _p(S)), "qrw_mpc_get_gait")
        return gait, S.reshape(-1, 1)

    def mpc_stats(self):
        it, st = np.empty(self.B, np.int32), np.empty(self.B, np.int32)
        rho, pri, dua = np.empty(self.B), np.empty(self.B), np.empty(self.B)
        _check(self._lib.qrw_mpc_get_stats(self._handle, it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), _p(rho),
                                           _p(pri), _p(dua)), "qrw_mpc_get_stats")
        return dict(iters=it, status=st, rho=rho, pri_res=pri, dua_res=dua)

    def mpc_order(self):
        """Diagnostic: (order, ema) of the next solve's longest-first block order, or None while there is none."""
        order, ema, has = np.empty(self.B, np.int32), np.empty(self.B, np.float32), C.c_int32(0)
        _check(self._lib.qrw_mpc_get_order(self._handle, order.ctypes.data_as(C.c_void_p), ema.ctypes.data_as(C.c_void_p),
                                           C.cast(C.byref(has), C.c_void_p)), "qrw_mpc_get_order")
        return (order, ema) if has.value else None

    def mpc_slice_stats(self):
        """Diagnostic: bookkeeping of the last time-sliced solve (N > 16, batch above the resident slots): priority levels and
        slice length in use, solves parked into each level, taker workgroups that took a parked solve, finished instances."""
        lv, ch, tk, fin = C.c_int32(0), C.c_int32(0), C.c_uint32(0), C.c_uint32(0)
        parks = np.zeros(9, np.uint32)
        _check(self._lib.qrw_mpc_get_slice_stats(self._handle, C.cast(C.byref(lv), _vp), C.cast(C.byref(ch), _vp),
                                                 parks.ctypes.data_as(C.c_void_p), C.cast(C.byref(tk), _vp),
                                                 C.cast(C.byref(fin), _vp)), "qrw_mpc_get_slice_stats")
        return {"levels": lv.value, "chunk": ch.value, "parks_per_level": parks, "takers": tk.value, "finished": fin.value}

    def mpc_state(self, b=0):
        N = self.N
        x, z, y, D, E = np.empty(24 * N), np.empty(44 * N), np.empty(44 * N), np.empty(24 * N), np.empty(44 * N)
        c = C.c_double(0.0)
        _check(self._lib.qrw_mpc_get_state(self._handle, int(b), _p(x), _p(z), _p(y), _p(D), _p(E), C.byref(c)),
               "qrw_mpc_get_state")
        return dict(x=x, z=z, y=y, D=D, E=E, c=c.value)

    def wbc_stats(self):
        it, st = np.empty(self.B, np.int32), np.empty(self.B, np.int32)
        rho, ksc = np.empty(self.B), np.empty((self.B, 4))
        _check(self._lib.qrw_wbc_get_stats(self._handle, it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), _p(rho),
                                           _p(ksc)), "qrw_wbc_get_stats")
        return dict(iters=it, status=st, rho=rho, k_since_contact=ksc)

    def base_inertia_diag(self):
        Y = np.empty(6)
        _check(self._lib.qrw_get_base_inertia_diag(self._handle, _p(Y)), "qrw_get_base_inertia_diag")
        return Y

    def state_bytes(self):
        return int(self._lib.qrw_state_bytes(self._handle))


# ---------------------------------------------------------------------------------------------------------------------
# One process-wide batch-1 handle for the single-robot drop-in classes (libquadruped_reactive_walking.MPC / QPWBC / InvKin,
# QP_WBC.wbc_controller, solo12InvKin.Solo12InvKin): the reference's Controller builds one object of each, and giving every
# object its own handle meant five qrw_create calls (25 allocations each) for one robot.  A handle holds one set of persistent
# state per family -- "mpc" (warm start, rho, stale B / S entries), "wbc" (the box-QP's iterates) -- so the first object that
# needs a family's state on a given configuration takes it from the shared handle; a second object of the same kind gets a
# handle of its own (its state must be separate), and stateless uses ("stateless": InvKin, the fixed-base foot kinematics, the
# base inertia) share without limit.  A family is handed out once per shared handle, never recycled: a new object must start
# from the state a fresh handle has.
_shared_batch1 = {}  # configuration -> (Batch, set of families handed out)


def shared_batch1(family, n_steps=16, N_gait=20, dt_mpc=0.02, T_gait=0.32, dt_wbc=0.002, device=0):
    """The process-wide batch-1 `Batch` of this configuration if `family` ("mpc", "wbc" or "stateless") is still free on it,
    a private one otherwise."""
    if family not in ("mpc", "wbc", "stateless"):
        raise QrwError("unknown state family %r" % (family,))
    key = (int(n_steps), int(N_gait), float(dt_mpc), float(T_gait), float(dt_wbc), int(device))
    ent = _shared_batch1.get(key)
    if ent is None:
        ent = _shared_batch1[key] = (Batch(1, n_steps=key[0], N_gait=key[1], dt_mpc=key[2], T_gait=key[3], dt_wbc=key[4],
                                           device=key[5]), set())
    b, taken = ent
    if family == "stateless":
        return b
    if family in taken:
        return Batch(1, n_steps=key[0], N_gait=key[1], dt_mpc=key[2], T_gait=key[3], dt_wbc=key[4], device=key[5])
    taken.add(family)
    return b


class StreamGroups:
    """One GPU's fleet as `groups` independent sub-batches, each with its own handle (`Batch`) and stream.

    A launch of `mpc_solve` ends with its longest solve (2 000-2 750 ADMM iterations against a mean of ~515 on the bench
    workload) while most of the chip is already idle; with two groups in flight the stragglers of one group run beside
    the next step of the other (+10 % control steps/s at batch 4096, DESIGN.md 4.1 "Block scheduling").  The robots are
    independent, so the results are the ones a single handle gives, bit for bit.

    `control_step` enqueues MPC solve -> f_cmd -> WBC for every group on that group's stream and returns at once; the
    outputs (views of `mpc_out`, `wbc_out[...]`, whole-fleet tensors) are valid after `synchronize()` or on the group's
    stream.  Inputs must stay untouched until then.  Groups below 1025 instances run without the longest-first block
    order (qrw_mpc_solve builds it for larger batches only); two groups are the measured optimum at batch 4096.

    The groups only overlap if their streams sit on DIFFERENT hardware queues.  HIP multiplexes a process's streams onto
    four of them (GPU_MAX_HW_QUEUES) in the order of first use, so a process that has already used other streams can end
    up with both groups on one queue — measured: 600 k instead of 970 k steps/s on the closed sequence with exactly two
    streams used earlier (scripts/gpu_stream_pool_exp.py).  The streams are therefore created once per device and group
    index and shared by every StreamGroups object of the process, and load_library() raises GPU_MAX_HW_QUEUES to 8 when
    it runs before the first HIP call of the process (set it yourself otherwise)."""

    _streams = {}  # (device, group index) -> torch.cuda.Stream, shared by all instances

    def __init__(self, batch, groups=2, n_steps=16, N_gait=20, dt_mpc=0.02, T_gait=0.32, dt_wbc=0.002, device=0):
        import torch

        if groups < 1 or batch % groups:
            raise QrwError("batch %d does not split into %d equal groups" % (batch, groups))
        self.B, self.S, self.Bs, self.N, self.N_gait, self.device = int(batch), int(groups), int(batch) // int(groups), int(n_steps), int(N_gait), int(device)
        dev = torch.device("cuda", self.device)
        self.engines = [Batch(self.Bs, n_steps=n_steps, N_gait=N_gait, dt_mpc=dt_mpc, T_gait=T_gait, dt_wbc=dt_wbc, device=device)
                        for _ in range(self.S)]
        for g in range(self.S):
            if (self.device, g) not in StreamGroups._streams:
                StreamGroups._streams[(self.device, g)] = torch.cuda.Stream(dev)
        self.streams = [StreamGroups._streams[(self.device, g)] for g in range(self.S)]
        self._sl = [slice(g * self.Bs, (g + 1) * self.Bs) for g in range(self.S)]
        f64 = dict(dtype=torch.float64, device=dev)
        self.mpc_out = torch.empty((self.B, 24, self.N), **f64)
        self.f_cmd = torch.empty((self.B, 12), **f64)
        self.wbc_out = dict(tau_ff=torch.empty((self.B, 12), **f64), qdes=torch.empty((self.B, 19), **f64),
                            vdes=torch.empty((self.B, 18), **f64), f_with_delta=torch.empty((self.B, 12), **f64),
                            ddq_res=torch.empty((self.B, 6), **f64), feet=torch.empty((self.B, 3, 3, 4), **f64))
        self._wbc_views = [{k: v[sl] for k, v in self.wbc_out.items()} for sl in self._sl]

    def control_step(self, xref, fsteps, num_iter, q, dq, contacts, pgoals, vgoals, agoals):
        """One control step (MPC + WBC, 1:1) of the whole fleet; all inputs whole-fleet CUDA float64 tensors
        (num_iter: int, or CUDA int32 (B,) as for Batch.mpc_solve)."""
        import torch

        cur = torch.cuda.current_stream(self.device)
        for g, (eng, st, sl) in enumerate(zip(self.engines, self.streams, self._sl)):
            st.wait_stream(cur)  # inputs produced on the caller's stream are ready
            with torch.cuda.stream(st):
                eng.mpc_solve(xref[sl], fsteps[sl], num_iter[sl] if isinstance(num_iter, torch.Tensor) else num_iter,
                              out=self.mpc_out[sl])
                self.f_cmd[sl].copy_(self.mpc_out[sl][:, 12:, 0])
                eng.wbc_compute(q[sl], dq[sl], self.f_cmd[sl], contacts[sl], pgoals[sl], vgoals[sl], agoals[sl],
                                out=self._wbc_views[g])
        return self.wbc_out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def join(self, stream=None):
        """Make `stream` (default: the caller's current stream) wait for every group's queued work, without blocking the host."""
        import torch

        stream = stream or torch.cuda.current_stream(self.device)
        for st in self.streams:
            stream.wait_stream(st)

    def close(self):
        self.synchronize()
        for e in self.engines:
            e.close()
