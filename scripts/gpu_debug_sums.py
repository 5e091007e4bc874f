"""Diagnostic: first-factorisation checksums of the MPC kernel (builds with -DQRW_DEBUG_SUMS), instance 0."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth
B, NH = 8, int(os.environ.get("QRW_PHASES_N", "16"))
sb = synth.SyntheticBatch(B, NH, N_gait=20, gaits=("trot",))
g = qrw_hip.Batch(B, NH, N_gait=20, T_gait=0.02 * NH)
lib = qrw_hip.load_library()
lib.qrw_mpc_get_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
d = sb.step(0)
g.mpc_solve_host(d["xref"], d["fsteps"], 0)
prof = np.zeros((B, 10)); lib.qrw_mpc_get_phase_cycles(g._handle, prof.ctypes.data_as(C.POINTER(C.c_double)))
np.set_printoptions(precision=15, linewidth=200)
print(os.environ.get("QRW_HIP_LIB"), g.mpc_stats()["iters"][:4], g.mpc_stats()["status"][:4])
print(prof[0]); print(prof[3])
