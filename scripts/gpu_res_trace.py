"""Record, for S consecutive solves of BASELINE config 4's workload (batch 4096, N = 32, walk / trot / bounding), how far from
termination every solve is at each adaptive-rho test (every 200 iterations): primal residual / its tolerance, dual residual / its
tolerance, rho -- the data the time-sliced launch's priority levels are derived from (scripts/pre_priority_sim.py).
Needs the diagnostic build: make -C quadruped-reactive-walking_amd/csrc trace; QRW_HIP_LIB=build/libqrw_hip_trace.so."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("QRW_HIP_LIB", os.path.join(ROOT, "build", "libqrw_hip_trace.so"))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth
B, N, S = int(os.environ.get("QRW_TRACE_B", "4096")), int(os.environ.get("QRW_TRACE_N", "32")), int(os.environ.get("QRW_TRACE_S", "10"))
gaits = tuple(os.environ.get("QRW_TRACE_GAITS", "walk,trot,bounding").split(","))
Ng = max(20, N + 4)
sb = synth.SyntheticBatch(B, N, N_gait=Ng, gaits=gaits, n_seq=S + 1)
g = qrw_hip.Batch(B, N, N_gait=Ng, T_gait=0.02 * N)
its = np.zeros((S, B), dtype=np.int32)
W = 64 + 160  # kMpcProfItems of the trace build: 20 x (primal ratio, dual ratio, rho) at the rho tests + the primal ratio at every check
tr = np.zeros((S, B, W), dtype=np.float32)
buf = np.zeros((B, W), dtype=np.float64)
for s in range(S):
    d = sb.step(s)
    g.mpc_solve_host(d["xref"], d["fsteps"], s)
    its[s] = g.mpc_stats()["iters"]
    rc = g._lib.qrw_mpc_get_phase_cycles(g._handle, buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    tr[s] = buf
    print(s, its[s].mean(), (its[s] >= 4000).mean(), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "res_trace_n%d.npz" % N), iters=its, trace=tr, kind=sb.kind, phase=sb.phase)
