"""Per-task duration / iteration count in the classic launch and in the sequence launch (-DQRW_SEQ_STATS build)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
import qrw_hip, synth
B, N, W, K = 4096, 16, 4, 5
sb = synth.SyntheticBatch(B, N, N_gait=20, gaits=("trot",), n_seq=W + K)
steps = [sb.step(s) for s in range(W + K)]
dev = torch.device("cuda", 0)
xs = torch.from_numpy(np.stack([st["xref"] for st in steps])).to(dev)
fs = torch.from_numpy(np.stack([st["fsteps"] for st in steps])).to(dev)
lib = qrw_hip.load_library()
lib.qrw_mpc_get_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
def prof(eng):
    p = np.zeros((B, 10)); lib.qrw_mpc_get_phase_cycles(eng._handle, p.ctypes.data_as(C.POINTER(C.c_double))); return p
def report(tag, ticks, iters):
    us = ticks / 100.0
    ok = iters > 0
    per = us[ok] / iters[ok]
    # least squares: us = a + b * iters
    A = np.stack([np.ones(ok.sum()), iters[ok]], axis=1)
    a, b = np.linalg.lstsq(A, us[ok], rcond=None)[0]
    print("%-10s tasks %6d  mean %.1f us  mean iters %.0f  fit: %.1f us + %.3f us/iteration   (median us/iter %.3f)" % (
        tag, ok.sum(), us[ok].mean(), iters[ok].mean(), a, b, np.median(per)), flush=True)
eng = qrw_hip.Batch(B, n_steps=N, N_gait=20)
for s in range(W + K):
    eng.mpc_solve(xs[s], fs[s], s)
torch.cuda.synchronize()
p = prof(eng).reshape(-1)
report("classic", p[0:2 * B:2], p[1:2 * B:2])
eng.close()
eng = qrw_hip.Batch(B, n_steps=N, N_gait=20)
for s in range(W):
    eng.mpc_solve(xs[s], fs[s], s)
eng.mpc_solve_sequence(xs[W:W + K].contiguous(), fs[W:W + K].contiguous(), W)
torch.cuda.synchronize()
p = prof(eng).reshape(-1)
report("sequence", p[0:2 * B * K:2], p[1:2 * B * K:2])
