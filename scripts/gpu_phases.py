"""Phase-cycle breakdown of the MPC kernel (diagnostic build with -DQRW_PROFILE_PHASES)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["QRW_HIP_LIB"] = os.environ.get("QRW_HIP_LIB", os.path.join(ROOT, "build", "libqrw_hip_prof.so"))
os.environ["QRW_PREEMPT_CHUNK"] = "0"  # whole solves per workgroup: the counters of a time-sliced solve are those of its last slice only
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, time
import qrw_hip, synth
names = ["factor", "rhs", "elim_g", "fwd_chain", "middle", "bwd_chain", "backsub+A+upd", "check block (every 25th)", "setup (assemble+Ruiz)",
         "end of update + loop edge"]
NH = int(os.environ.get("QRW_PHASES_N", "16"))
for B in (8, 4096):
    sb = synth.SyntheticBatch(B, NH, N_gait=max(20, NH + 4), gaits=("trot",) if NH == 16 else ("walk", "trot", "bounding"))
    g = qrw_hip.Batch(B, NH, N_gait=max(20, NH + 4), T_gait=0.02 * NH)
    lib = qrw_hip.load_library()
    lib.qrw_mpc_get_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    for s in range(3):
        d = sb.step(s)
        t0 = time.time(); g.mpc_solve_host(d["xref"], d["fsteps"], s); t1 = time.time()
        prof = np.zeros((B, 10)); lib.qrw_mpc_get_phase_cycles(g._handle, prof.ctypes.data_as(C.POINTER(C.c_double)))
        it = g.mpc_stats()["iters"].astype(float)
        tot = prof.sum(1)
        print("B=%d step %d: wall %.2f ms, mean iters %.0f, cycles/iter (mean over instances) %.0f" % (B, s, (t1 - t0) * 1e3, it.mean(), (tot / it).mean()))
        per = prof / it[:, None]
        for i, n in enumerate(names):
            print("   %-26s %8.0f cyc/iter  (%.1f%%)" % (n, per[:, i].mean(), 100 * prof[:, i].sum() / tot.sum()))
