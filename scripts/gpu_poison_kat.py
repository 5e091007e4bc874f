"""Known-answer solves of every horizon / launch form after the LDS of every compute unit was filled with NaN / Inf / huge patterns
(qrw_test_known_answer): finds reads of LDS the kernel has not written."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "quadruped-reactive-walking_amd")]
import qrw_hip
lib = qrw_hip.load_library()
bad = 0
pats = {"nan": 0xFFFFFFFFFFFFFFFF, "inf": 0x7FF0000000000000, "1e300": 0x7E37E43C8800759C, "none": 0}
for name, pat in pats.items():
    for mode in (0, 1, 2):
        for N in range(1, 33):
            if mode == 1 and N <= 16: continue
            it, st, rho, err = C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
            rc = lib.qrw_test_known_answer(N, mode, pat, 0 if name == "none" else 1, C.byref(it), C.byref(st), C.byref(rho), C.byref(err))
            if rc != 0:
                bad += 1
                print("FAIL pattern %s mode %d N %d: rc %d iters %d status %d rho %g err %g" % (name, mode, N, rc, it.value, st.value, rho.value, err.value), flush=True)
    print("pattern", name, "done", flush=True)
print("failures:", bad)
