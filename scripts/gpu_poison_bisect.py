"""Diagnostic build (-DQRW_DEBUG_POISON, QRW_HIP_LIB=build/lib_poison.so): the known-answer solve with ONE member of the kernel's LDS
struct filled with NaN at the top of the kernel -- which unwritten LDS does a solve read?"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "quadruped-reactive-walking_amd")]
import qrw_hip
lib = qrw_hip.load_library()
names = ["-", "sN", "sX", "sDump", "sE", "sW", "sOm", "sDg", "sA", "sB", "sRed+sPre"]
for N in (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,1,7,16,24").split(",")):
    for sel in range(0, 11):
        os.environ["QRW_DEBUG_POISON_SEL"] = str(sel)
        it, st, rho, err = C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
        rc = lib.qrw_test_known_answer(N, 0, 0, 0, C.byref(it), C.byref(st), C.byref(rho), C.byref(err))
        print("N %2d  poisoned %-10s rc %d iters %d status %d rho %g" % (N, names[sel], rc, it.value, st.value, rho.value), flush=True)
