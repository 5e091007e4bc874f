// Shared device-side helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qrw {

typedef double v4d __attribute__((ext_vector_type(4)));

// ---- OSQP constants restated (third-party dependency of the reference, osqp constants.h 0.6.x)
constexpr double kOsqpInfty = 1e30;
constexpr double kRhoMin = 1e-6;
constexpr double kRhoMax = 1e6;
constexpr double kRhoEqOverIneq = 1e3;
constexpr double kMinScaling = 1e-4;
constexpr double kMaxScaling = 1e4;

// Every 64-byte line of the kernel-argument segment touched by one scalar load each, all in flight together, one wait.  The
// loop's kernels run one wavefront per SIMD and take 500-800 bytes of arguments that the compiler fetches where it needs them
// (the scalar register file cannot hold them all): each such fetch is then a full trip to memory on the wavefront's critical
// path (measured: 6.8 us of control_pre_quad_kernel's time appears / disappears with the argument segment in host / device
// memory, HIP_FORCE_DEV_KERNARG).  After this the later fetches hit the scalar cache: control_pre_quad_kernel 19.4 -> 18.4 us per
// 4096 robots; wbc16_kernel (400 bytes of arguments, fetched early anyway) gains nothing and does not use it.
// ONE asm statement: the destination registers stay reserved until the wait (separate statements would let the compiler reuse
// a destination while its load is still in flight -- seen as an intermittent fault in a first version).
template <int LINES>
__device__ __forceinline__ void kernarg_warm() {
  static_assert(LINES >= 1 && LINES <= 16, "one register per line");
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int kLast = 64 * (LINES - 1);
#define QRW_KO(i) ((64 * (i) < kLast) ? 64 * (i) : kLast)
  int t0, t1, t2, t3, t4, t5, t6, t7;
  if constexpr (LINES <= 8) {
    asm volatile("s_load_dword %0, %8, %9\n\ts_load_dword %1, %8, %10\n\ts_load_dword %2, %8, %11\n\ts_load_dword %3, %8, %12\n\t"
                 "s_load_dword %4, %8, %13\n\ts_load_dword %5, %8, %14\n\ts_load_dword %6, %8, %15\n\ts_load_dword %7, %8, %16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
                 : "s"(ka), "n"(QRW_KO(0)), "n"(QRW_KO(1)), "n"(QRW_KO(2)), "n"(QRW_KO(3)), "n"(QRW_KO(4)), "n"(QRW_KO(5)), "n"(QRW_KO(6)),
                   "n"(QRW_KO(7))
                 : "memory");
  } else {
    int t8, t9, t10, t11, t12, t13, t14, t15;
    asm volatile("s_load_dword %0, %16, %17\n\ts_load_dword %1, %16, %18\n\ts_load_dword %2, %16, %19\n\ts_load_dword %3, %16, %20\n\t"
                 "s_load_dword %4, %16, %21\n\ts_load_dword %5, %16, %22\n\ts_load_dword %6, %16, %23\n\ts_load_dword %7, %16, %24\n\t"
                 "s_load_dword %8, %16, %25\n\ts_load_dword %9, %16, %26\n\ts_load_dword %10, %16, %27\n\ts_load_dword %11, %16, %28\n\t"
                 "s_load_dword %12, %16, %29\n\ts_load_dword %13, %16, %30\n\ts_load_dword %14, %16, %31\n\ts_load_dword %15, %16, %32\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10),
                   "=&s"(t11), "=&s"(t12), "=&s"(t13), "=&s"(t14), "=&s"(t15)
                 : "s"(ka), "n"(QRW_KO(0)), "n"(QRW_KO(1)), "n"(QRW_KO(2)), "n"(QRW_KO(3)), "n"(QRW_KO(4)), "n"(QRW_KO(5)), "n"(QRW_KO(6)),
                   "n"(QRW_KO(7)), "n"(QRW_KO(8)), "n"(QRW_KO(9)), "n"(QRW_KO(10)), "n"(QRW_KO(11)), "n"(QRW_KO(12)), "n"(QRW_KO(13)),
                   "n"(QRW_KO(14)), "n"(QRW_KO(15))
                 : "memory");
  }
#undef QRW_KO
}

// ---- DPP quad operations on doubles (two 32-bit halves, 1 VALU op each, no LDS traffic)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // every lane has a source lane under these controls: no tied "old" value, so the result needs no copy of v first
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// broadcast lane J (0..3) of each quad to the whole quad
template <int J>
__device__ __forceinline__ double quad_bcast(double v) {
  return dpp_quad<J * 0x55>(v);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_quad<0xB1>(v);  // lanes [1,0,3,2]
  v += dpp_quad<0x4E>(v);  // lanes [2,3,0,1]
  return v;
}
__device__ __forceinline__ double quad_max(double v) {
  v = fmax(v, dpp_quad<0xB1>(v));
  v = fmax(v, dpp_quad<0x4E>(v));
  return v;
}
__device__ __forceinline__ double shfl(double v, int src) { return __shfl(v, src, 64); }
// max over the wavefront, result uniform: quad (DPP quad_perm), 16-lane row (DPP row_ror 4, 8), then the four rows
// through v_readlane -- no LDS round trips (the ds_bpermute butterfly costs four dependent ~100-clock trips)
__device__ __forceinline__ double wave_max(double v) {
  v = quad_max(v);
  v = fmax(v, dpp_quad<0x124>(v));  // row_ror:4
  v = fmax(v, dpp_quad<0x128>(v));  // row_ror:8
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return fmax(fmax(r0, r1), fmax(r2, r3));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = quad_sum(v);
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// A double parked in two accumulation registers (gfx950: 256 AGPRs beside the 256 architectural VGPRs; VALU
// instructions cannot read them, one v_accvgpr_read per half brings the value back).
// (plain registers instead, allocation left to the compiler: scripts/experiments/slower_forms.patch)
struct AccD {
  int lo, hi;
  __device__ __forceinline__ void set(double v) {
    const int l = __double2loint(v), h = __double2hiint(v);
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(l));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(h));
  }
  // twelve at once: one asm statement, so the compiler pads for the asm-to-VALU hazard once instead of per value
  static __device__ __forceinline__ void get12(const AccD (&a)[12], double (&v)[12]) {
    int l[12], h[12];
    asm("v_accvgpr_read_b32 %0, %24\n\tv_accvgpr_read_b32 %1, %25\n\tv_accvgpr_read_b32 %2, %26\n\tv_accvgpr_read_b32 %3, %27\n\t"
        "v_accvgpr_read_b32 %4, %28\n\tv_accvgpr_read_b32 %5, %29\n\tv_accvgpr_read_b32 %6, %30\n\tv_accvgpr_read_b32 %7, %31\n\t"
        "v_accvgpr_read_b32 %8, %32\n\tv_accvgpr_read_b32 %9, %33\n\tv_accvgpr_read_b32 %10, %34\n\tv_accvgpr_read_b32 %11, %35\n\t"
        "v_accvgpr_read_b32 %12, %36\n\tv_accvgpr_read_b32 %13, %37\n\tv_accvgpr_read_b32 %14, %38\n\tv_accvgpr_read_b32 %15, %39\n\t"
        "v_accvgpr_read_b32 %16, %40\n\tv_accvgpr_read_b32 %17, %41\n\tv_accvgpr_read_b32 %18, %42\n\tv_accvgpr_read_b32 %19, %43\n\t"
        "v_accvgpr_read_b32 %20, %44\n\tv_accvgpr_read_b32 %21, %45\n\tv_accvgpr_read_b32 %22, %46\n\tv_accvgpr_read_b32 %23, %47"
        : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3]), "=&v"(l[4]), "=&v"(h[4]),
          "=&v"(l[5]), "=&v"(h[5]), "=&v"(l[6]), "=&v"(h[6]), "=&v"(l[7]), "=&v"(h[7]), "=&v"(l[8]), "=&v"(h[8]), "=&v"(l[9]), "=&v"(h[9]),
          "=&v"(l[10]), "=&v"(h[10]), "=&v"(l[11]), "=&v"(h[11])
        : "a"(a[0].lo), "a"(a[0].hi), "a"(a[1].lo), "a"(a[1].hi), "a"(a[2].lo), "a"(a[2].hi), "a"(a[3].lo), "a"(a[3].hi), "a"(a[4].lo),
          "a"(a[4].hi), "a"(a[5].lo), "a"(a[5].hi), "a"(a[6].lo), "a"(a[6].hi), "a"(a[7].lo), "a"(a[7].hi), "a"(a[8].lo), "a"(a[8].hi),
          "a"(a[9].lo), "a"(a[9].hi), "a"(a[10].lo), "a"(a[10].hi), "a"(a[11].lo), "a"(a[11].hi));
#pragma unroll
    for (int c = 0; c < 12; c++) v[c] = __hiloint2double(h[c], l[c]);
  }
  __device__ __forceinline__ double get() const {
    int l, h;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
    return __hiloint2double(h, l);
  }
};

// Reciprocal from the hardware estimate and two Newton steps (<= 1 ulp for the well-scaled positive values it is used
// on: pivots, scalings, rho) -- a quarter of the instructions of the IEEE division sequence.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// OSQP limit_scaling()
__device__ __forceinline__ double limit_scaling(double d) {
  d = d < kMinScaling ? 1.0 : d;
  d = d > kMaxScaling ? kMaxScaling : d;
  return d;
}

}  // namespace qrw
