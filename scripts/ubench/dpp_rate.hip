// Micro-benchmark: issue rate / latency of v_fmac_f64_dpp (row_newbcast) vs v_fma_f64 for one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(X) X X X X X X X X
__global__ void k(double* out, unsigned long long* cyc, double s) {
  double x = threadIdx.x * 0.001 + s, m = 1.0000001;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  unsigned long long t0, t1, t2, t3, t4;
  t0 = __builtin_amdgcn_s_memtime();
  // 64 dependent fmac_dpp (one accumulator)
  asm volatile(R8(R8("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a0) : "v"(x), "v"(m));
  t1 = __builtin_amdgcn_s_memtime();
  // 64 fmac_dpp over 8 independent accumulators
  asm volatile(R8("v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                  "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(m));
  t2 = __builtin_amdgcn_s_memtime();
  // 64 plain v_fma_f64 over 8 accumulators
  asm volatile(R8("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                  "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(m));
  t3 = __builtin_amdgcn_s_memtime();
  // 64 dependent plain v_fma_f64
  asm volatile(R8(R8("v_fma_f64 %0, %1, %2, %0\n\t")) : "+v"(a0) : "v"(x), "v"(m));
  t4 = __builtin_amdgcn_s_memtime();
  // 64 two-accumulator fmac_dpp (the chain's pattern)
  asm volatile(R8(R8("v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a0), "+v"(a1) : "v"(x), "v"(m));
  unsigned long long t5 = __builtin_amdgcn_s_memtime();
  // 3 accumulators
  asm volatile(R8(R8("v_fmac_f64_dpp %0, %3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %3, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x), "v"(m));
  unsigned long long t6 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 64);
  unsigned long long h[6];
  for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, 0.5); hipDeviceSynchronize(); }
  hipMemcpy(h, c, 48, hipMemcpyDeviceToHost);
  printf("ticks per instr: dep fmac_dpp %.2f | 8-acc fmac_dpp %.2f | 8-acc fma %.2f | dep fma %.2f | 2-acc fmac_dpp %.2f | 3-acc fmac_dpp %.2f\n",
         h[0] / 64.0, h[1] / 64.0, h[2] / 64.0, h[3] / 64.0, h[4] / 128.0, h[5] / 192.0);
  return 0;
}
