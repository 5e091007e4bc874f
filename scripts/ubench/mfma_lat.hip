// Micro-benchmark: dependent-chain latency and issue interval of the FP64 MFMA forms on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define REP 256
__global__ void k_dep16(double* out, unsigned long long* cyc, double a, double b) {
  v4d c = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep16_b(double* out, unsigned long long* cyc, double a, double b) {  // D feeds B of the next
  v4d c = {0, 0, 0, 0};
  double bb = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c, 0, 0, 0); bb = c[0]; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ind16(double* out, unsigned long long* cyc, double a, double b) {
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < REP / 4; i++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep4(double* out, unsigned long long* cyc, double a, double b) {
  double c = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dep4_b(double* out, unsigned long long* cyc, double a, double b) {
  double c = 0, bb = b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) { c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, 0, 0, 0); bb = c; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma(double* out, unsigned long long* cyc, double a, double b) {
  double c = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) c = fma(a, c, b);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_ind(double* out, unsigned long long* cyc, double a, double b) {
  double c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < REP / 8; i++) {
    c0 = fma(a, c0, b); c1 = fma(a, c1, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b);
    c4 = fma(a, c4, b); c5 = fma(a, c5, b); c6 = fma(a, c6, b); c7 = fma(a, c7, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_bperm(double* out, unsigned long long* cyc, double a, double b) {
  double c = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) c = __shfl(c, (threadIdx.x + 4) & 63, 64) + a;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(double* out, unsigned long long* cyc, double a, double b) {
  __shared__ double s[256];
  s[threadIdx.x] = threadIdx.x; s[threadIdx.x + 64] = 1; s[threadIdx.x + 128] = 2; s[threadIdx.x + 192] = 3;
  __syncthreads();
  int idx = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < REP; i++) idx = (int)s[idx & 255];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = idx;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
#define RUN(k) do { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.5); hipDeviceSynchronize(); \
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, 1.0000001, 0.5); hipDeviceSynchronize(); \
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-12s %8.1f ticks/op\n", #k, (double)h / REP); } while (0)
int main() {
  double* d; unsigned long long* c;
  hipMalloc((void**)&d, 64 * 8); hipMalloc((void**)&c, 8);
  RUN(k_dep16); RUN(k_dep16_b); RUN(k_ind16); RUN(k_dep4); RUN(k_dep4_b); RUN(k_fma); RUN(k_fma_ind); RUN(k_bperm); RUN(k_lds);
  // clock calibration: s_memtime ticks vs wall
  return 0;
}
