// Micro-benchmark: 12x12 block-bidiagonal sweep u_k = r_k + M_k u_{k-1} done on the FP64 VALU with DPP row
// broadcasts (v_fmac_f64_dpp row_newbcast:j), one matrix row per lane, vs a scalar loop. Prints cycles per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>

#define FM(J) "v_fmac_f64_dpp %0, %2, %" #J "+3 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define FM1(J) "v_fmac_f64_dpp %1, %2, %" #J "+3 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"

#if ACC == 1
__device__ __forceinline__ double step12(double r, double x, const double m[12]) {
  double a0 = r;
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(a0)
      : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]),
        "v"(m[9]), "v"(m[10]), "v"(m[11]));
  return a0;
}
#else
__device__ __forceinline__ double step12(double r, double x, const double m[12]) {
  double a0 = r, a1 = 0.0;
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(a0), "+v"(a1)
      : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]),
        "v"(m[9]), "v"(m[10]), "v"(m[11]));
  return a0 + a1;
}
#endif

#ifndef CS
#define CS 12
#endif
#ifndef TR
#define TR 0
#endif
constexpr int kSteps = 15, kSlot = 12 * CS;
// entry (i,c) of a step's matrix: TR=0 at i*CS+c (lane reads contiguous), TR=1 at c*CS+i (lane reads strided)
#define MIDX(i, c) (TR ? (c) * CS + (i) : (i) * CS + (c))

__global__ void k_chain(const double* M, const double* r, double* out, unsigned long long* cyc) {
  __shared__ double sM[4][kSteps * kSlot];
  __shared__ double sR[4][(kSteps + 1) * 12];
  __shared__ double sU[4][(kSteps + 1) * 12];
  const int lane = threadIdx.x, row = lane >> 4, i = lane & 15;
  for (int e = lane; e < 4 * kSteps * kSlot; e += 64) (&sM[0][0])[e] = M[e];
  for (int e = lane; e < 4 * (kSteps + 1) * 12; e += 64) (&sR[0][0])[e] = r[e];
  __syncthreads();
  const int ii = i < 12 ? i : 11;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (lane < ACTIVE) {
  double x = sR[row][ii];
  sU[row][ii] = x;
  double m[12], mn[12], rk = sR[row][12 + ii], rn = 0.0;
#pragma unroll
  for (int c = 0; c < 12; c++) m[c] = sM[row][MIDX(ii, c)];
#pragma unroll
  for (int k = 1; k <= kSteps; k++) {
#ifndef NOLOAD
    if (k < kSteps) {
      rn = sR[row][(k + 1) * 12 + ii];
#pragma unroll
      for (int c = 0; c < 12; c++) mn[c] = sM[row][k * kSlot + MIDX(ii, c)];
    }
#else
    rn = rk * 0.5;
#pragma unroll
    for (int c = 0; c < 12; c++) mn[c] = m[c];
#endif
    x = step12(rk, x, m);
#ifndef NOSTORE
    sU[row][k * 12 + ii] = x;
#endif
    rk = rn;
#pragma unroll
    for (int c = 0; c < 12; c++) m[c] = mn[c];
  }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  for (int e = lane; e < 4 * (kSteps + 1) * 12; e += 64) out[e] = (&sU[0][0])[e];
  if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
  std::vector<double> M(4 * kSteps * kSlot), r(4 * (kSteps + 1) * 12), out(r.size()), ref(r.size());
  for (size_t e = 0; e < M.size(); e++) M[e] = 0.3 * sin(0.37 * e + 1.0);
  for (size_t e = 0; e < r.size(); e++) r[e] = cos(0.11 * e);
  for (int p = 0; p < 4; p++) {
    double* u = &ref[p * (kSteps + 1) * 12];
    const double* rr = &r[p * (kSteps + 1) * 12];
    for (int c = 0; c < 12; c++) u[c] = rr[c];
    for (int k = 1; k <= kSteps; k++)
      for (int i = 0; i < 12; i++) {
        double s = rr[k * 12 + i];
        for (int c = 0; c < 12; c++) s += M[p * kSteps * kSlot + (k - 1) * kSlot + MIDX(i, c)] * u[(k - 1) * 12 + c];
        u[k * 12 + i] = s;
      }
  }
  double *dM, *dr, *dout; unsigned long long* dc;
  hipMalloc(&dM, M.size() * 8); hipMalloc(&dr, r.size() * 8); hipMalloc(&dout, r.size() * 8); hipMalloc(&dc, 8);
  hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dr, r.data(), r.size() * 8, hipMemcpyHostToDevice);
  unsigned long long c = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dM, dr, dout, dc);
    hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  }
  hipMemcpy(out.data(), dout, r.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, mx = 0;
  for (size_t e = 0; e < r.size(); e++) { err = fmax(err, fabs(out[e] - ref[e])); mx = fmax(mx, fabs(ref[e])); }
  printf("CS=%d TR=%d ACC=%d ACTIVE=%d dpp chain: %d steps, %llu memtime ticks total, %.1f per step; max err %.3e (max |u| %.3e)\n", CS, TR, ACC, ACTIVE, kSteps, c,
         (double)c / kSteps, err, mx);
  return err <= 1e-12 * mx ? 0 : 1;
}
