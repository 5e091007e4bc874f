// Micro-benchmark + numerical check of the production sweeps (csrc/chain_sweep.h) in isolation: one wavefront, N = 16,
// random chain matrices in the production LDS layout; compares with a scalar host evaluation of the same recursions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "chain_sweep.h"
using namespace qrw;
constexpr int N = 16, S = 16;
__global__ void k(const double* M, const double* r, double* out, unsigned long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) double sN[S * kSlot];
  __shared__ double sX[(S + 2) * 12];
  __shared__ double sDump[(S / 2 + 2) * 12];
  const int lane = threadIdx.x;
  for (int e = lane; e < S * kSlot; e += 64) sN[e] = M[e];
  unsigned long long tf = 0, tb = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int e = lane; e < (S + 2) * 12; e += 64) sX[e] = r[e];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    chain_forward<N>(sN, sX, sDump, N, lane);
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    chain_backward<N>(sN, sX, sDump, N, lane);
    __syncthreads();
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    tf += t1 - t0; tb += t2 - t1;
  }
  for (int e = lane; e < (S + 2) * 12; e += 64) out[e] = sX[e];
  if (lane == 0) { cyc[0] = tf / reps; cyc[1] = tb / reps; }
}
int main() {
  const int m = N / 2;
  std::vector<double> M(S * kSlot, 0.0), r((S + 2) * 12, 0.0), out(r.size()), ref(r.size());
  for (int s = 0; s < N - 1; s++)
    for (int i = 0; i < 12; i++)
      for (int c = 0; c < 12; c++) M[s * kSlot + c * kCol + i] = 0.25 * sin(0.37 * (s * 144 + i * 12 + c) + 1.0);
  for (int k = 0; k < N; k++)
    for (int i = 0; i < 12; i++) r[chain_pos(k, m, N) * 12 + i] = cos(0.11 * (k * 12 + i));
  // host reference: forward, then (no middle phase) backward on the same data
  auto Mat = [&](int slot, int i, int c) { return M[slot * kSlot + c * kCol + i]; };
  std::vector<std::vector<double>> u(N, std::vector<double>(12));
  for (int k = 0; k < N; k++) for (int i = 0; i < 12; i++) u[k][i] = r[chain_pos(k, m, N) * 12 + i];
  for (int k = 1; k < m; k++) for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(k - 1, i, c) * u[k - 1][c]; u[k][i] = s; }
  for (int k = N - 2; k > m; k--) { std::vector<double> t(12); for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(m + N - 2 - k, i, c) * u[k + 1][c]; t[i] = s; } u[k] = t; }
  { std::vector<double> t(12); for (int i = 0; i < 12; i++) { double s = u[m][i]; for (int c = 0; c < 12; c++) s += Mat(m - 1, i, c) * u[m - 1][c]; double s2 = 0; for (int c = 0; c < 12; c++) s2 += Mat(m + N - 2 - m, i, c) * u[m + 1][c]; t[i] = s + s2; } u[m] = t; }
  for (int k = m - 1; k >= 0; k--) { std::vector<double> t(12); for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(k, c, i) * u[k + 1][c]; t[i] = s; } u[k] = t; }
  for (int k = m + 1; k < N; k++) { std::vector<double> t(12); for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(m + N - 2 - (k - 1), c, i) * u[k - 1][c]; t[i] = s; } u[k] = t; }
  double *dM, *dr, *dout; unsigned long long* dc;
  hipMalloc(&dM, M.size() * 8); hipMalloc(&dr, r.size() * 8); hipMalloc(&dout, r.size() * 8); hipMalloc(&dc, 16);
  hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dr, r.data(), r.size() * 8, hipMemcpyHostToDevice);
  unsigned long long c[2];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dM, dr, dout, dc, 1);
  hipDeviceSynchronize();
  hipMemcpy(out.data(), dout, r.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, mx = 0;
  for (int kk = 0; kk < N; kk++) for (int i = 0; i < 12; i++) { err = fmax(err, fabs(out[chain_pos(kk, m, N) * 12 + i] - u[kk][i])); mx = fmax(mx, fabs(u[kk][i])); }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dM, dr, dout, dc, 50);
  hipDeviceSynchronize();
  hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  printf("chain sweeps N=%d: forward %llu ticks (%.1f/step), backward %llu ticks (%.1f/step); max err %.3e (max |x| %.3e)\n", N, c[0],
         c[0] / 8.0, c[1], c[1] / 8.0, err, mx);
  return err <= 1e-12 * mx ? 0 : 1;
}
