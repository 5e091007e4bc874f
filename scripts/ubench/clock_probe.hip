// Round 4: does the chip hold its clock when every SIMD runs the MPC kernel's sweep mix (FP64 DPP FMAs + LDS reads)?
// Each workgroup (one wavefront, 35 KB of LDS: four per compute unit like the N = 16 kernel) runs the production sweeps REPS
// times and reports shader clocks (s_memtime) and 100 MHz ticks (s_memrealtime) of the same interval; the host adds the wall time.
// ./clock_probe <blocks>   e.g. 256 (one wavefront per compute unit), 1024 (four), 4096 (four rounds of four)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <chrono>
#include "chain_sweep.h"
using namespace qrw;
constexpr int N = 16, S = 16;
__global__ __launch_bounds__(64, 1) void k(const double* M, const double* r, double* out, unsigned long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) double sN[S * kSlot + 32];
  __shared__ double sX[(S + 2) * 12];
  __shared__ double sDump[(S / 2 + 2) * 12];
  __shared__ double pad[1500];  // ~35 KB per workgroup in all: four workgroups per compute unit
  const int lane = threadIdx.x;
  for (int e = lane; e < S * kSlot; e += 64) sN[e] = M[e];
  for (int e = lane; e < 1500; e += 64) pad[e] = 0.0;
  for (int e = lane; e < (S + 2) * 12; e += 64) sX[e] = r[e];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int rep = 0; rep < reps; rep++) {
    chain_forward<N>(sN, sX, sDump, N, lane);
    asm volatile("" ::: "memory");
    chain_backward<N>(sN, sX, sDump, N, lane);
    asm volatile("" ::: "memory");
    if ((rep & 15) == 15) for (int e = lane; e < (S + 2) * 12; e += 64) sX[e] = r[e];  // keep the values bounded
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
  if (lane < 12) out[blockIdx.x * 12 + lane] = sX[8 * 12 + lane] + pad[lane];
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 2000;
  std::vector<double> M(S * kSlot, 0.0), r((S + 2) * 12, 0.0);
  for (int s = 0; s < N - 1; s++)
    for (int i = 0; i < 12; i++)
      for (int c = 0; c < 12; c++) M[chain_slot(s, 8) + c * kCol + i] = 0.05 * sin(0.37 * (s * 144 + i * 12 + c) + 1.0);
  for (int e = 0; e < N * 12; e++) r[e] = cos(0.11 * e);
  double *dM, *dr, *dout; unsigned long long* dc;
  hipMalloc(&dM, M.size() * 8); hipMalloc(&dr, r.size() * 8); hipMalloc(&dout, (size_t)blocks * 12 * 8); hipMalloc(&dc, (size_t)blocks * 16);
  hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dr, r.data(), r.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dM, dr, dout, dc, 50);
  hipDeviceSynchronize();
  auto a = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dM, dr, dout, dc, reps);
  hipDeviceSynchronize();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
  std::vector<unsigned long long> c(2 * blocks);
  hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
  double st = 0, sr = 0;
  for (int b = 0; b < blocks; b++) { st += c[2 * b]; sr += c[2 * b + 1]; }
  printf("blocks %5d reps %d: wall %.3f ms; per workgroup mean %.0f s_memtime ticks, %.0f 100-MHz ticks -> %.3f GHz if s_memtime counts shader clocks; %.1f ticks per sweep pair\n",
         blocks, reps, wall * 1e3, st / blocks, sr / blocks, (st / sr) * 0.1, st / blocks / reps);
  return 0;
}
