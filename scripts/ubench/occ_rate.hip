// Micro-benchmark: aggregate instruction throughput of one SIMD as a function of the wavefronts resident on it
// (1, 2, 4 per SIMD), for the instruction mixes of the MPC kernel's ADMM loop.  Answers "what would a second
// wavefront per SIMD buy" (the kernel runs one, because it needs all 512 registers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
constexpr int kLoops = 2000;
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, unsigned long long* cyc, double s) {
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i * 1e-3;
  __syncthreads();
  double x = threadIdx.x * 0.001 + s, m = 1.0000001;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, l0 = 0, l1 = 0;
  int v0 = threadIdx.x, v1 = 3;
  const unsigned p = (threadIdx.x & 63) * 8;  // byte offset into lds (the only LDS object: offset 0)
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kLoops; it++) {
    if (MODE == 0) {  // 64 dependent FP64 FMAs
      asm volatile(R16(R4("v_fma_f64 %0, %1, %2, %0\n\t")) : "+v"(a0) : "v"(x), "v"(m));
    } else if (MODE == 1) {  // 64 FP64 FMAs over 4 accumulators
      asm volatile(R16("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(m));
    } else if (MODE == 2) {  // 64 32-bit moves (dependent pairs)
      asm volatile(R16("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\t") : "+v"(v0), "+v"(v1));
    } else if (MODE == 3) {  // the loop's mix: per 16: 6 FP64 FMA (dependent), 6 moves, 2 LDS reads, 2 scalar
      asm volatile(R4("v_fma_f64 %0, %6, %7, %0\n\tv_mov_b32 %2, %3\n\tv_fma_f64 %0, %6, %7, %0\n\tv_mov_b32 %3, %2\n\t"
                      "ds_read_b64 %4, %8\n\tv_fma_f64 %1, %6, %7, %1\n\tv_mov_b32 %2, %3\n\ts_nop 0\n\t"
                      "v_fma_f64 %1, %6, %7, %1\n\tv_mov_b32 %3, %2\n\tds_read_b64 %5, %8 offset:512\n\tv_fma_f64 %0, %6, %7, %0\n\t"
                      "v_mov_b32 %2, %3\n\ts_nop 0\n\tv_fma_f64 %1, %6, %7, %1\n\tv_mov_b32 %3, %2\n\t")
                   "s_waitcnt lgkmcnt(0)\n\t"
                   : "+v"(a0), "+v"(a1), "+v"(v0), "+v"(v1), "=v"(l0), "=v"(l1) : "v"(x), "v"(m), "v"(p));
      a2 += l0 + l1;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + v0 + v1;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE>
void run(const char* name, double* d, unsigned long long* c) {
  for (int waves : {4, 8, 16}) {
    unsigned long long h[16];
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, d, c, 0.5); hipDeviceSynchronize(); }
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
    const double n = (MODE == 3 ? 65.0 : 64.0) * kLoops;
    printf("%-28s %2d waves/SIMD: %.2f clocks per instruction per wavefront, %.2f per SIMD instruction slot\n", name, waves / 4,
           mx / n, mx / n / (waves / 4));
  }
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 1024 * 8); hipMalloc(&c, 16 * 8);
  run<0>("dependent v_fma_f64", d, c);
  run<1>("4-accumulator v_fma_f64", d, c);
  run<2>("v_mov_b32", d, c);
  run<3>("loop mix (fma/mov/lds/salu)", d, c);
  return 0;
}
