// Micro-benchmark: do LDS reads / FP64 FMAs of a lone wavefront get cheaper when only part of the wavefront is enabled
// in EXEC?  (The sweeps of the MPC kernel only need lanes 0..31.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R4(X) X X X X
#define R12(X) R4(X) R4(X) R4(X)
__global__ void k(double* out, unsigned long long* cyc, unsigned long long mask) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i * 1e-3;
  __syncthreads();
  const unsigned p = (threadIdx.x & 15) * 8;
  double a = 0, x = threadIdx.x, m = 1.0000001, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 q0 = {0, 0}, q1 = {0, 0};
  unsigned long long t[5], saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
  t[0] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 64; r++)
    asm volatile(R12("ds_read_b64 %0, %4 offset:0\n\tds_read_b64 %1, %4 offset:112\n\tds_read_b64 %2, %4 offset:224\n\tds_read_b64 %3, %4 offset:336\n\t")
                 "s_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3) : "v"(p));
  t[1] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 64; r++)
    asm volatile(R12("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:1344\n\t") "s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(p * 14));
  t[2] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 64; r++)
    asm volatile(R12(R4("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a) : "v"(x), "v"(m));
  t[3] = __builtin_amdgcn_s_memtime();
  asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
  out[threadIdx.x] = a + l0 + l1 + l2 + l3 + q0.x + q1.y;
  if (threadIdx.x == 0) { cyc[0] = t[1] - t[0]; cyc[1] = t[2] - t[1]; cyc[2] = t[3] - t[2]; }
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 64);
  const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0x0FFF0FFFull, 0xFFFFull, 0xFFFull};
  for (unsigned long long mk : masks) {
    unsigned long long h[3];
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, mk); hipDeviceSynchronize(); }
    hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
    printf("exec %016llx: ds_read_b64 %.2f clk each | ds_read_b128 %.2f | v_fmac_f64_dpp %.2f\n", mk, h[0] / (64.0 * 48), h[1] / (64.0 * 24), h[2] / (64.0 * 48));
  }
  return 0;
}
