// Micro-benchmark: LDS read throughput of the access patterns used by the chain sweeps (one wavefront).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R6(X) X X X X X X
#define R16(X) X X X X X X X X X X X X X X X X
__global__ void k(double* out, unsigned long long* cyc) {
  __shared__ double lds[6144];
  for (int i = threadIdx.x; i < 6144; i += 64) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x, i = (lane & 15) < 12 ? (lane & 15) : 11, rw = (lane >> 4) & 1;
  unsigned a_row = (unsigned)(size_t)&lds[rw * 2304 + i * 12];   // contiguous 12 doubles per lane (96 B stride)
  unsigned a_col = (unsigned)(size_t)&lds[rw * 2304 + i];        // strided: entry c at +c*12 doubles
  unsigned a_row13 = (unsigned)(size_t)&lds[rw * 2304 + i * 13];
  unsigned a_row14 = (unsigned)(size_t)&lds[rw * 2304 + i * 14];
  unsigned long long t[9];
  double acc = 0;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v0, v1, v2, v3, v4, v5;
#define TIME(IDX, ASM, ADDR)                                                            \
  t[IDX] = __builtin_amdgcn_s_memtime();                                                \
  asm volatile(R16(ASM) "s_waitcnt lgkmcnt(0)\n\t"                                      \
               : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5)        \
               : "v"(ADDR) : "memory");                                                 \
  acc += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v0.y + v1.y + v2.y + v3.y + v4.y + v5.y;
  // 6 x ds_read_b128 (contiguous row), 16 repetitions
  TIME(0, "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\tds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t", a_row)
  // 6 x ds_read2_b64 strided by 12 doubles
  TIME(1, "ds_read2_b64 %0, %6 offset0:0 offset1:12\n\tds_read2_b64 %1, %6 offset0:24 offset1:36\n\tds_read2_b64 %2, %6 offset0:48 offset1:60\n\tds_read2_b64 %3, %6 offset0:72 offset1:84\n\tds_read2_b64 %4, %6 offset0:96 offset1:108\n\tds_read2_b64 %5, %6 offset0:120 offset1:132\n\t", a_col)
  // 6 x ds_read2_b64 contiguous (adjacent pairs)
  TIME(2, "ds_read2_b64 %0, %6 offset0:0 offset1:1\n\tds_read2_b64 %1, %6 offset0:2 offset1:3\n\tds_read2_b64 %2, %6 offset0:4 offset1:5\n\tds_read2_b64 %3, %6 offset0:6 offset1:7\n\tds_read2_b64 %4, %6 offset0:8 offset1:9\n\tds_read2_b64 %5, %6 offset0:10 offset1:11\n\t", a_row)
  // same, rows padded to 13 doubles
  TIME(3, "ds_read2_b64 %0, %6 offset0:0 offset1:1\n\tds_read2_b64 %1, %6 offset0:2 offset1:3\n\tds_read2_b64 %2, %6 offset0:4 offset1:5\n\tds_read2_b64 %3, %6 offset0:6 offset1:7\n\tds_read2_b64 %4, %6 offset0:8 offset1:9\n\tds_read2_b64 %5, %6 offset0:10 offset1:11\n\t", a_row13)
  TIME(4, "ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\tds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t", a_row14)
  t[5] = __builtin_amdgcn_s_memtime();
  // 12 x ds_read_b64 strided by 12 doubles
  double w0, w1, w2, w3, w4, w5;
  unsigned long long t5a = __builtin_amdgcn_s_memtime();
  asm volatile(R16("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:96\n\tds_read_b64 %2, %6 offset:192\n\tds_read_b64 %3, %6 offset:288\n\tds_read_b64 %4, %6 offset:384\n\tds_read_b64 %5, %6 offset:480\n\t"
                   "ds_read_b64 %0, %6 offset:576\n\tds_read_b64 %1, %6 offset:672\n\tds_read_b64 %2, %6 offset:768\n\tds_read_b64 %3, %6 offset:864\n\tds_read_b64 %4, %6 offset:960\n\tds_read_b64 %5, %6 offset:1056\n\t")
               "s_waitcnt lgkmcnt(0)\n\t"
               : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5) : "v"(a_col) : "memory");
  unsigned long long t5b = __builtin_amdgcn_s_memtime();
  acc += w0 + w1 + w2 + w3 + w4 + w5;
  // 12 x ds_read_b64, rows padded to 13 doubles, lane reads its own row (transposed direction)
  unsigned long long t6a = __builtin_amdgcn_s_memtime();
  asm volatile(R16("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %6 offset:16\n\tds_read_b64 %3, %6 offset:24\n\tds_read_b64 %4, %6 offset:32\n\tds_read_b64 %5, %6 offset:40\n\t"
                   "ds_read_b64 %0, %6 offset:48\n\tds_read_b64 %1, %6 offset:56\n\tds_read_b64 %2, %6 offset:64\n\tds_read_b64 %3, %6 offset:72\n\tds_read_b64 %4, %6 offset:80\n\tds_read_b64 %5, %6 offset:88\n\t")
               "s_waitcnt lgkmcnt(0)\n\t"
               : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5) : "v"(a_row13) : "memory");
  unsigned long long t6b = __builtin_amdgcn_s_memtime();
  acc += w0 + w1 + w2 + w3 + w4 + w5;
  // 12 x ds_read_b64 strided, only lanes 0..31 active
  unsigned long long t7a = 0, t7b = 0;
  if (lane < 32) {
    t7a = __builtin_amdgcn_s_memtime();
    asm volatile(R16("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:96\n\tds_read_b64 %2, %6 offset:192\n\tds_read_b64 %3, %6 offset:288\n\tds_read_b64 %4, %6 offset:384\n\tds_read_b64 %5, %6 offset:480\n\t"
                     "ds_read_b64 %0, %6 offset:576\n\tds_read_b64 %1, %6 offset:672\n\tds_read_b64 %2, %6 offset:768\n\tds_read_b64 %3, %6 offset:864\n\tds_read_b64 %4, %6 offset:960\n\tds_read_b64 %5, %6 offset:1056\n\t")
                 "s_waitcnt lgkmcnt(0)\n\t"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5) : "v"(a_col) : "memory");
    t7b = __builtin_amdgcn_s_memtime();
    acc += w0 + w1 + w2 + w3 + w4 + w5;
  }
  out[lane] = acc;
  if (lane == 0) { for (int q = 0; q < 4; q++) cyc[q] = t[q + 1] - t[q]; cyc[4] = t5b - t5a; cyc[5] = t6b - t6a; cyc[6] = t7b - t7a; cyc[7] = t[5] - t[4]; }
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 64);
  unsigned long long h[8];
  for (int r = 0; r < 3; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c); hipDeviceSynchronize(); }
  hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  printf("ticks per 12-double row fetch: 6xb128 contiguous %.1f | 6xread2_b64 strided %.1f | 6xread2_b64 contiguous %.1f | same, pad 13 %.1f | 12xb64 strided %.1f | 12xb64 own row pad 13 %.1f | 12xb64 strided 32 lanes %.1f | 6xb128 own row pad 14 %.1f\n",
         h[0] / 16.0, h[1] / 16.0, h[2] / 16.0, h[3] / 16.0, h[4] / 16.0, h[5] / 16.0, h[6] / 16.0, h[7] / 16.0);
  return 0;
}
