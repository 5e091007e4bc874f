// Round 4: semantics of the DPP controls wbc16_kernel relies on (16 lanes per robot instance), checked on the device:
// row_newbcast on a 32-bit move, row_shl / row_shr by 4 with a bank mask (bank = quad of the 16-lane row) = "lane ^ 4",
// row_ror:8 = "lane ^ 8", and v_fmac_f64_dpp with row_newbcast.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int RM, int BM>
__device__ __forceinline__ int dppmov(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, RM, BM, false); }
__global__ void k(int* out, double* outd) {
  const int lane = threadIdx.x;
  int v = 100 + lane;
  out[0 * 64 + lane] = dppmov<0x150 + 5, 0xF, 0xF>(v, v);                     // row_newbcast:5
  int t = dppmov<0x104, 0xF, 0x5>(v, v);                                      // row_shl:4, banks 0 and 2
  t = dppmov<0x114, 0xF, 0xA>(t, v);                                          // row_shr:4, banks 1 and 3
  out[1 * 64 + lane] = t;                                                     // expect value of lane ^ 4
  out[2 * 64 + lane] = dppmov<0x128, 0xF, 0xF>(v, v);                         // row_ror:8: expect lane ^ 8
  double acc = 1000.0, x = (double)lane, m = 2.0;
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m));
  outd[lane] = acc;                                                           // 1000 + 2 * (lane 9 of the row)
}
int main() {
  int* d; double* dd; hipMalloc(&d, 3 * 64 * 4); hipMalloc(&dd, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dd); hipDeviceSynchronize();
  int h[192]; double hd[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const int row = l & ~15;
    if (h[l] != 100 + row + 5) bad++;
    if (h[64 + l] != 100 + (l ^ 4)) bad++;
    if (h[128 + l] != 100 + (l ^ 8)) bad++;
    if (hd[l] != 1000.0 + 2.0 * (row + 9)) bad++;
  }
  printf("dpp_row_probe: %d mismatches; lane 21: bcast5 %d xor4 %d xor8 %d fmac %.1f\n", bad, h[21], h[64 + 21], h[128 + 21], hd[21]);
  return bad ? 1 : 0;
}
