// Micro-benchmark of the hand-scheduled chain sweep variants (cycles per step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LOADB(A0, A1, A2, TLO, THI) \
  "ds_read_b64 " A0 ", %0\n\tds_read_b64 " A1 ", %1\n\tds_read_b64 " A2 ", %2\n\t" \
  "ds_read2_b64 " TLO ", %3 offset1:4\n\tds_read_b64 " THI ", %3 offset:64\n\t" \
  "v_add_u32 %0, %5, %0\n\tv_add_u32 %1, %5, %1\n\tv_add_u32 %2, %5, %2\n\tv_add_u32 %3, %6, %3\n\t"
#ifndef NOPS
#define NOPS "s_nop 15\n\ts_nop 2\n\t"
#endif
#ifndef VARIANT
#define VARIANT 0
#endif
#if VARIANT == 0
#define STEPB(TC, A0, A1, A2, P01, P23, P45, LOADNEXT) \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A0 ", " P01 ", " TC "\n\t" \
  "ds_write2_b64 %4, " P01 ", " P23 " offset1:4\n\tds_write_b64 %4, " P45 " offset:64\n\tv_add_u32 %4, %6, %4\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A1 ", " P23 ", " TC "\n\t" LOADNEXT \
  "v_mfma_f64_16x16x4_f64 " TC ", " A2 ", " P45 ", " TC "\n\t" NOPS
#elif VARIANT == 1  /* no LDS traffic at all: pure MFMA + nops */
#define STEPB(TC, A0, A1, A2, P01, P23, P45, LOADNEXT) \
  "v_mfma_f64_16x16x4_f64 " TC ", " A0 ", " P01 ", " TC "\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A1 ", " P23 ", " TC "\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A2 ", " P45 ", " TC "\n\t" NOPS
#elif VARIANT == 2  /* loads right after MFMA1, stores after MFMA2 */
#define STEPB(TC, A0, A1, A2, P01, P23, P45, LOADNEXT) \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A0 ", " P01 ", " TC "\n\t" LOADNEXT \
  "v_mfma_f64_16x16x4_f64 " TC ", " A1 ", " P23 ", " TC "\n\t" \
  "ds_write2_b64 %4, " P01 ", " P23 " offset1:4\n\tds_write_b64 %4, " P45 " offset:64\n\tv_add_u32 %4, %6, %4\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A2 ", " P45 ", " TC "\n\t" NOPS
#elif VARIANT == 3  /* everything after MFMA3 instead of nops */
#define STEPB(TC, A0, A1, A2, P01, P23, P45, LOADNEXT) \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A0 ", " P01 ", " TC "\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A1 ", " P23 ", " TC "\n\t" \
  "v_mfma_f64_16x16x4_f64 " TC ", " A2 ", " P45 ", " TC "\n\t" \
  "ds_write2_b64 %4, " P01 ", " P23 " offset1:4\n\tds_write_b64 %4, " P45 " offset:64\n\tv_add_u32 %4, %6, %4\n\t" LOADNEXT \
  "s_nop 6\n\t"
#endif
__global__ void k_chain(double* out, unsigned long long* cyc, int steps_in) {
  __shared__ double lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (i % 152 >= 144) ? 0.0 : 1e-3 * ((i * 7) % 13 - 6);
  __syncthreads();
  const int lane = threadIdx.x, mrow = lane & 15, mq = lane >> 4;
  const int off0 = (mrow < 12) ? mq * 12 + mrow : 144;
  unsigned long long tot = 0;
  for (int rep = 0; rep < 64; rep++) {
    unsigned pa = (unsigned)(size_t)&lds[152 + off0], pa1 = pa + ((mrow < 12) ? 384 : 0), pa2 = pa + ((mrow < 12) ? 768 : 0);
    unsigned pc = (unsigned)(size_t)&lds[3000 + 12 + mq], ps = (unsigned)(size_t)&lds[3000 + mq];
    int dA = 152 * 8, dX = 96, steps = steps_in;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "v_mov_b32 v186, 0\n\tv_mov_b32 v187, 0\n\tv_mov_b32 v194, 0\n\tv_mov_b32 v195, 0\n\tv_mov_b32 v202, 0\n\tv_mov_b32 v203, 0\n\t"
        "ds_read2_b64 v[196:199], %4 offset1:4\n\tds_read_b64 v[200:201], %4 offset:64\n\t"
        "s_cmp_lt_i32 %7, 1\n\ts_cbranch_scc1 9f\n\t"
        LOADB("v[204:205]", "v[206:207]", "v[208:209]", "v[180:183]", "v[184:185]")
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        STEPB("v[180:187]", "v[204:205]", "v[206:207]", "v[208:209]", "v[196:197]", "v[198:199]", "v[200:201]",
              LOADB("v[210:211]", "v[212:213]", "v[214:215]", "v[188:191]", "v[192:193]"))
        "s_sub_u32 %7, %7, 1\n\ts_cmp_eq_u32 %7, 0\n\ts_cbranch_scc1 7f\n\t"
        STEPB("v[188:195]", "v[210:211]", "v[212:213]", "v[214:215]", "v[180:181]", "v[182:183]", "v[184:185]",
              LOADB("v[216:217]", "v[218:219]", "v[220:221]", "v[196:199]", "v[200:201]"))
        "s_sub_u32 %7, %7, 1\n\ts_cmp_eq_u32 %7, 0\n\ts_cbranch_scc1 8f\n\t"
        STEPB("v[196:203]", "v[216:217]", "v[218:219]", "v[220:221]", "v[188:189]", "v[190:191]", "v[192:193]",
              LOADB("v[204:205]", "v[206:207]", "v[208:209]", "v[180:183]", "v[184:185]"))
        "s_sub_u32 %7, %7, 1\n\ts_cmp_eq_u32 %7, 0\n\ts_cbranch_scc0 1b\n\t"
        "9:\n\ts_waitcnt lgkmcnt(0)\n\tds_write2_b64 %4, v[196:197], v[198:199] offset1:4\n\tds_write_b64 %4, v[200:201] offset:64\n\ts_branch 6f\n\t"
        "7:\n\ts_waitcnt lgkmcnt(0)\n\tds_write2_b64 %4, v[180:181], v[182:183] offset1:4\n\tds_write_b64 %4, v[184:185] offset:64\n\ts_branch 6f\n\t"
        "8:\n\ts_waitcnt lgkmcnt(0)\n\tds_write2_b64 %4, v[188:189], v[190:191] offset1:4\n\tds_write_b64 %4, v[192:193] offset:64\n\t"
        "6:\n\ts_waitcnt lgkmcnt(0)"
        : "+v"(pa), "+v"(pa1), "+v"(pa2), "+v"(pc), "+v"(ps), "+s"(dA), "+s"(dX), "+s"(steps)
        :
        : "memory", "scc", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191",
          "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205",
          "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219",
          "v220", "v221");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    tot += t1 - t0;
  }
  out[threadIdx.x] = lds[3000 + threadIdx.x];
  if (threadIdx.x == 0) cyc[0] = tot;
}
int main() {
  double* d; unsigned long long* c;
  hipMalloc((void**)&d, 64 * 8); hipMalloc((void**)&c, 8);
  for (int steps : {15, 30}) {
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, c, steps); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, c, steps); hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("variant %d steps %d: %.1f ticks/sweep, %.1f ticks/step\n", VARIANT, steps, (double)h / 64, (double)h / 64 / steps);
  }
  return 0;
}
