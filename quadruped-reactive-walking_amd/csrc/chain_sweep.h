// Sweeps of the twisted block LDL' (12x12 blocks) on the FP64 VALU with DPP row broadcasts — gfx950 (MI355X).
// Shared by mpc_kernel.hip and the micro-benchmark scripts/ubench/chain_bench.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "qrw_device.h"

namespace qrw {

// LDS layout of the chain matrices.  Column stride 14 doubles: entry (i,c) at c*14+i -- the forward access (ds_read_b64, lane i
// reads (i,c)) and the transposed one (ds_read_b128, lane i reads two entries of row i of M') are free of bank conflicts WITHIN
// a chain.  ACROSS the two chains they are not: both chains are read by one instruction (chain A in DPP row 0, chain B in row 1,
// i.e. lanes 0-15 / 16-31 of the same bank group: ds_read_b64 is served in two groups of 32 lanes, ds_read_b128 in four groups
// of 16 -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 --, 64 banks of 4 bytes), and with the slot stride 168 every
// sweep read is a 2-way conflict: SQ_LDS_BANK_CONFLICT is a third of the kernel's LDS-array cycles.  A layout without them
// (slot stride 176 and a gap of 16 doubles in front of chain B's slots) was measured in round 4: the conflicts go (0.165 ->
// 0.029 of the wave cycles, profiles/r4_pmc_lds_contention.txt) and the launch time does not move (677 k control steps/s either
// way), for 1.1 KB more LDS per instance and a few dwords of scratch in the N = 32 instantiations.  Not shipped; the variant is
// scripts/experiments/slower_forms.patch (docs/HISTORY.md, round 4).  chain_gap / chain_slot are where it plugs in.
constexpr int kCol = 14;    // column stride of a chain matrix in LDS
constexpr int kSlot = 168;  // slot stride (12 * kCol)
__host__ __device__ constexpr int chain_gap(int m) { return 0; }  // doubles in front of chain B's slots
__host__ __device__ constexpr int chain_slot(int slot, int m) { return slot * kSlot + (slot >= m ? chain_gap(m) : 0); }
// row whose transposed entries a lane reads in the backward sweep (lanes 12-15 have none)
__device__ __forceinline__ int chain_row_t(int lane) {
  const int l = lane & 15;
  return (l < 12) ? l : 11;
}
__host__ __device__ constexpr int chain_lds_doubles(int slots, int m) { return slots * kSlot + chain_gap(m); }

// ---------------------------------------------------------------------------------------------------------
// 12x12 row-times-vector step on the FP64 VALU: returns r + sum_c m[c] * x(lane c of this 16-lane row).
// v_fmac_f64 is a VOP2 on gfx950 and its DPP form takes row_newbcast (the only DPP control FP64 ops accept), so the
// broadcast of the 12 source entries costs no extra instruction.  One accumulator: a dependent v_fmac_f64 issues
// every ~5.5 clocks for a lone wavefront, the same as independent ones (scripts/ubench/dpp_rate.hip).
// One asm statement: the compiler does not model the "VALU write -> DPP read" hazard inside inline asm, hence the
// leading s_nop 1 (x is usually produced by the instruction just before).
__device__ __forceinline__ double dpp_step12(double r, double x, const double (&m)[12]) {
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

// Twisted (two-ended) block LDL' of the block-tridiagonal state system: chain A eliminates steps 0..m-1 upwards,
// chain B steps N-1..m+1 downwards, both meet in the root step m = N/2.  Chain A runs in DPP row 0 (lanes 0..11
// hold vector entries 0..11 and one matrix row each), chain B in row 1, in the same instruction stream; rows 2,3
// shadow rows 0,1 and store nothing.  Storage is arranged so that both chains walk LDS in the same direction with
// the same stride (every access is one per-lane base register plus an immediate offset):
//   matrices (negated, column-major with column stride kCol = 14: entry (i,c) at c*14+i; with that stride both the
//   forward access -- 12 ds_read_b64, lanes consecutive -- and the transposed one -- 6 ds_read_b128 per lane, 112 B
//   apart -- are free of bank conflicts, 56 clocks per 12x12 operand, scripts/ubench/lds_rate.hip):
//     slot s < m : -N_{s+1},        N_k  = C_k Delta_{k-1}^-1        (k = 1..m)
//     slot s >= m: -Nt_{N-2-(s-m)}, Nt_k = C_{k+1}' Delta_{k+1}^-1   (k = m..N-2; slot m + N-2-k)
//     slot s at chain_slot(s, m) doubles from the base (see kSlot above)
//   vectors: step k lives at position pos(k) = k (k <= m), m + N - k (k > m); position N holds zeros.
// Forward:  A: u_k = r_k - N_k u_{k-1};  B: u_k = r_k - Nt_k u_{k+1};  root: u_m = r_m - N_m u_{m-1} - Nt_m u_{m+1}.
// In place in sX (u overwrites r).  NC > 0: compile-time N (fully unrolled), NC == 0: runtime N.
// Operands are fetched one step ahead into two alternating register buffers (the LDS counter tracks at most 15
// outstanding operations, a step needs 13 (forward) or 7 (backward)).
__host__ __device__ __forceinline__ int chain_pos(int k, int m, int N) { return (k <= m) ? k : m + N - k; }

struct ChainOp {
  double m[12];
  double r;
};
typedef double qrw_d2 __attribute__((ext_vector_type(2)));

// lanes 0..31 <-> lanes 32..63 of the same register (gfx950 v_permlane32_swap with both operands the same VGPR).
// The s_nop is required: without wait states after the VALU write of v the exchange reads stale data (measured).
__device__ __forceinline__ double swap_halves(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %0\n\tv_permlane32_swap_b32 %1, %1" : "+v"(lo), "+v"(hi));
  return __hiloint2double(hi, lo);
}

// 16-lane rows 0 <-> 1 and 2 <-> 3 of the same register (v_permlane16_swap with both operands the same VGPR)
__device__ __forceinline__ double swap_rows16(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %0\n\tv_permlane16_swap_b32 %1, %1" : "+v"(lo), "+v"(hi));
  return __hiloint2double(hi, lo);
}

// Paired sweeps (compile-time N with an even number of steps per chain): an LDS read moves 64 lanes whatever EXEC says
// (scripts/ubench/exec_rate.hip), and the chains only fill lanes 0..31.  So one set of reads fetches the operands of
// TWO consecutive steps -- lanes 0..31 those of step 2p+1, lanes 32..63 those of step 2p+2 -- and the running vector
// changes halves between the steps (two v_permlane32_swap): half the LDS instructions and half the LDS bandwidth of
// the one-step form, the same 12 FMAs per step in the same order (bit-identical results).
template <int NC>
__device__ __forceinline__ void chain_forward_paired(const double* sN, double* sX, double* sDump, int lane) {
  constexpr int N = NC, m = N >> 1, LA = m, LB = N - 1 - m, NP = LA / 2;
  static_assert(LA % 2 == 0 && LB == LA - 1, "paired sweeps need an even chain length");
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
  const int h = lane >> 5;
  const bool rw = (lane & 16) != 0;
  const bool own = (lane & 15) < 12;
  const double* pm = sN + (rw ? m * kSlot + chain_gap(m) : 0) + h * kSlot + i;  // pair p: + 2p*kSlot + c*kCol
  double* px = sX + (rw ? (m + 1) * 12 : 0) + i;                 // step t: + t*12
  const double* pr = px + h * 12;                                // rhs of pair p: + (2p+1)*12
  double* dump = sDump + i;
  // a step's result is stored AFTER it has changed halves (the exchange can then work in place: nothing else still
  // needs the unexchanged value), i.e. by the lanes of the half the step was NOT computed in
  double* ps_odd = (own && h == 1) ? px : dump;           // odd steps (computed in lanes 0..31), both chains
  double* ps_even = (own && h == 0) ? px : dump;          // even steps (computed in lanes 32..63), both chains
  double* ps_oddA = (own && h == 1 && !rw) ? px : dump;   // step LB (odd): chain A only
  double x = px[0], pB = 0.0;
  ChainOp b0, b1;
  auto fetch = [&](ChainOp& b, int p) {
    const double* q = pm + 2 * p * kSlot;
#pragma unroll
    for (int c = 0; c < 12; c++) b.m[c] = __hip_atomic_load(q + c * kCol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    b.r = pr[(2 * p + 1) * 12];
  };
  auto pair = [&](const ChainOp& b, int p) {
    const int t1 = 2 * p + 1, t2 = t1 + 1;
    x = dpp_step12(b.r, x, b.m);  // step t1, computed in lanes 0..31
    x = swap_halves(x);           // now in lanes 32..63
    if (t1 == LB) pB = x;
    (t1 < LB ? ps_odd : ps_oddA)[t1 * 12] = x;
    x = dpp_step12(b.r, x, b.m);  // step t2, computed in lanes 32..63
    if (t2 < LA) {
      x = swap_halves(x);         // now in lanes 0..31
      ps_even[t2 * 12] = x;
    }
  };
  fetch(b0, 0);
#pragma unroll
  for (int p = 0; p < NP; p += 2) {
    if (p + 1 < NP) fetch(b1, p + 1);
    __builtin_amdgcn_sched_barrier(0);
    pair(b0, p);
    if (p + 1 < NP) {
      if (p + 2 < NP) fetch(b0, p + 2);
      __builtin_amdgcn_sched_barrier(0);
      pair(b1, p + 1);
    }
  }
  // root: chain A's last step sits in lanes 32..43 (row 2), chain B's contribution (step LB, odd, exchanged) in lanes
  // 48..59 (row 3): a row exchange instead of an LDS round trip
  x += swap_rows16(pB);
  if (lane >= 32 && lane < 44) sX[m * 12 + i] = x;
}
template <int NC>
__device__ __forceinline__ void chain_backward_paired(const double* sN, double* sX, double* sDump, int lane) {
  constexpr int N = NC, m = N >> 1, LA = m, LB = N - 1 - m, NP = LA / 2;
  static_assert(LA % 2 == 0 && LB == LA - 1, "paired sweeps need an even chain length");
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
  const int h = lane >> 5;
  const bool rw = (lane & 16) != 0;
  const bool own = (lane & 15) < 12;
  const int ib = chain_row_t(lane);
  const double* pm = sN + ((rw ? (N - 1) : m) - LA - h) * kSlot + (rw ? chain_gap(m) : 0) + ib * kCol;  // pair p: + (LA-2p-1)*kSlot + c
  double* px = sX + ((rw ? N : m) - LA) * 12 + i;                             // step t: + (LA-t)*12
  const double* pr = px - h * 12;                                             // pair p: + (LA-2p-1)*12
  double* dump = sDump + i;
  double* ps_odd = (own && h == 1) ? px : dump;           // odd steps (stored after the exchange), both chains
  double* ps_even = (own && h == 0) ? px : dump;          // even steps t < LA (stored after the exchange), both chains
  double* ps_lastA = (own && h == 1 && !rw) ? px : dump;  // step LA (even, not exchanged): chain A only
  double x = sX[m * 12 + i];
  ChainOp b0, b1;
  auto fetch = [&](ChainOp& b, int p) {
    const qrw_d2* q = reinterpret_cast<const qrw_d2*>(pm + (LA - 2 * p - 1) * kSlot);
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const qrw_d2 v = q[c];
      b.m[2 * c] = v.x;
      b.m[2 * c + 1] = v.y;
    }
    b.r = pr[(LA - 2 * p - 1) * 12];
  };
  auto pair = [&](const ChainOp& b, int p) {
    const int t1 = 2 * p + 1, t2 = t1 + 1;
    x = dpp_step12(b.r, x, b.m);  // step t1, computed in lanes 0..31
    x = swap_halves(x);
    ps_odd[(LA - t1) * 12] = x;
    x = dpp_step12(b.r, x, b.m);  // step t2, computed in lanes 32..63
    if (t2 < LA) {
      x = swap_halves(x);
      ps_even[(LA - t2) * 12] = x;  // t2 < LA = LB + 1: both chains
    } else {
      ps_lastA[(LA - t2) * 12] = x;
    }
  };
  fetch(b0, 0);
#pragma unroll
  for (int p = 0; p < NP; p += 2) {
    if (p + 1 < NP) fetch(b1, p + 1);
    __builtin_amdgcn_sched_barrier(0);
    pair(b0, p);
    if (p + 1 < NP) {
      if (p + 2 < NP) fetch(b0, p + 2);
      __builtin_amdgcn_sched_barrier(0);
      pair(b1, p + 1);
    }
  }
}

// sDump: (NC/2 + 2) * 12 doubles of LDS that absorb the stores of lanes / steps that must not write (cheaper than
// switching EXEC around every store: a lone wavefront pays full issue time for each scalar instruction).
template <int NC>
__device__ __forceinline__ void chain_forward(const double* sN, double* sX, double* sDump, int Nrt, int lane) {
  if constexpr (NC > 0 && (NC / 2) % 2 == 0) {
    chain_forward_paired<NC>(sN, sX, sDump, lane);
    return;
  }
  const int N = NC ? NC : Nrt;
  const int m = N >> 1, LA = m, LB = N - 1 - m;
  if (LA == 0) return;
  // opaque to the optimiser: keeps the address arithmetic inside the ADMM loop (hoisted, it gets spilled to scratch)
  asm volatile("" : "+v"(lane));
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
  const bool rw = (lane & 16) != 0;
  const bool wr = (lane < 32) && ((lane & 15) < 12);
  const double* pm = sN + (rw ? m * kSlot + chain_gap(m) : 0) + i;  // step t: + (t-1)*kSlot + c*kCol
  double* px = sX + (rw ? (m + 1) * 12 : 0) + i;     // step t: + t*12 (t = 0: the chain's first vector)
  double* ps_early = wr ? px : sDump + i;            // stores of steps t < LB
  double* ps_late = (wr && !rw) ? px : sDump + i;    // stores of steps LB <= t < LA (chain A only)
  double x = px[0], pB = 0.0;
  ChainOp b0, b1;
  auto fetch = [&](ChainOp& b, int t) {
    const double* q = pm + (t - 1) * kSlot;
#pragma unroll
    for (int c = 0; c < 12; c++) {
      // relaxed atomic load: stays a ds_read_b64 (the merged ds_read2_b64 form a plain load compiles to takes twice as long
      // per byte; that variant is in scripts/experiments/slower_forms.patch)
      b.m[c] = __hip_atomic_load(q + c * kCol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    b.r = px[t * 12];  // chain B, t = LB: position N (zeros); t > LB: unused
  };
  auto step = [&](const ChainOp& b, int t) {
    x = dpp_step12(b.r, x, b.m);
    if (t == LB) pB = x;
    if (t < LA) (t < LB ? ps_early : ps_late)[t * 12] = x;
  };
  fetch(b0, 1);
#pragma unroll
  for (int t = 1; t <= LA; t += 2) {  // operands of step t+1 are requested before the 12 FMAs of step t are issued
    if (t + 1 <= LA) fetch(b1, t + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above the FMAs (the scheduler sinks them otherwise)
    step(b0, t);
    if (t + 1 <= LA) {
      if (t + 2 <= LA) fetch(b0, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      step(b1, t + 1);
    }
  }
  if (LB > 0) x += shfl(pB, lane + 16);
  if (lane < 12) sX[m * 12 + i] = x;
}
// Backward:  x_m = v_m;  A: x_k = v_k - N_{k+1}' x_{k+1} (k = m-1..0);  B: x_k = v_k - Nt_{k-1}' x_{k-1} (k = m+1..N-1).
// Transposed reads of the same slots (row i of M' is contiguous: 6 ds_read_b128).
template <int NC>
__device__ __forceinline__ void chain_backward(const double* sN, double* sX, double* sDump, int Nrt, int lane) {
  if constexpr (NC > 0 && (NC / 2) % 2 == 0) {
    chain_backward_paired<NC>(sN, sX, sDump, lane);
    return;
  }
  const int N = NC ? NC : Nrt;
  const int m = N >> 1, LA = m, LB = N - 1 - m;
  if (LA == 0) return;
  asm volatile("" : "+v"(lane));
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
  const bool rw = (lane & 16) != 0;
  const bool wr = (lane < 32) && ((lane & 15) < 12);
  // both chains walk downwards: step t reads slot (top - t) and vector position (top - t); addressed from the lowest
  // one so that every immediate offset is non-negative
  const int ib = chain_row_t(lane);
  const double* pm = sN + ((rw ? (N - 1) : m) - LA) * kSlot + (rw ? chain_gap(m) : 0) + ib * kCol;  // step t: + (LA-t)*kSlot + c
  double* px = sX + ((rw ? N : m) - LA) * 12 + i;                         // step t: + (LA-t)*12
  double* ps_early = wr ? px : sDump + i;            // steps t <= LB
  double* ps_late = (wr && !rw) ? px : sDump + i;    // steps t > LB (chain A only)
  double x = sX[m * 12 + i];
  ChainOp b0, b1;
  auto fetch = [&](ChainOp& b, int t) {
    const qrw_d2* q = reinterpret_cast<const qrw_d2*>(pm + (LA - t) * kSlot);
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const qrw_d2 v = q[c];
      b.m[2 * c] = v.x;
      b.m[2 * c + 1] = v.y;
    }
    b.r = px[(LA - t) * 12];
  };
  auto step = [&](const ChainOp& b, int t) {
    x = dpp_step12(b.r, x, b.m);
    (t <= LB ? ps_early : ps_late)[(LA - t) * 12] = x;
  };
  fetch(b0, 1);
#pragma unroll
  for (int t = 1; t <= LA; t += 2) {
    if (t + 1 <= LA) fetch(b1, t + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above the FMAs (the scheduler sinks them otherwise)
    step(b0, t);
    if (t + 1 <= LA) {
      if (t + 2 <= LA) fetch(b0, t + 2);
      __builtin_amdgcn_sched_barrier(0);
      step(b1, t + 1);
    }
  }
}

// ---- in-register Gauss-Jordan on the FP64 VALU: lane i of a 16-lane DPP row holds row i of a 12x12 matrix in m[0..11]
// One pivot: every row gets  m[c] += m_P[c] * f  with m_P[c] read from lane P by row_newbcast (f = -m[P]/pivot for the
// other rows, 1/pivot - 1 for row P itself), then column P is replaced.  12 pivots invert the matrix in place.
#define QRW_GJ_FM(C) "v_fmac_f64_dpp %" #C ", %" #C ", %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
template <int P>
__device__ __forceinline__ void gj_pivot(double (&m)[12], int i) {
  double piv = 0.0;
  const double one = 1.0;
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(piv) : "v"(m[P]), "v"(one), "n"(P));
  const double d = fast_rcp(piv);
  const double f = (i == P) ? (d - 1.0) : (-m[P] * d);
  asm("s_nop 1\n\t" QRW_GJ_FM(0) QRW_GJ_FM(1) QRW_GJ_FM(2) QRW_GJ_FM(3) QRW_GJ_FM(4) QRW_GJ_FM(5) QRW_GJ_FM(6) QRW_GJ_FM(7)
      QRW_GJ_FM(8) QRW_GJ_FM(9) QRW_GJ_FM(10) QRW_GJ_FM(11)
      : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]),
        "+v"(m[9]), "+v"(m[10]), "+v"(m[11])
      : "v"(f), "n"(P));
  m[P] = (i == P) ? d : f;
}
#undef QRW_GJ_FM
__device__ __forceinline__ void gj_invert12(double (&m)[12], int i) {
  gj_pivot<0>(m, i); gj_pivot<1>(m, i); gj_pivot<2>(m, i); gj_pivot<3>(m, i); gj_pivot<4>(m, i); gj_pivot<5>(m, i);
  gj_pivot<6>(m, i); gj_pivot<7>(m, i); gj_pivot<8>(m, i); gj_pivot<9>(m, i); gj_pivot<10>(m, i); gj_pivot<11>(m, i);
}
// acc[c] += m_J[c] * coef for c = 0..11 (row J of the matrix held across the lanes, coef per lane)
template <int J>
__device__ __forceinline__ void row_bcast_fma(double (&acc)[12], const double (&m)[12], double coef) {
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %12, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %13, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %14, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %15, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %16, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %5, %17, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %18, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %7, %19, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %8, %20, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %9, %21, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %10, %22, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %11, %23, %24 row_newbcast:%25 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
        "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])
      : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]),
        "v"(m[10]), "v"(m[11]), "v"(coef), "n"(J));
}
// value of the same register six lanes below in the 16-lane row (0 where there is none)
__device__ __forceinline__ double row_shr6(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x116, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x116, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

}  // namespace qrw
