// N = 32 horizon (two wavefronts per instance): the block-tridiagonal state system solved by a ONE-LEVEL NESTED DISSECTION
// around horizon step 16, so that BOTH wavefronts sweep -- gfx950 (MI355X).  Used by mpc_kernel.hip (NW = 2, N = 32).
//
// The twisted (two-ended) factorisation of chain_sweep.h runs both of its chains in DPP rows 0 / 1 of ONE instruction
// stream, so at N = 32 wavefront 0 swept 16 + 16 dependent steps per ADMM iteration while wavefront 1 idled (58 % of the
// iteration, profiles/r2_mpc_phase_cycles_n32.txt).  Here the horizon is cut at step 16 (the global root):
//   left half   steps 0..15   wavefront 0: chain A 0 -> 7 (upwards), chain B 15 -> 9 (downwards), half root 8
//   right half  steps 17..31  wavefront 1: chain A 31 -> 25 (downwards, preceded by a dummy step so that both halves have
//                             the shape chain_*_paired<16> expects), chain B 17 -> 23 (upwards), half root 24
// Each half is EXACTLY the 16-step twisted system of the N = 16 kernel (same paired sweeps, 8 + 8 dependent steps), swept
// by its own wavefront on its own SIMD at the same time.  What couples the halves to step 16 is carried as FILL: the chain
// that starts next to step 16 (15 -> 9 on the left, 17 -> 23 on the right) and the half root it ends in have a non-zero
// block G_k = K[k,16] once their predecessor is eliminated,
//   G_15 = C_16',  G_k = (-Nt_k) G_{k+1}  (k = 14..8)        G_17 = C_17,  G_k = (-N_k) G_{k-1}  (k = 18..24)
// (C_k = K[k,k-1]; -Nt_k / -N_k are the negated chain matrices the sweeps use anyway).  With E_k = -Delta_k^-1 G_k
// (16 blocks of 12 x 12, LDS) the solve of K x = r is
//   forward sweeps of both halves (u), v_k = Delta_k^-1 u_k                         as before, per half
//   u_16 = r_16 + sum_k E_k' u_k,   x_16 = Delta_16^-1 u_16                          (Delta_16 = T_16 + sum_k G_k' E_k)
//   v_k += E_k x_16 on the 16 fill steps, backward sweeps of both halves             as before, per half
// An exact solve of the same system (tests/test_dissection_algebra.py restates it in numpy against a dense solve; the
// library's self-test checks this code against a dense host solve), so the ADMM iterate sequence is the one the twisted
// form produces up to rounding.  Per ADMM iteration: 3 workgroup barriers (boundary patch, fill contributions -> root, end of
// the backward sweeps) instead of ~10, and sweeps of 8 + 8 steps on both wavefronts instead of 16 + 16 on one.
#pragma once
#include <hip/hip_runtime.h>

#include "chain_sweep.h"
#include "qrw_device.h"

namespace qrw {

constexpr int kDisWSz = 36;          // 6 x 6 K^-1 per step (factor phase)
constexpr int kFillStride = 148;     // doubles between consecutive E_k (dense 12 x 12 row-major; 148: the eight fill quads of
                                     // a wavefront hit disjoint LDS banks in the column access of phase 1)
constexpr int kDisRootRhs = 34, kDisRootX = 35;  // vector positions of step 16: right-hand side / solution
constexpr int kDisRightPos = 17;     // first vector position of the right half (its dummy step: always zero)
constexpr int kDisRightSlot = 15;    // first chain slot of the right half (its dummy step's coupling: always zero)
constexpr int kDisRightBase = chain_slot(kDisRightSlot, 8);  // its offset in sN = the left half's slot 15 (what the left half's idle
                                                             // chain reads in its discarded last step); each half addresses its
                                                             // slots with chain_slot(s, 8) from its own base (chain_sweep.h)

// LDS of one instance: 81 360 bytes, two instances per compute unit (160 KB); 83 536 with -DQRW_BANK_FREE_LAYOUT=1 (chain_sweep.h).
struct alignas(16) MpcLdsDis {
  double sN[kDisRightBase + chain_lds_doubles(15, 8)];  // chain matrices: left half slots 0..14, right half from kDisRightBase
                                         // (its slot 0: the zero matrix both halves start their chains from)
  double sFill[15 * kFillStride + 144];  // E_k, fill index f = k - 8 (k = 8..15), k - 9 (k = 17..24).  MUST follow sN: the
                                         // right half's idle read (its slot 15) lands in E_8 -- finite, result discarded
  double sX[36 * 12];                    // vectors: left half positions 0..16 (16: zeros), right 17..33 (17, 33: zeros), 34, 35
  double sC[12 * 16];                    // fill contributions [entry][fill index].  sX + sC: hand-off buffers of the factor phase
  double sRootInv[kSlot];                // Delta_16^-1 in chain-matrix layout
  double sEb[4][12];                     // k = 15 <-> 16 exchange: [0], [1] the hot loop's right-hand-side patch, [2] next, [3] prev
  double sW[32 * kDisWSz];               // K_k^-1 per step (factor phase); its head doubles as the sweeps' store dump
  double sOm[32 * 12];                   // omega_D per step (factor phase)
  double sDg[32 * 12];                   // c*w + sigma/Dx^2 per step (factor phase)
  double sRed[4];
  unsigned long long sBal[2];
  double sPre[8];
};
static_assert(sizeof(MpcLdsDis) <= (QRW_BANK_FREE_LAYOUT ? 84 * 1024 : 81920), "two N = 32 instances per compute unit need <= 80 KB each");
static_assert(36 * 12 + 12 * 16 >= 4 * 144, "the four hand-off buffers of the factor phase overlay sX + sC");
static_assert(32 * kDisWSz >= (16 / 2 + 2) * 12, "the sweeps' dump area overlays sW");

// vector position of horizon step k (solution side; step 16's right-hand side sits at kDisRootRhs)
__host__ __device__ __forceinline__ int dis_pos(int k) {
  if (k <= 15) return (k <= 8) ? k : 24 - k;
  if (k == 16) return kDisRootX;
  const int kap = 32 - k;  // the right half walks the horizon backwards: local step 0 is the dummy, 8 the half root (k = 24)
  return kDisRightPos + ((kap <= 8) ? kap : 24 - kap);
}
__host__ __device__ __forceinline__ bool dis_is_fill(int k) { return k >= 8 && k <= 24 && k != 16; }
__host__ __device__ __forceinline__ int dis_fill_index(int k) { return (k < 16) ? k - 8 : k - 9; }

// ---- k <-> k+-1 neighbour values: inside a wavefront by ds_bpermute, across the wavefront boundary (15 <-> 16) through LDS.
// Two workgroup barriers each: for the set-up, the factorisations and the termination check; the ADMM loop itself patches
// the boundary terms into the right-hand side instead (mpc_kernel.hip).
__device__ __forceinline__ void nb_next_dis(const double v[3], double same[3], double shifted[3], double* sE12, int k, int j,
                                            int lane) {
  if (k == 16) {
#pragma unroll
    for (int t = 0; t < 3; t++) sE12[3 * j + t] = v[t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; t++) { same[t] = shfl(v[t], lane + 4); shifted[t] = shfl(v[t], lane + 2); }
  if (k == 15) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
      same[t] = sE12[3 * j + t];
      if (j >= 2) shifted[t] = sE12[3 * (j - 2) + t];
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void nb_prev_dis(const double v[3], double same[3], double shifted[3], double* sE12, int k, int j,
                                            int lane) {
  if (k == 15) {
#pragma unroll
    for (int t = 0; t < 3; t++) sE12[3 * j + t] = v[t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; t++) { same[t] = shfl(v[t], lane - 4); shifted[t] = shfl(v[t], lane - 2); }
  if (k == 16) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
      same[t] = sE12[3 * j + t];
      if (j < 2) shifted[t] = sE12[3 * (j + 2) + t];
    }
  }
  __syncthreads();
}

// acc += M * coefs: acc_i[c] += sum_J coef_i[J] * M_J[c]   (lane i of a DPP row holds row i of acc, M and its own coefs)
__device__ __forceinline__ void row_matmul_acc(double (&acc)[12], const double (&M)[12], const double (&coef)[12]) {
  row_bcast_fma<0>(acc, M, coef[0]); row_bcast_fma<1>(acc, M, coef[1]); row_bcast_fma<2>(acc, M, coef[2]);
  row_bcast_fma<3>(acc, M, coef[3]); row_bcast_fma<4>(acc, M, coef[4]); row_bcast_fma<5>(acc, M, coef[5]);
  row_bcast_fma<6>(acc, M, coef[6]); row_bcast_fma<7>(acc, M, coef[7]); row_bcast_fma<8>(acc, M, coef[8]);
  row_bcast_fma<9>(acc, M, coef[9]); row_bcast_fma<10>(acc, M, coef[10]); row_bcast_fma<11>(acc, M, coef[11]);
}

// Factorisation of the dissected system.  Per wavefront the twisted block LDL' of its half exactly as chain_factorize
// (mpc_kernel.hip) does it for a whole horizon -- DPP row 0 and row 1 each eliminate one chain, lane i of a row holding row
// i of the step's 12 x 12 block, in-register Gauss-Jordan, Delta^-1 rows handed to the quad that owns the step, the negated
// coupling matrices written to the chain slots -- plus, on the chain that starts next to step 16 (row 1 in both
// wavefronts) and the half root: the fill block G_k, E_k = -Delta_k^-1 G_k (into sFill) and the half's contribution
// sum_k G_k' E_k to Delta_16.  Then Delta_16^-1 (into sRootInv).
//   wavefront 0: row 0 chain "up" 0..7, row 1 chain "down" 15..9 (fill), root 8
//   wavefront 1: row 0 chain "down" 31..25, row 1 chain "up" 17..23 (fill), root 24      (one idle step first: both
//                wavefronts run the same nine rounds and meet at the same barriers)
// Reads sW / sOm / sDg of all 32 steps; uses sX + sC as hand-off buffers (the zero vectors of sX are restored at the end).
// Register budget: this runs inside the ADMM loop (rho updates) with the whole loop state live, so only R (the Schur
// contribution) stays in registers across the rounds: the previous coupling row is re-read from the chain slot it was just
// written to, the fill block G waits in the E slot of the half root (f = 0 / 15, written last), and the Delta^-1 rows go
// to the owner through SetDi(t, c, value) (accumulation registers in the kernel) instead of through an array.
template <typename LdsT, typename SetFn>
__device__ __forceinline__ void dis_factorize(LdsT& L, double dt, int tid, int k, int j, SetFn&& SetDi) {
  const int lane = tid & 63, wv = tid >> 6;
  const bool rowB = (lane & 16) != 0;
  const bool wlane = lane < 32 && (lane & 15) < 12;  // rows 2, 3 shadow rows 0, 1 and store nothing
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
  const int i6 = (i >= 6) ? i - 6 : i;
  const bool down = rowB != (wv == 1);  // chain direction of this row: down = towards smaller k ("chain B" formulas)
  const double sA = down ? 0.0 : 1.0, sB = down ? 1.0 : 0.0;
  const double hi6 = (i >= 6) ? 1.0 : 0.0, lo6 = (i < 6) ? 1.0 : 0.0;
  double* hand = L.sX + (2 * wv + (rowB ? 1 : 0)) * 144;
  double* slots = L.sN + wv * kDisRightBase;
  double* pG = L.sFill + ((wv == 0) ? 0 : 15) * kFillStride;  // home of this wavefront's fill block G between the rounds
  double R[12];
#pragma unroll
  for (int c = 0; c < 12; c++) R[c] = 0.0;
  int prev_off = kDisRightBase - wv * kDisRightBase;  // offset of the zero matrix (the right half's slot 0) until the chain has made one
  {
    double G[12];
#pragma unroll
    for (int c = 0; c < 12; c++) G[c] = 0.0;
    if (rowB) {
    // fill block of the chain's first step: wavefront 0  G_15 = C_16',  wavefront 1  G_17 = C_17, with
    // C_k = K[k,k-1] = -[[Om_k, dt Om_k], [0, W_k]]  (Om_k: omega_D of the position rows of step k, W_k = K_k^-1)
    const int kc = (wv == 0) ? 16 : 17;
    const double om = L.sOm[kc * 12 + i6];
#pragma unroll
    for (int c = 0; c < 12; c++) {
      double v = 0.0;
      if (i < 6) {
        if (c == i) v = -om;
        if (wv == 1 && c == i + 6) v = -dt * om;
      } else {
        if (wv == 0 && c == i - 6) v = -dt * om;
        if (c >= 6) v = -L.sW[kc * kDisWSz + i6 * 6 + (c - 6)];
      }
      G[c] = v;
    }
    }
    if (rowB && wlane) {
#pragma unroll
      for (int c = 0; c < 12; c++) pG[i * 12 + c] = G[c];
    }
  }
  for (int s = 0; s <= 8; s++) {
    const bool root = (s == 8);
    bool active;
    int kk;
    if (wv == 0) {
      active = root || !rowB || s < 7;
      kk = root ? 8 : (rowB ? 15 - s : s);
    } else {
      active = root || s >= 1;
      kk = root ? 24 : (rowB ? 16 + s : 32 - s);
    }
    if (!active) kk = (wv == 0) ? 8 : 24;
    const bool hasprev = root || (wv == 0 ? s > 0 : s > 1);
    const bool last = (kk == 31);
    const int kn = last ? kk : kk + 1;
    const double nl = last ? 0.0 : 1.0;
    double m[12];
    {
      // ---- Ttilde_kk, row i (as chain_factorize)
      const double omki = L.sOm[kk * 12 + i], omni = nl * L.sOm[kn * 12 + i], omn6 = nl * L.sOm[kn * 12 + i6];
      const double diag = L.sDg[kk * 12 + i] + lo6 * (omki + omni) + hi6 * (dt * dt * omn6);
#pragma unroll
      for (int c = 0; c < 12; c++) {
        double v = (c == i) ? diag : 0.0;
        if (c >= 6) v = (c - 6 == i) ? dt * omni : v;
        if (c < 6) v = (c + 6 == i) ? dt * omn6 : v;
        if (c >= 6) v += hi6 * (L.sW[kk * kDisWSz + i6 * 6 + (c - 6)] + nl * L.sW[kn * kDisWSz + i6 * 6 + (c - 6)]);
        m[c] = v;
      }
      // ---- Schur term of the step this chain eliminated just before: up  -N_kk C_kk',  down  -Nt_kk C_{kk+1}
      const int ks = down ? kn : kk;
      const double* omS = &L.sOm[ks * 12];
      const double* WS = &L.sW[ks * kDisWSz];
      double term[12], nn[12];  // nn: row i of the negated coupling matrix this chain made in its previous round
#pragma unroll
      for (int c = 0; c < 12; c++) nn[c] = slots[prev_off + c * kCol + i];
#pragma unroll
      for (int c = 0; c < 6; c++) term[c] = -omS[c] * nn[c] - sA * (dt * omS[c]) * nn[c + 6];
#pragma unroll
      for (int c = 6; c < 12; c++) {
        double v = -sB * (dt * omS[c - 6]) * nn[c - 6];
#pragma unroll
        for (int mm = 0; mm < 6; mm++) v -= WS[(c - 6) * 6 + mm] * nn[6 + mm];
        term[c] = v;
      }
      if (root) {  // the half root couples to both chains of its half
#pragma unroll
        for (int c = 0; c < 12; c++) {
          const double mine = term[c];
          const double other = shfl(mine, lane ^ 16);
          m[c] += mine + other;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 12; c++) m[c] += hasprev ? term[c] : 0.0;
      }
    }
    gj_invert12(m, i);
    if (active && wlane && !(root && rowB)) {
#pragma unroll
      for (int c = 0; c < 12; c++) hand[i * 12 + c] = m[c];
    }
    // ---- fill of this step: E = -Delta^-1 G into sFill, R += E' G (the half's part of Delta_16's Schur complement)
    {
      const bool fill_here = active && rowB;
      const int f = fill_here ? dis_fill_index(kk) : 0;
      double* pE = L.sFill + f * kFillStride;
      double E[12], ng[12], G[12];
#pragma unroll
      for (int c = 0; c < 12; c++) { E[c] = 0.0; ng[c] = -m[c]; G[c] = rowB ? pG[i * 12 + c] : 0.0; }
      asm volatile("" ::: "memory");  // G is in registers before E_root overwrites its home
      row_matmul_acc(E, G, ng);
      if (fill_here && wlane) {
#pragma unroll
        for (int c = 0; c < 12; c++) pE[i * 12 + c] = E[c];
      }
      asm volatile("" ::: "memory");  // the wavefront's own stores precede its loads (the LDS executes them in order)
      double col[12];
#pragma unroll
      for (int r = 0; r < 12; r++) col[r] = pE[r * 12 + i];  // column i of E_k
      double Rn[12];
#pragma unroll
      for (int c = 0; c < 12; c++) Rn[c] = R[c];
      row_matmul_acc(Rn, G, col);
#pragma unroll
      for (int c = 0; c < 12; c++) R[c] = fill_here ? Rn[c] : R[c];
    }
    __syncthreads();
    {
      // the quad that owns the eliminated step takes its three rows of Delta^-1
      int k0, k1;  // steps of row 0 / row 1 of MY wavefront in this round
      if (wv == 0) { k0 = root ? 8 : s; k1 = (!root && s < 7) ? 15 - s : -1; }
      else { k0 = root ? 24 : (s >= 1 ? 32 - s : -1); k1 = (!root && s >= 1) ? 16 + s : -1; }
      if (k == k0 || k == k1) {
        const double* Mi = L.sX + (2 * wv + (k == k1 ? 1 : 0)) * 144;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
          for (int c = 0; c < 12; c++) SetDi(t, c, Mi[(3 * j + t) * 12 + c]);
      }
    }
    if (!root) {
      // ---- next coupling matrix, negated: up  -N_{kk+1} = -C_{kk+1} Delta_kk^-1,  down  -Nt_{kk-1} = -C_kk' Delta_kk^-1
      const double omki = L.sOm[kk * 12 + i], omni = L.sOm[kn * 12 + i];
      const double omk6 = L.sOm[kk * 12 + i6];
      const double* Wc = down ? &L.sW[kk * kDisWSz] : &L.sW[kn * kDisWSz];
      double nx[12], coef[6];
      const double own = lo6 * (down ? omki : omni);
#pragma unroll
      for (int c = 0; c < 12; c++) nx[c] = own * m[c];
#pragma unroll
      for (int mm = 0; mm < 6; mm++) {
        double cf = hi6 * Wc[i6 * 6 + mm];
        if (mm == i) cf = sA * (dt * omni);
        coef[mm] = cf;
      }
      row_bcast_fma<6>(nx, m, coef[0]); row_bcast_fma<7>(nx, m, coef[1]); row_bcast_fma<8>(nx, m, coef[2]);
      row_bcast_fma<9>(nx, m, coef[3]); row_bcast_fma<10>(nx, m, coef[4]); row_bcast_fma<11>(nx, m, coef[5]);
      {
        const double cb = sB * hi6 * (dt * omk6);
#pragma unroll
        for (int c = 0; c < 12; c++) nx[c] += cb * row_shr6(m[c]);
      }
      if (active && wlane) {
        // wavefront 0: up -> slot kk, down -> slot 23 - kk;  wavefront 1 (slots relative to the right half's first):
        // down -> 32 - kk, up -> kk - 9
        const int slot = (wv == 0) ? (down ? 23 - kk : kk) : (down ? 32 - kk : kk - 9);
#pragma unroll
        for (int c = 0; c < 12; c++) slots[chain_slot(slot, 8) + c * kCol + i] = nx[c];
      }
      {  // fill of the next step of this chain (or of the half root): G <- (negated coupling) G
        double Gn[12], G[12];
#pragma unroll
        for (int c = 0; c < 12; c++) { Gn[c] = 0.0; G[c] = rowB ? pG[i * 12 + c] : 0.0; }
        row_matmul_acc(Gn, G, nx);
        if (active && rowB && wlane) {
#pragma unroll
          for (int c = 0; c < 12; c++) pG[i * 12 + c] = Gn[c];
        }
      }
      if (active) prev_off = chain_slot((wv == 0) ? (down ? 23 - kk : kk) : (down ? 32 - kk : kk - 9), 8);
    }
    __syncthreads();
  }
  // ---- the global root, step 16: Delta_16 = Ttilde_16 + R_left + R_right, inverted by row 0 of wavefront 0
  if (rowB && wlane) {
#pragma unroll
    for (int c = 0; c < 12; c++) L.sX[wv * 144 + i * 12 + c] = R[c];
  }
  __syncthreads();
  {
    const int kk = 16, kn = 17;
    const double omki = L.sOm[kk * 12 + i], omni = L.sOm[kn * 12 + i], omn6 = L.sOm[kn * 12 + i6];
    const double diag = L.sDg[kk * 12 + i] + lo6 * (omki + omni) + hi6 * (dt * dt * omn6);
    double m[12];
#pragma unroll
    for (int c = 0; c < 12; c++) {
      double v = (c == i) ? diag : 0.0;
      if (c >= 6) v = (c - 6 == i) ? dt * omni : v;
      if (c < 6) v = (c + 6 == i) ? dt * omn6 : v;
      if (c >= 6) v += hi6 * (L.sW[kk * kDisWSz + i6 * 6 + (c - 6)] + L.sW[kn * kDisWSz + i6 * 6 + (c - 6)]);
      m[c] = v + L.sX[i * 12 + c] + L.sX[144 + i * 12 + c];
    }
    gj_invert12(m, i);
    if (wv == 0 && !rowB && wlane) {
#pragma unroll
      for (int c = 0; c < 12; c++) L.sRootInv[c * kCol + i] = m[c];
    }
  }
  __syncthreads();
  // the hand-off buffers overlaid the vectors: restore the positions the sweeps rely on being zero
  if (tid < 12) {
    L.sX[16 * 12 + tid] = 0.0;
    L.sX[kDisRightPos * 12 + tid] = 0.0;
    L.sX[(kDisRightPos + 16) * 12 + tid] = 0.0;
  }
  __syncthreads();
}

// Solve of the dissected system for one right-hand side, all 128 threads of the instance.  rX: this lane's three entries
// of the right-hand side (step k, entries 3j..3j+2); DiRows(t, dr): the lane's row t of Delta_k^-1.  The solution is left in
// L.sX at dis_pos(step).  Two workgroup barriers inside (fill contributions -> root; end of the backward sweeps).
// Mark(i): phase marker hook of diagnostic builds (3: after the forward sweeps, 4: after phases 1-3, 5: at the end).
template <typename LdsT, typename RowFn, typename MarkFn>
__device__ __forceinline__ void dis_solve(LdsT& L, const double (&rX)[3], RowFn&& DiRows, int lane, int wv, int k, int j, MarkFn&& Mark) {
  const bool isfill = dis_is_fill(k);
  const bool isroot = (k == 16);
  const int kx = dis_pos(k);
  double* dump = L.sW;
  {
    double* px = isroot ? &L.sX[kDisRootRhs * 12 + 3 * j] : &L.sX[kx * 12 + 3 * j];
#pragma unroll
    for (int t = 0; t < 3; t++) px[t] = rX[t];
  }
  asm volatile("" ::: "memory");  // a half's right-hand side is written by the wavefront that sweeps it
  chain_forward_paired<16>(L.sN + wv * kDisRightBase, L.sX + wv * kDisRightPos * 12, dump, lane);
  asm volatile("" ::: "memory");
  Mark(3);
  // ---- phase 1: v = Delta^-1 u (kept in registers), fill contributions d = E' u into sC
  const double* pE = L.sFill + (isfill ? dis_fill_index(k) : 0) * kFillStride;
  double v[3];
  {
    double u[12];
#pragma unroll
    for (int c = 0; c < 12; c++) u[c] = L.sX[(isroot ? kDisRootRhs : kx) * 12 + c];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      double s_ = 0.0, dr[12];
      DiRows(t, dr);
#pragma unroll
      for (int c = 0; c < 12; c++) s_ += dr[c] * u[c];
      v[t] = s_;
    }
    double d[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 12; r++) {
#pragma unroll
      for (int t = 0; t < 3; t++) d[t] += pE[r * 12 + 3 * j + t] * u[r];
    }
    double* pC = isfill ? &L.sC[(3 * j) * 16 + dis_fill_index(k)] : dump + 24 + 3 * j;
    const int cs = isfill ? 16 : 1;
#pragma unroll
    for (int t = 0; t < 3; t++) pC[t * cs] = d[t];
  }
  __syncthreads();
  // ---- phase 2: step 16, by DPP row 0 of EACH wavefront (the same arithmetic twice: no second barrier to publish x_16)
  {
    const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
    const v4d* pc = reinterpret_cast<const v4d*>(&L.sC[i * 16]);
    const v4d c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3];
    double mr[12];
#pragma unroll
    for (int c = 0; c < 12; c++) mr[c] = __hip_atomic_load(&L.sRootInv[c * kCol + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const double r16 = L.sX[kDisRootRhs * 12 + i];
    const double sl = ((c0.x + c0.y) + (c0.z + c0.w)) + ((c1.x + c1.y) + (c1.z + c1.w));
    const double sr = ((c2.x + c2.y) + (c2.z + c2.w)) + ((c3.x + c3.y) + (c3.z + c3.w));
    const double u16 = r16 + (sl + sr);
    const double x16 = dpp_step12(0.0, u16, mr);
    double* px = (lane < 12) ? &L.sX[kDisRootX * 12 + i] : dump + 48 + i;
    *px = x16;
  }
  asm volatile("" ::: "memory");
  // ---- phase 3: v += E x_16 on the fill steps; v into the vector positions
  {
    double x16[12];
#pragma unroll
    for (int c = 0; c < 12; c++) x16[c] = L.sX[kDisRootX * 12 + c];
    double a[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
      for (int c = 0; c < 12; c++) a[t] += pE[(3 * j + t) * 12 + c] * x16[c];
    }
    double* px = isroot ? dump + 64 + 3 * j : &L.sX[kx * 12 + 3 * j];
#pragma unroll
    for (int t = 0; t < 3; t++) px[t] = isfill ? v[t] + a[t] : v[t];
  }
  asm volatile("" ::: "memory");
  Mark(4);
  chain_backward_paired<16>(L.sN + wv * kDisRightBase, L.sX + wv * kDisRightPos * 12, dump, lane);
  __syncthreads();
  Mark(5);
}

}  // namespace qrw
