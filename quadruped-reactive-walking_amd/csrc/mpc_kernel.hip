// MPC QP build + OSQP-style ADMM solve for a batch of Solo12 instances — gfx950 (MI355X).
//
// Replaces, per instance, MPC::run of the reference (/root/reference/src/MPC.cpp:626-649):
//   construct_gait :686-701, construct_S :665-681, create_ML/update_ML :74-256/:418-464 (only
//   the 36 B coefficients + 12 S flags per step vary; the sparsity is compile-time here),
//   create_NK/update_NK :261-312/:469-496, create_weight_matrices :317-391, call_solver
//   :501-564 (OSQP 0.6.x: scale_data, osqp_update_A/bounds, osqp_solve — restated from the
//   published algorithm and the 0.6.x source structure) and retrieve_result :569-599.
//
// Mapping: ONE WAVEFRONT PER ROBOT INSTANCE, lane = 4*k + j with k = horizon step, j = foot.
// Lane (k,j) owns state entries X_k[3j..3j+2], force entries f_k[3j..3j+2], dynamics rows
// (k,3j..3j+2), force-enable rows (k,3j..3j+2) and the 5 friction-cone rows of foot j at step k:
// 6 of the 24N variables and 11 of the 44N constraints per lane — all ADMM vectors live in
// registers.  The KKT solve (P + sigma I + A' R A) x = r is done by block elimination:
//   1. forces f_k are eliminated per step inside each quad through the matrix inversion lemma (a 3x3 inverse per foot
//      and a 6x6 inverse per step, force_block_factor),
//   2. the remaining block-tridiagonal system in the states (12x12 blocks, N steps) is solved by a twisted
//      (two-ended) block LDL' recursion; factorisation and the two sequential sweeps run on the FP64 VALU with DPP
//      row broadcasts (v_fmac_f64_dpp row_newbcast: lane i of a 16-lane row holds vector entry i and matrix row i;
//      chain_sweep.h), both chains in one instruction stream,
//   3. forces are back-substituted inside each quad.
// The loop carries OSQP's scaled iterates in unscaled ("hatted") form, see the comment at the loop variables.
// LDS holds the N-1 chain matrices (column-major 12x12) and one 12N exchange vector.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>
#include <vector>

#include "chain_sweep.h"
#include "qrw_device.h"
#include "qrw_kernels.h"

namespace qrw {

namespace {

constexpr int kMatSz = 144;
constexpr int kWSz = 36;

// LDS carve (doubles) for NW wavefronts per instance (16 horizon steps per wavefront): chain matrices -N_k
// (k = 1..N-1), exchange vector, neighbour-exchange buffer (NW > 1 only), factor scratch.
template <int NW>
struct alignas(16) MpcLdsT {
  static constexpr int S = 16 * NW;
  double sN[chain_lds_doubles(S, S / 2)];  // S-1 chain matrices (+1 slot only ever read by the idle chain's discarded step); chain_slot()
  double sX[(S + 2) * 12];   // step k at chain_pos(k); position N is a zero vector, N+1 padding for the idle step
  double sDump[(S / 2 + 2) * 12];  // sink for the sweeps' masked stores
  double sE[S * 12];     // step k <-> k+-1 exchange across wavefronts
  double sW[S * kWSz];   // K_k^-1 = Omega_k - Gbar F^-1 Gbar' (6x6) per step (factor phase)
  double sOm[S * 12];    // omega_D per step (factor phase)
  double sDg[S * 12];    // c*w + sigma/Dx^2 per step (factor phase)
  double sA[kMatSz];     // Delta_{k-1}^-1
  double sB[kMatSz];     // Delta_k (inverted in place)
  double sRed[4];        // cross-wavefront reductions
  unsigned long long sBal[2];
  double sPre[8];        // time-sliced launch: (primal residual, its norm, dual residual, its norm) of the last two adaptive-rho tests
};

// Workgroup barrier of the ADMM loop.  One wavefront per instance: the LDS executes a wavefront's operations in order, so
// no barrier and no drain of the LDS queue is needed between a phase's stores and the next phase's loads -- only the
// compiler must keep their order.  (__syncthreads() would cost an s_waitcnt lgkmcnt(0) at every phase boundary.)
template <int NW>
__device__ __forceinline__ void wg_sync_t() {
  if constexpr (NW == 1) asm volatile("" ::: "memory");
  else __syncthreads();
}
#define wg_sync() wg_sync_t<NW>()
#ifdef QRW_PROFILE_PHASES
#define PH_DECL unsigned long long ph_t0 = __builtin_amdgcn_s_memtime(), ph_acc[10] = {0,0,0,0,0,0,0,0,0,0};
#ifndef QRW_PH_MASK
#define QRW_PH_MASK 0x3FF
#endif
#define PH(i) do { if ((QRW_PH_MASK >> (i)) & 1) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_t0; ph_t0 = t_; } } while (0)
#elif defined(QRW_MARK_PHASES)  // markers in the ISA listing (scripts/phase_mix.py counts instructions per phase)
#define PH_DECL
#define PH(i) asm volatile("; QRW_PHASE " #i)
#else
#define PH_DECL
#define PH(i)
#endif

// ---- friction-cone block helpers (rows: fx-mu fz, -fx-mu fz, fy-mu fz, -fy-mu fz, -fz; MPC.cpp:130-146)
__device__ __forceinline__ void cone_apply(const double f[3], double mu, double out[5]) {
  out[0] = f[0] - mu * f[2];
  out[1] = -f[0] - mu * f[2];
  out[2] = f[1] - mu * f[2];
  out[3] = -f[1] - mu * f[2];
  out[4] = -f[2];
}
__device__ __forceinline__ void cone_apply_t(const double w[5], double mu, double out[3]) {
  out[0] = w[0] - w[1];
  out[1] = w[2] - w[3];
  out[2] = -mu * (w[0] + w[1] + w[2] + w[3]) - w[4];
}

// ---- step k <-> k+1 / k-1 neighbour values. One wavefront: ds_bpermute; two wavefronts: through LDS.
// same[t] = value of lane (k+1, j); shifted[t] = value of lane (k+1, j-2) (only meaningful for j >= 2).
// (Measured and dropped, round 3: leaving out the second barrier where another workgroup barrier separates this exchange from the
// next one through the buffer anyway -- the force-elimination exchange of the ADMM loop -- is 1.9 % SLOWER at N = 32, A/B on one
// box: 91.4 k against 93.2 k control steps/s; a second buffer for the other exchange costs the kernel 12-36 B of scratch.)
// CL (one wavefront, runtime horizon N < 16): a step without a successor reads ITSELF, as the LDS form below does.  The caller
// multiplies what comes back by a 0 coefficient then, but lane + 4 would be a quad of an idle step, whose values derive from LDS
// nobody has written: 0 x NaN.  (With the compile-time horizon lane + 4 of step 15 wraps to step 0: finite.)  Found by filling the
// LDS with NaN before the known-answer solve (qrw_test_known_answer; tests/test_gpu_api_errors.py).
template <int NW, bool CL = false>
__device__ __forceinline__ void nb_next(const double v[3], double same[3], double shifted[3], double* sE, int k, int j,
                                        int lane, bool has_next) {
  if constexpr (NW == 1) {
    const int s4 = (CL && !has_next) ? lane : lane + 4, s2 = (CL && !has_next) ? lane : lane + 2;
#pragma unroll
    for (int t = 0; t < 3; t++) { same[t] = shfl(v[t], s4); shifted[t] = shfl(v[t], s2); }
  } else {
#pragma unroll
    for (int t = 0; t < 3; t++) sE[k * 12 + 3 * j + t] = v[t];
    __syncthreads();
    const int kn = has_next ? k + 1 : k;
#pragma unroll
    for (int t = 0; t < 3; t++) { same[t] = sE[kn * 12 + 3 * j + t]; shifted[t] = sE[kn * 12 + 3 * ((j + 2) & 3) + t]; }
    __syncthreads();  // (the schedule around it stays what it was with the barrier: the register file is full)
  }
}
// same[t] = value of lane (k-1, j); shifted[t] = value of lane (k-1, j+2) (only meaningful for j < 2).
template <int NW, bool CL = false>
__device__ __forceinline__ void nb_prev(const double v[3], double same[3], double shifted[3], double* sE, int k, int j,
                                        int lane, bool has_prev) {
  if constexpr (NW == 1) {
    const int s4 = (CL && !has_prev) ? lane : lane - 4, s2 = (CL && !has_prev) ? lane : lane - 2;
#pragma unroll
    for (int t = 0; t < 3; t++) { same[t] = shfl(v[t], s4); shifted[t] = shfl(v[t], s2); }
  } else {
#pragma unroll
    for (int t = 0; t < 3; t++) sE[k * 12 + 3 * j + t] = v[t];
    __syncthreads();
    const int kp = has_prev ? k - 1 : k;
#pragma unroll
    for (int t = 0; t < 3; t++) { same[t] = sE[kp * 12 + 3 * j + t]; shifted[t] = sE[kp * 12 + 3 * ((j + 2) & 3) + t]; }
    __syncthreads();
  }
}
template <int NW>
__device__ __forceinline__ double block_max(double v, double* sRed, int wv, int lane) {
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  else {
    if (lane == 0) sRed[wv] = v;
    __syncthreads();
    const double r = fmax(sRed[0], sRed[1]);
    __syncthreads();
    return r;
  }
}
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* sRed, int wv, int lane) {
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  else {
    if (lane == 0) sRed[wv] = v;
    __syncthreads();
    const double r = sRed[0] + sRed[1];
    __syncthreads();
    return r;
  }
}

}  // namespace

// y = K^-1 s for the quad's 6x6 K^-1: lane j computes rows j and (j < 2 ? j + 4 : a duplicate) from its two rows Kr,
// the six results are then broadcast inside the quad.  s must be the same in all four lanes.
__device__ __forceinline__ void kinv_apply(const double (&Kr)[2][6], const double (&s)[6], double (&y)[6]) {
  double ya = Kr[0][0] * s[0], yb = Kr[1][0] * s[0];
#pragma unroll
  for (int c = 1; c < 6; c++) { ya += Kr[0][c] * s[c]; yb += Kr[1][c] * s[c]; }
  y[0] = quad_bcast<0>(ya); y[1] = quad_bcast<1>(ya); y[2] = quad_bcast<2>(ya); y[3] = quad_bcast<3>(ya);
  y[4] = quad_bcast<0>(yb); y[5] = quad_bcast<1>(yb);
}

// Force block of the KKT system of one horizon step, per quad.  F_k = D + B' Omega B with D block-diagonal (one 3x3
// per foot: cone rows, force-enable rows, cost and sigma) and B = [dt/m I; dt I^-1 skew(lever)] (6x12, the velocity rows
// of the dynamics), so by the matrix inversion lemma everything the elimination needs comes from D_j^-1 (3x3, per lane)
// and K^-1 = (Omega^-1 + B D^-1 B')^-1 (6x6, per step):
//   Gbar F^-1 r        = -K^-1 B D^-1 r                       (Gbar = -Omega B; the term taken out of the state rhs)
//   F^-1 (r - Gbar' d) = D^-1 r - D^-1 B' K^-1 (B D^-1 r - d) (force back-substitution)
//   Omega - Gbar F^-1 Gbar' = K^-1                            (what the velocity block of the state system sees)
// omL / omA: omega_D of the linear / angular velocity rows of the step (the values lanes 2 / 3 of the quad own),
// omS / omC: of this lane's force-enable / cone rows.  Dinv: D_j^-1 as (00, 01, 02, 11, 12, 22); Kinv: K^-1 (every
// lane of the quad holds the same copy).
__device__ __forceinline__ void force_block_factor(bool act, const double (&Bang)[3][3], const double (&sfl)[3],
                                                   const double (&iDf)[3], const double (&omL)[3], const double (&omA)[3],
                                                   const double (&omS)[3], const double (&omC)[5], double cs, double wF,
                                                   double sigma, double dtm, double mu, double (&Dinv)[6],
                                                   double (&Kinv)[6][6]) {
  // ---- D_j and its inverse (3x3 symmetric positive definite: adjugate / determinant)
  const double s4 = omC[0] + omC[1] + omC[2] + omC[3];
  const double d00 = omC[0] + omC[1] + (cs * wF + sigma * iDf[0] * iDf[0] + sfl[0] * sfl[0] * omS[0]);
  const double d11 = omC[2] + omC[3] + (cs * wF + sigma * iDf[1] * iDf[1] + sfl[1] * sfl[1] * omS[1]);
  const double d22 = mu * mu * s4 + omC[4] + (cs * wF + sigma * iDf[2] * iDf[2] + sfl[2] * sfl[2] * omS[2]);
  const double d02 = -mu * (omC[0] - omC[1]), d12 = -mu * (omC[2] - omC[3]);  // d01 = 0
  {
    const double c00 = d11 * d22 - d12 * d12, c01 = d02 * d12, c02 = -d02 * d11;
    const double c11 = d00 * d22 - d02 * d02, c12 = -d00 * d12, c22 = d00 * d11;
    const double idet = fast_rcp(d00 * c00 + d02 * c02);
    Dinv[0] = act ? c00 * idet : 1.0; Dinv[1] = act ? c01 * idet : 0.0; Dinv[2] = act ? c02 * idet : 0.0;
    Dinv[3] = act ? c11 * idet : 1.0; Dinv[4] = act ? c12 * idet : 0.0; Dinv[5] = act ? c22 * idet : 1.0;
  }
  const double Ds[3][3] = {{Dinv[0], Dinv[1], Dinv[2]}, {Dinv[1], Dinv[3], Dinv[4]}, {Dinv[2], Dinv[4], Dinv[5]}};
  // ---- K = Omega^-1 + sum over the feet of B_j D_j^-1 B_j'   (B_j = [dtm I; Bang_j])
  double E[3][3];  // Bang D^-1
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) E[r][c] = Bang[r][0] * Ds[0][c] + Bang[r][1] * Ds[1][c] + Bang[r][2] * Ds[2][c];
  double K[6][6];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = r; c < 3; c++) {
      K[r][c] = quad_sum(act ? dtm * dtm * Ds[r][c] : 0.0);
      K[3 + r][3 + c] = quad_sum(act ? E[r][0] * Bang[c][0] + E[r][1] * Bang[c][1] + E[r][2] * Bang[c][2] : 0.0);
    }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) K[c][3 + r] = quad_sum(act ? dtm * E[r][c] : 0.0);
#pragma unroll
  for (int c = 0; c < 3; c++) { K[c][c] += fast_rcp(omL[c]); K[3 + c][3 + c] += fast_rcp(omA[c]); }
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < r; c++) K[r][c] = K[c][r];
  // ---- K^-1 by Gauss-Jordan (symmetric positive definite: no pivoting), the same in every lane of the quad
#pragma unroll
  for (int p = 0; p < 6; p++) {
    const double d = fast_rcp(K[p][p]);
    double prow[6];
#pragma unroll
    for (int c = 0; c < 6; c++) prow[c] = K[p][c] * d;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r == p) continue;
      const double f = K[r][p];
#pragma unroll
      for (int c = 0; c < 6; c++) K[r][c] = (c == p) ? -f * d : K[r][c] - f * prow[c];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) K[p][c] = (c == p) ? d : prow[c];
  }
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < 6; c++) Kinv[r][c] = K[r][c];
}

// Twisted block LDL' over the states (see chain_sweep.h), in registers: DPP row 0 of wavefront 0 factorises chain A
// (steps 0..m-1 upwards), row 1 chain B (steps N-1..m+1 downwards) AT THE SAME TIME, lane i of a row holding row i of
// the step's 12x12 block; both then assemble and invert the root step m.  Per step: build Delta = Ttilde - (coupling
// to the step eliminated just before), invert it by in-register Gauss-Jordan, hand its rows to the quad that owns the
// step (Di), and form the next coupling matrix (negated, into sN).  Replaces a version that kept the block in LDS and
// needed two workgroup barriers per pivot (about 7x the time).
template <int NT, typename LdsT>
__device__ __forceinline__ void chain_factorize(LdsT& L, int N, double dt, int tid, int k, int j, double (*Di)[12]) {
  const int mroot = N >> 1, nA = mroot, nB = N - 1 - mroot;
  const int steps = (nA > nB) ? nA : nB;
  const int lane = tid & 63;
  const bool worker = tid < 64;
  const bool rowB = (lane & 16) != 0;
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11;
  const int i6 = (i >= 6) ? i - 6 : i;  // index into the 6x6 velocity block (clamped for the position rows)
  const double sA = rowB ? 0.0 : 1.0, sB = rowB ? 1.0 : 0.0;
  const double hi6 = (i >= 6) ? 1.0 : 0.0, lo6 = (i < 6) ? 1.0 : 0.0;
  double nn[12];  // row i of the negated coupling matrix made by this chain's previous step
#pragma unroll
  for (int c = 0; c < 12; c++) nn[c] = 0.0;
  for (int s = 0; s <= steps; s++) {
    const bool root = (s == steps);
    const bool active = root ? true : (rowB ? (s < nB) : (s < nA));
    const bool hasprev = root ? (rowB ? (nB > 0) : (nA > 0)) : (s > 0);
    int kk = root ? mroot : (rowB ? N - 1 - s : s);
    if (!active) kk = mroot;
    const bool last = (kk + 1 == N);
    const int kn = last ? kk : kk + 1;
    const double nl = last ? 0.0 : 1.0;
    double* Mout = rowB ? L.sB : L.sA;
    double m[12];
    if (worker) {
      // ---- Ttilde_kk, row i
      const double omki = L.sOm[kk * 12 + i], omni = nl * L.sOm[kn * 12 + i], omn6 = nl * L.sOm[kn * 12 + i6];
      // velocity rows: the dynamics rows' own omega is inside K^-1 (force_block_factor)
      const double diag = L.sDg[kk * 12 + i] + lo6 * (omki + omni) + hi6 * (dt * dt * omn6);
#pragma unroll
      for (int c = 0; c < 12; c++) {
        double v = (c == i) ? diag : 0.0;
        if (c >= 6) v = (c - 6 == i) ? dt * omni : v;          // (i, i+6), i < 6
        if (c < 6) v = (c + 6 == i) ? dt * omn6 : v;           // (i, i-6), i >= 6
        if (c >= 6) v += hi6 * (L.sW[kk * kWSz + i6 * 6 + (c - 6)] + nl * L.sW[kn * kWSz + i6 * 6 + (c - 6)]);
        m[c] = v;
      }
      // ---- Schur term of the step this chain eliminated just before: chain A  -N_kk C_kk', chain B  -Nt_kk C_{kk+1}
      {
        const int ks = rowB ? kn : kk;
        const double* omS = &L.sOm[ks * 12];
        const double* WS = &L.sW[ks * kWSz];
        double term[12];
#pragma unroll
        for (int c = 0; c < 6; c++) term[c] = -omS[c] * nn[c] - sA * (dt * omS[c]) * nn[c + 6];
#pragma unroll
        for (int c = 6; c < 12; c++) {
          double v = -sB * (dt * omS[c - 6]) * nn[c - 6];
#pragma unroll
          for (int mm = 0; mm < 6; mm++) v -= WS[(c - 6) * 6 + mm] * nn[6 + mm];
          term[c] = v;
        }
        if (root) {  // the root couples to both chains: add the other chain's term
#pragma unroll
          for (int c = 0; c < 12; c++) {
            const double mine = hasprev ? term[c] : 0.0;
            const double other = shfl(mine, lane ^ 16);
            m[c] += mine + other;
          }
        } else {
#pragma unroll
          for (int c = 0; c < 12; c++) m[c] += hasprev ? term[c] : 0.0;
        }
      }
      gj_invert12(m, i);
      if (active && lane < 32 && (lane & 15) < 12 && !(root && rowB)) {
#pragma unroll
        for (int c = 0; c < 12; c++) Mout[i * 12 + c] = m[c];
      }
    }
    wg_sync_t<NT / 64>();
    if (Di != nullptr) {
      const int kA = root ? mroot : ((s < nA) ? s : -1), kB = root ? -1 : ((s < nB) ? N - 1 - s : -1);
      if (k == kA || k == kB) {
        const double* Mi = (k == kB) ? L.sB : L.sA;
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
          for (int c = 0; c < 12; c++) Di[t][c] = Mi[(3 * j + t) * 12 + c];
      }
    }
    if (worker && !root) {
      // ---- next coupling matrix, negated: chain A  -N_{kk+1} = -C_{kk+1} Delta_kk^-1 (slot kk),
      //                                      chain B  -Nt_{kk-1} = -C_kk' Delta_kk^-1   (slot m + N-2-(kk-1))
      const double omki = L.sOm[kk * 12 + i], omni = L.sOm[kn * 12 + i];
      const double omk6 = L.sOm[kk * 12 + i6];
      const double* Wc = rowB ? &L.sW[kk * kWSz] : &L.sW[kn * kWSz];
      double nx[12], coef[6];
      const double own = lo6 * (rowB ? omki : omni);
#pragma unroll
      for (int c = 0; c < 12; c++) nx[c] = own * m[c];
      // rows 6..11 of Delta^-1: the 6x6 velocity-block coupling K^-1 (both chains; symmetric) and chain A's dt term
#pragma unroll
      for (int mm = 0; mm < 6; mm++) {
        double cf = hi6 * Wc[i6 * 6 + mm];
        if (mm == i) cf = sA * (dt * omni);  // i < 6 only: row i+6
        coef[mm] = cf;
      }
      row_bcast_fma<6>(nx, m, coef[0]); row_bcast_fma<7>(nx, m, coef[1]); row_bcast_fma<8>(nx, m, coef[2]);
      row_bcast_fma<9>(nx, m, coef[3]); row_bcast_fma<10>(nx, m, coef[4]); row_bcast_fma<11>(nx, m, coef[5]);
      // chain B's dt term: row i-6 (i >= 6)
      {
        const double cb = sB * hi6 * (dt * omk6);
#pragma unroll
        for (int c = 0; c < 12; c++) nx[c] += cb * row_shr6(m[c]);
      }
      if (active && lane < 32 && (lane & 15) < 12) {
        const int slot = rowB ? (mroot + N - 2 - (kk - 1)) : kk;
#pragma unroll
        for (int c = 0; c < 12; c++) L.sN[chain_slot(slot, mroot) + c * kCol + i] = nx[c];
      }
#pragma unroll
      for (int c = 0; c < 12; c++) nn[c] = active ? nx[c] : nn[c];
    }
    wg_sync_t<NT / 64>();
  }
}

// ---- task queues of the sequence kernel (qrw_mpc_solve_sequence).  A task = call * B + instance; K * B tasks in all.
// Why more than one queue: with a single FIFO a ready task waits behind ~3 B / 4 others (3.8 ms at batch 4096 on 1024
// resident workgroups), and an instance that needs ~2000 iterations call after call (there are such: 38 650 iterations in
// 20 calls against a mean of 10 300) pays that wait on top of every one of its long solves: its chain, not the work, ends
// the sequence (measured: 6.7 ms per call, slower than one launch per call).  The instances are therefore ranked by the
// longest-first order of the handle (moving average of their iteration counts) into kSeqLevels levels (the first 1/16, up to 1/4, up to 1/2 of the order, the rest), each with its own
// FIFO, and a workgroup looking for work takes from the highest level that has a task: the long chains never wait, the
// short tasks absorb the waiting.
//   Per level l: tasks in queue[base_l + slot], -1 while unfilled; head (next slot to take) and tail (next slot to fill) in
//   words kSeqStride * l and + 1 of qctr (one 64-byte line per level).  Slots are RESERVED with fetch_add (no
//   compare-and-swap loops, no shared line polled by everybody): a workgroup reserves in a level only when it sees
//   head < tail there — or, in the lowest level, whenever that level still has tasks to come — and then waits for its own
//   slot.  Seeing head < tail and losing the race leaves it with a slot that the level's next finished task fills.
//   A level of n instances sees exactly n * K tasks: a reserved slot beyond that means the level is exhausted.
//   The first tasks (as many as there are workgroups) are dealt by workgroup index, not through the queues: a thousand
//   workgroups starting together would otherwise all reserve in level 0.
// Hand-off of an instance's solver state between workgroups on different CUs / XCDs: release at agent scope by the
// finishing workgroup after its stores have left (s_waitcnt vmcnt(0)), relaxed queue store; relaxed polls, then ONE acquire
// at agent scope by the taking workgroup (MI355X guide, inter-workgroup visibility: L1 is never refreshed by other CUs'
// stores, the XCDs' L2s are not coherent with each other).  Measured: dropping the fences changes results (stale state).
constexpr int kSeqLevels = 4;
constexpr int kSeqStride = 16;                     // unsigned words per level in qctr
static_assert(kSeqLevels * kSeqStride + 16 + kSeqLevels <= kSeqQctrWords, "qctr too small");
constexpr int kSeqErr = kSeqLevels * kSeqStride;
static_assert(kSeqErr == kSeqErrWord, "qrw_kernels.h and mpc_kernel.hip disagree on the queue counters' layout");   // error flag; + 1 diagnostics; + 2 + l: tasks through level l's queue; + 8: finished tasks (progress); + 16 + l: its base
constexpr int kSeqProgress = kSeqErr + 8;
__device__ __forceinline__ unsigned q_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Give-up clock of a workgroup that waits for work.  It measures the time since the last OBSERVED PROGRESS of the whole
// sequence (the finished-task counter), not since the workgroup started to look: at the tail of a long sequence the only
// tasks left are the later calls of a few long chains (K calls x up to 4000 iterations x 2.3 us can be many seconds of serial
// work), every other workgroup is resident and polling, and a clock started at workgroup start would make them exit without a
// task -- calls that then never run.  A task ends at least every ~15 ms (4000 iterations at N = 32) while anything runs, so
// 2 s without a single finished task means the queue is really stuck.  The counter is read only when the 2 s have passed.
struct SeqGiveUp {
  unsigned long long t0;
  unsigned last;
  const unsigned* prog;
  __device__ __forceinline__ SeqGiveUp(const unsigned* p) : t0(__builtin_amdgcn_s_memrealtime()), last(q_load(p)), prog(p) {}
  __device__ __forceinline__ bool expired() {
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    if (now - t0 <= 200000000ull) return false;
    const unsigned p = q_load(prog);
    if (p != last) { last = p; t0 = now; return false; }
    return true;
  }
};
template <int NW>
__device__ __forceinline__ int seq_next_task(const MpcArgs& a, unsigned long long* sh, int tid, bool first_pass) {
  int task = -1;
  if (tid == 0) {
    const unsigned long long t_look = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    SeqGiveUp clock(&a.qctr[kSeqProgress]);
    if (first_pass) task = a.seq_first[blockIdx.x];  // dealt by workgroup index (the init kernel left these out of the queues)
    bool give_up = first_pass;
    while (task < 0 && !give_up) {
      bool all_done = true;
      for (int l = 0; l < kSeqLevels && task < 0; l++) {
        const unsigned n_tot = a.qctr[kSeqErr + 2 + l];  // tasks that pass through this level's queue (init kernel)
        if (n_tot == 0) continue;
        unsigned* head = &a.qctr[l * kSeqStride];
        const unsigned h = q_load(head);
        if (h >= n_tot) continue;  // exhausted
        all_done = false;
        if (h < q_load(head + 1) || l == kSeqLevels - 1) {
          const unsigned slot = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (slot >= n_tot) continue;
          const int* q = a.queue + a.qctr[kSeqErr + 16 + l] + slot;
          for (;;) {
            task = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (task >= 0) break;
            __builtin_amdgcn_s_sleep(32);
            if (clock.expired()) {  // 2 s without any task of the sequence finishing: the task never came; give up, loudly
              __hip_atomic_store(&a.qctr[kSeqErr], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              give_up = true;
              break;
            }
          }
        }
      }
      if (task >= 0 || all_done) break;
      __builtin_amdgcn_s_sleep(127);  // upper levels empty right now and the lowest one exhausted: look again in a few microseconds
      if (clock.expired()) {
        __hip_atomic_store(&a.qctr[kSeqErr], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        give_up = true;
      }
    }
    if (task >= 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef QRW_SEQ_STATS  // diagnostics: 100 MHz ticks spent looking for work, summed over the workgroups
    __hip_atomic_fetch_add(&a.qctr[kSeqErr + 1], (unsigned)(__builtin_amdgcn_s_memrealtime() - t_look), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }
  if constexpr (NW == 1) {
    task = __builtin_amdgcn_readfirstlane(task);
  } else {
    int* si = reinterpret_cast<int*>(sh);
    __syncthreads();  // the previous task's last LDS reads are done
    if (tid == 0) si[0] = task;
    __syncthreads();
    task = si[0];
    __syncthreads();
  }
  return task;
}
template <int NW>
__device__ __forceinline__ void seq_finish_task(const MpcArgs& a, int b, int seq_s, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores of the instance's state and results have left
  if constexpr (NW > 1) __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(&a.qctr[kSeqProgress], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // SeqGiveUp's clock
  if (tid == 0 && seq_s + 1 < a.seq_K) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int l = a.seq_hot[b];
    const unsigned slot = __hip_atomic_fetch_add(&a.qctr[l * kSeqStride + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.queue[a.qctr[kSeqErr + 16 + l] + slot], (seq_s + 1) * a.B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- PRE: time slicing of one call's solves inside one launch (qrw_mpc_solve at N > 16, batch > resident slots).
// A launch's workgroups are started in index order on the 512 resident slots (two-wavefront instances) and each runs to the
// end of its solve, 150..4000 ADMM iterations that nobody can predict BEFOREHAND well enough (3.5 % of the N = 32 mixed-gait
// solves hit max_iter, never the same instances; profiles/r3_lpt_sim_n32.txt): whatever the order, some long solves start in
// the last round and the launch ends 1.26-1.29 x later than work / slots.  Here a workgroup runs AT MOST `pre_chunk`
// iterations (cut at a multiple of 200, where OSQP's adaptive-rho test sits), then parks the instance -- the ADMM loop
// variables go into the state slots the warm start uses, bit for bit -- and queues it; the grid has B * pre_cmax
// workgroups, the first B take the instances directly, every later one takes a parked instance.  A resumed solve re-assembles
// and re-equilibrates (deterministic: same inputs), reloads the loop variables, re-factors for the rho it was parked with and
// goes on with iteration it0 + 1: the same arithmetic as the uninterrupted solve, so iteration counts, status and results
// are identical (the parity tests run through this path), WHATEVER the order the parked solves are taken in.
//   Which parked solve next (round 3, second half): one FIFO (round robin: all solves advance together, so when the slots
// outnumber the unfinished solves those are the genuinely long ones, near their end) ends 1.09 x after work / slots.  But once
// a solve RUNS its length can be predicted: ADMM converges linearly, the residuals fall by a constant factor per iteration
// between rho updates, so with r = max(primal residual / its tolerance, dual residual / its tolerance) at two consecutive
// adaptive-rho tests the remaining iterations are ~ 200 ln r / ln(r_prev / r) (log-error 0.23 at iteration 600, 0.12 at 1200 on
// the recorded traces, scripts/gpu_res_trace.py).  So: a solve parked after its FIRST slice goes to level 0 (a FIFO: the
// prediction at iteration 600 still misses the slow mode that takes over later in about 1 % of the solves, and one such solve
// left for the end costs more than all the ordering gains); from the second park on its level is 1 + (levels - 2 -
// predicted remaining / pre_bin), most remaining first, re-predicted at every park.  A workgroup takes from the lowest-numbered
// level that has a solve: longest-remaining-first on 200-iteration bins, 1.03 x in the simulation (scripts/pre_priority_sim.py).
//   Queues: pre_queue[level][slot]; per level a head (next slot to take) and a tail (next slot to fill).
//   NOBODY WAITS FOR WORK THAT A NOT-YET-RUNNING WORKGROUP WOULD HAVE TO PRODUCE (round 5; rounds 3-4 gated the takers with
// tickets and let them spin until a running slice parked: up to a slice's length, with a 2 s give-up clock -- outside the
// forward-progress guarantees of the programming model).  The B * (pre_cmax - 1) taker workgroups are a BUDGET, `claims` counts
// how many of them are spoken for:
//   * a slice that reaches its end asks for a taker (`claims`++ < budget): granted -> it parks its solve and queues it, and its
//     own exit frees the slot the next taker starts on; refused -> it simply goes on with the next slice itself, in place,
//     without the set-up a resumed slice repeats;
//   * a taker that finds every level empty gives itself up (`claims`++ < budget -> it leaves; the launch's last phase, when all
//     unfinished solves are running and nothing is left to take); if THAT is refused, all the budget is spoken for, so exactly
//     as many solves are (about to be) queued as takers are left, one of them for this workgroup: it polls the levels for the
//     microseconds between a parker's claim and its store (the only wait left, on a RUNNING workgroup's next few instructions;
//     the give-up clock guards it all the same).
// Takers start after every index-dealt workgroup has (workgroups start in index order), i.e. when most first slices have ended
// and the levels are full; the levels run empty only when the unfinished solves fit the resident slots -- from where on parking
// would buy nothing.  If workgroups ever started in another order, takers would leave early and later slices continue in place:
// slower, never wrong, never stuck.  A batch that is resident all at once (the known-answer gate, the tests' small batches)
// would see every taker leave before the first park: its takers are launched as a grid of their own behind the index-dealt one
// (mpc_preemptive_launch), so the park / resume path is exercised deterministically.
//   pre_ctr words: qrw_kernels.h (kPre*Word)
constexpr int kPreTaken = kPreTicketWord, kPreParks = kPreParksWord, kPreDone = kPreDoneWord, kPreErr = kPreErrWord, kPreProgress = kPreProgressWord,
              kPreClaims = kPreClaimsWord;
// error word of the time-sliced launch (3, 4: the solve reserved for a taker did not arrive; 2: a level's queue overran; 9: set by
// the host before the launch, tests only: every taker leaves at once).  mpc_pre_error_flush copies it to a host-mapped word behind
// the launch (a store to host memory from inside this kernel, cold as it is, cost the N = 32 instantiation 1 % -- A/B in round 4)
__device__ __forceinline__ void pre_set_error(const MpcArgs& a, unsigned code) {
  __hip_atomic_store(&a.pre_ctr[kPreErr], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one unit of the taker budget (tid 0): true = granted
__device__ __forceinline__ bool pre_claim(const MpcArgs& a) {
  return __hip_atomic_fetch_add(&a.pre_ctr[kPreClaims], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.pre_cap;
}
template <int NW>
__device__ __forceinline__ int pre_next_task(const MpcArgs& a, unsigned long long* sh, int tid) {
  int task = -1;
  if (tid == 0 && q_load(&a.pre_ctr[kPreErr]) == 0u) {
    SeqGiveUp clock(&a.pre_ctr[kPreProgress]);
    const int* q = nullptr;
    bool owed = false;  // the budget is spoken for and this workgroup is still here: a parked solve is (about to be) queued for it
    for (;;) {
      for (int l = 0; l < a.pre_levels && !q; l++) {
        unsigned* head = &a.pre_ctr[kPreLevelWord + 2 * l];
        unsigned h = q_load(head);
        while (h < q_load(head + 1)) {
          if (__hip_atomic_compare_exchange_strong(head, &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            q = a.pre_queue + (size_t)l * a.pre_cap + h;
            break;
          }
        }
      }
      if (q) break;
      if (!owed) {
        if (pre_claim(a)) break;  // nothing to take and nothing promised: leave
        owed = true;
      }
      __builtin_amdgcn_s_sleep(8);
      if (q_load(&a.pre_ctr[kPreErr]) != 0u) break;
      if (clock.expired()) {
        pre_set_error(a, 3u);
        break;
      }
    }
    while (q) {  // the slot's store follows its reservation closely
      task = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (task >= 0) break;
      __builtin_amdgcn_s_sleep(2);
      if (clock.expired()) {
        pre_set_error(a, 4u);
        break;
      }
    }
    if (task >= 0) {
      __hip_atomic_fetch_add(&a.pre_ctr[kPreTaken], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if constexpr (NW == 1) {
    task = __builtin_amdgcn_readfirstlane(task);
  } else {
    int* si = reinterpret_cast<int*>(sh);
    if (tid == 0) si[0] = task;
    __syncthreads();
    task = si[0];
    __syncthreads();
  }
  return task;
}
// end of a slice: may this solve be parked (a taker is granted) or does the workgroup go on with it?  Same answer in every thread.
template <int NW>
__device__ __forceinline__ bool pre_may_park(const MpcArgs& a, unsigned long long* sh, int tid) {
  int ok = 0;
  if (tid == 0) ok = pre_claim(a) ? 1 : 0;
  if constexpr (NW == 1) {
    ok = __builtin_amdgcn_readfirstlane(ok);
  } else {
    int* si = reinterpret_cast<int*>(sh);
    __syncthreads();
    if (tid == 0) si[0] = ok;
    __syncthreads();
    ok = si[0];
    __syncthreads();
  }
  return ok != 0;
}
// priority level of a solve parked at iteration `iter` (tid 0): r, r_prev = max(primal residual / tolerance, dual residual /
// tolerance) at this and at the previous adaptive-rho test of this slice (0: none)
__device__ __forceinline__ int pre_level(const MpcArgs& a, int iter, int it_resume, double r, double r_prev) {
  if (a.pre_levels < 2 || it_resume == 0) return 0;  // one FIFO / parked after the first slice
  const int nl = a.pre_levels - 1;
  float rem = (float)(4000 - iter);
  if (r_prev > r && r > 1.0) rem = fminf(rem, 200.0f * __logf((float)r) / __logf((float)(r_prev / r)));
  int bin = (int)(rem / (float)a.pre_bin);
  bin = bin < 0 ? 0 : (bin > nl - 1 ? nl - 1 : bin);
  return 1 + (nl - 1 - bin);
}
// end of a chunk: parked (queue the instance in `level`) or finished (count it)
template <int NW>
__device__ __forceinline__ void pre_end_chunk(const MpcArgs& a, int b, bool parked, int level, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores of the instance's state have left
  if constexpr (NW > 1) __syncthreads();
  if (tid == 0) {
    if (parked) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned slot = __hip_atomic_fetch_add(&a.pre_ctr[kPreLevelWord + 2 * level + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < (unsigned)a.pre_cap) __hip_atomic_store(&a.pre_queue[(size_t)level * a.pre_cap + slot], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else pre_set_error(a, 2u);
      __hip_atomic_fetch_add(&a.pre_ctr[kPreParks], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (bookkeeping: solves parked in total)
    } else {
      __hip_atomic_fetch_add(&a.pre_ctr[kPreDone], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_fetch_add(&a.pre_ctr[kPreProgress], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// FULL: N == 16 * NW, every lane owns a live horizon step (the masks on `act` fold away)
// SEQ = false: one workgroup solves one instance's one MPC call (qrw_mpc_solve).
// SEQ = true:  K * B workgroups (one task each) work through K consecutive calls of every instance (qrw_mpc_solve_sequence): tasks
//   (call s, instance b) come from a queue in global memory; a task is queued when its predecessor (s-1, b) has finished,
//   so the only ordering is per instance and one instance's long solve delays nobody else's next call.
//   One task per workgroup: the sequence launch has K * B workgroups and the hardware starts a new one whenever a
//   resident one ends, exactly as in the one-call launch.  (A first version kept persistent workgroups in a task loop; with
//   the loop around it the body was compiled with 42-140 spilled SGPRs and 364 B of scratch per lane and ran 5 % slower per
//   ADMM iteration.  Without the loop the two forms compile alike.)
// PRE = true:  B * pre_cmax workgroups time-slice ONE call's solves (see pre_next_task above).
template <int NW, bool FULL, bool SEQ, bool PRE = false>
__global__ __launch_bounds__(64 * NW, 1) void mpc_solve_kernel(MpcArgs a) {
  static_assert(!(SEQ && PRE), "one queueing mode at a time");
  __shared__ MpcLdsT<NW> L;
  constexpr int T = 64 * NW;  // threads per instance
  // longest-first scheduling: blocks are dealt to the CUs in index order, so block i takes the instance with the
  // i-th largest iteration count of the PREVIOUS solve (a good predictor: warm-started receding-horizon problems)
  PH_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
#ifdef QRW_DEBUG_POISON
  // diagnostic build (scripts/gpu_poison_bisect.py): one member of the LDS struct filled with NaN before anything else runs, the
  // member chosen by the otherwise unused pre_bin -- which unwritten LDS does a solve read?
  if constexpr (!SEQ && !PRE) {
    const double qnan = __longlong_as_double(-1ll);
    auto fill = [&](double* p, int n) { for (int e = tid; e < n; e += 64 * NW) p[e] = qnan; };
    switch (a.pre_bin) {
      case 1: fill(L.sN, (int)(sizeof(L.sN) / 8)); break;
      case 2: fill(L.sX, (int)(sizeof(L.sX) / 8)); break;
      case 3: fill(L.sDump, (int)(sizeof(L.sDump) / 8)); break;
      case 4: fill(L.sE, (int)(sizeof(L.sE) / 8)); break;
      case 5: fill(L.sW, (int)(sizeof(L.sW) / 8)); break;
      case 6: fill(L.sOm, (int)(sizeof(L.sOm) / 8)); break;
      case 7: fill(L.sDg, (int)(sizeof(L.sDg) / 8)); break;
      case 8: fill(L.sA, kMatSz); break;
      case 9: fill(L.sB, kMatSz); break;
      case 10: fill(L.sRed, 4); fill(L.sPre, 8); break;
      default: break;
    }
    __syncthreads();
  }
#endif
  {
  int b, seq_s = 0;
#ifdef QRW_SEQ_STATS
  unsigned long long task_t0 = 0;
#endif
  if constexpr (SEQ) {
    const int task = seq_next_task<NW>(a, &L.sBal[0], tid, (int)blockIdx.x < a.seq_groups);
    if (task < 0) return;  // the queue timed out (the error flag in a.qctr is set)
#ifdef QRW_SEQ_STATS
    task_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    seq_s = task / a.B;
    b = task - seq_s * a.B;
  } else if constexpr (PRE) {
    if ((int)blockIdx.x + a.pre_block0 < a.B) {
      b = a.order ? a.order[blockIdx.x] : blockIdx.x;  // (index-dealt workgroups are always launched with pre_block0 == 0)
    } else {
      b = pre_next_task<NW>(a, &L.sBal[0], tid);
      if (b < 0) return;  // nothing to take (or the error word is set)
    }
  } else {
    b = a.order ? a.order[blockIdx.x] : blockIdx.x;
  }
  // PRE: iteration this instance was parked at by an earlier chunk of THIS launch (0: a fresh solve; a workgroup that takes
  // its instance by index starts the solve whatever an aborted launch may have left behind)
  int it_resume = 0;
  bool aborted = false;  // PRE: an earlier LAUNCH whose queue gave up left this instance parked: its state slots hold ADMM loop
                         // variables, not OSQP's iterates -- this solve starts cold (zero x, z, y, rho 0.1: what OSQP's
                         // store_solution leaves after a failed solve), its B coefficients / S flags are kept
  if constexpr (PRE) {
    // (agent-scope atomic loads: what another workgroup of THIS launch stored must never be served from a scalar or stale cache)
    const int pit = __hip_atomic_load(&a.pause_it[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)blockIdx.x + a.pre_block0 >= a.B) it_resume = pit;
    else aborted = (pit != 0);
  }
  const bool resumed = PRE && it_resume > 0;
  if constexpr (PRE) { if (tid < 8) L.sPre[tid] = 0.0; }
#ifdef QRW_TRACE_RES
  if (!resumed && a.prof) for (int e = tid; e < kMpcProfItems; e += T) a.prof[(size_t)b * kMpcProfItems + e] = 0.0;
#endif
  const int k = 16 * wv + (lane >> 2), j = lane & 3;
  const int N = a.N;
  const bool act = FULL ? true : (k < N);
  const bool has_next = act && (k + 1 < N);
  const bool has_prev = act && (k > 0);
  // lane-constant coefficients that replace per-iteration selects (0/1 masks folded into the multipliers)
  const bool next_here = has_next;
  const double mN = next_here ? 1.0 : 0.0, mN6 = (next_here && j >= 2) ? a.dt : 0.0;
  const double mP = has_prev ? 1.0 : 0.0, mP6 = (has_prev && j < 2) ? a.dt : 0.0;
  const double mG = (j >= 2) ? 1.0 : 0.0, mGN = (j >= 2 && next_here) ? 1.0 : 0.0;
  const int kp = has_prev ? k - 1 : k;
  // positions in sX
  const int kx = act ? chain_pos(k, N >> 1, N) : k;
  const int kpx = act ? chain_pos(kp, N >> 1, N) : k;
  if (tid < 24) L.sX[N * 12 + tid] = 0.0;
  for (int e = tid; e < kSlot; e += T) L.sN[chain_slot(16 * NW - 1, N >> 1) + e] = 0.0;

  // ---- constants (float literals promoted exactly as the reference does, MPC.cpp:17-29,330,346)
  const double dt = a.dt;
  const double mass = (double)2.50000279f;
  const double mu = (double)0.9f;
  const double dtm = dt / mass;
  const double g8 = (double)(9.81f) * dt;
  const double wF = (double)5e-5f;
  const double sigma = 1e-6, alpha = 1.6;
  const double eps_abs = 1e-6, eps_rel = 1e-6;
  const double w_all[12] = {2.0, 2.0, 20.0, 0.25, 0.25, 10.0, (double)0.2f, (double)0.2f,
                            (double)0.2f, 0.0, 0.0, (double)0.3f};
  double wX[3];
#pragma unroll
  for (int t = 0; t < 3; t++) wX[t] = (j == 0) ? w_all[t] : (j == 1) ? w_all[3 + t] : (j == 2) ? w_all[6 + t] : w_all[9 + t];

  const size_t io = (size_t)seq_s * a.B + b;  // call-major input / output blocks ([K][B][...]; K = 1 outside sequences)
  const double* xr = a.xref + io * 12 * (N + 1);
  const double* fs = a.fsteps + io * a.N_gait * 12;
  double* outp = a.out + io * 24 * N;
  double* st = a.st + (size_t)b * kMpcStItems * T;
  const int num_iter = (a.num_iter ? a.num_iter[b] : a.num_iter_scalar) + seq_s;
  const bool first = (num_iter == 0);
  bool parked = false;  // PRE: this chunk ended before the solve did
#define ST(item) st[(item)*T + tid]

  // (agent-scope atomic load: in a sequence launch the flag may have been set by another workgroup of this launch)
  if (!first && !__hip_atomic_load(&a.flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {  // reference would dereference an un-setup OSQP workspace
    if (tid == 0) {
      a.status[b] = kStatusNotSetup;
      a.iters[b] = 0;
    }
    if (act) {
#pragma unroll
      for (int t = 0; t < 3; t++) {
        outp[(3 * j + t) * N + k] = nan("");
        outp[(12 + 3 * j + t) * N + k] = nan("");
      }
    }
    if (SEQ && tid == 0 && a.seq_iters) a.seq_iters[io] = 0;
  } else {
  // =========================== A. assemble (MPC.cpp:626-640) ===========================
  // construct_gait (:686-701): rows until the first all-zero row of fsteps
  double f3[3] = {0, 0, 0};
  if (act) {
#pragma unroll
    for (int t = 0; t < 3; t++) f3[t] = fs[k * 12 + 3 * j + t];
  }
  const bool mynz = act && (f3[0] != 0.0 || f3[1] != 0.0 || f3[2] != 0.0);
  const unsigned long long bal = __ballot(mynz);
  // update_ML / construct_S (:422, :669) walk the GAIT matrix to its first all-zero row, and a gait entry is 0 where the
  // foothold's x is 0 (:691): a row of fsteps whose four x entries are 0 but which is not all zero (only y / z set) does
  // not end construct_gait, yet it ends the rewriting of B blocks and S flags -- `stop` <= `len`.
  const unsigned long long balx = __ballot(act && f3[0] != 0.0);
  int len = N, stop = N;
  if constexpr (NW == 1) {
    for (int kk = N - 1; kk >= 0; kk--) {
      if (((bal >> (4 * kk)) & 0xFull) == 0) len = kk;
      if (((balx >> (4 * kk)) & 0xFull) == 0) stop = kk;
    }
  } else {
    if (lane == 0) L.sBal[wv] = bal;
    __syncthreads();
    for (int kk = N - 1; kk >= 0; kk--)
      if (((L.sBal[kk >> 4] >> (4 * (kk & 15))) & 0xFull) == 0) len = kk;
    __syncthreads();
    if (lane == 0) L.sBal[wv] = balx;
    __syncthreads();
    for (int kk = N - 1; kk >= 0; kk--)
      if (((L.sBal[kk >> 4] >> (4 * (kk & 15))) & 0xFull) == 0) stop = kk;
    __syncthreads();
  }
  const bool in_table = k < len;   // rows construct_gait rewrites
  const bool in_gait = k < stop;   // rows update_ML / construct_S rewrite

  double Bang[3][3];  // B[9+r][3j+t] of step k (MPC.cpp:440)
  double sfl[3];      // S_gait entries of (k, foot j) (MPC.cpp:665-681)
  double rho;
  double xX[3], xF[3], zD[3], zC[5], yD[3], yS[3], yC[5];
  if (first && !resumed) {
    rho = 0.1;
#pragma unroll
    for (int t = 0; t < 3; t++) xX[t] = xF[t] = zD[t] = yD[t] = yS[t] = sfl[t] = 0.0;
#pragma unroll
    for (int c = 0; c < 5; c++) zC[c] = yC[c] = 0.0;
  } else {
    // (PRE, resumed: the same slots hold the ADMM loop variables of the parked solve -- xh in x, theta / eta in y, zeta in
    // z -- and B / S as the parked chunk used them, also for a first call)
    rho = ST(kStRho);
#pragma unroll
    for (int t = 0; t < 3; t++) {
      xX[t] = ST(kStXX + t); xF[t] = ST(kStXF + t); zD[t] = ST(kStZD + t);
      yD[t] = ST(kStYD + t); yS[t] = ST(kStYS + t); sfl[t] = ST(kStS + t);
#pragma unroll
      for (int r = 0; r < 3; r++) Bang[r][t] = ST(kStB + r * 3 + t);
    }
#pragma unroll
    for (int c = 0; c < 5; c++) { zC[c] = ST(kStZC + c); yC[c] = ST(kStYC + c); }
    if constexpr (PRE) {
      if (aborted) {
        rho = 0.1;
#pragma unroll
        for (int t = 0; t < 3; t++) xX[t] = xF[t] = zD[t] = yD[t] = yS[t] = 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) zC[c] = yC[c] = 0.0;
      }
    }
  }

  // B block: dt * I_inv * skew(lever) (MPC.cpp:213-232 first call, :425-447 afterwards)
  if (act && (first || in_gait)) {
    const double yaw = xr[5 * (N + 1) + k];
    const double c = cos(yaw), s = sin(yaw);
    const double gI[9] = {3.09249e-2, -8.00101e-7, 1.865287e-5, -8.00101e-7, 5.106100e-2,
                          1.245813e-4, 1.865287e-5, 1.245813e-4, 6.939757e-2};
    const double R[9] = {c, -s, 0.0, s, c, 0.0, 0.0, 0.0, 1.0};
    double T[9], M[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        double acc = 0;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc += R[kk * 3 + i] * gI[kk * 3 + jj];
        T[i * 3 + jj] = acc;
      }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        double acc = 0;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc += T[i * 3 + kk] * R[kk * 3 + jj];
        M[i * 3 + jj] = acc;
      }
    const double c00 = M[4] * M[8] - M[5] * M[7], c10 = M[5] * M[6] - M[3] * M[8], c20 = M[3] * M[7] - M[4] * M[6];
    const double invdet = 1.0 / (M[0] * c00 + M[1] * c10 + M[2] * c20);
    double Ii[9];
    Ii[0] = c00 * invdet; Ii[1] = (M[2] * M[7] - M[1] * M[8]) * invdet; Ii[2] = (M[1] * M[5] - M[2] * M[4]) * invdet;
    Ii[3] = c10 * invdet; Ii[4] = (M[0] * M[8] - M[2] * M[6]) * invdet; Ii[5] = (M[2] * M[3] - M[0] * M[5]) * invdet;
    Ii[6] = c20 * invdet; Ii[7] = (M[1] * M[6] - M[0] * M[7]) * invdet; Ii[8] = (M[0] * M[4] - M[1] * M[3]) * invdet;
    double l[3];
    if (first) {  // default footholds, NO offset_CoM (MPC.cpp:24,223)
      const double fx = (j < 2) ? 0.19 : -0.19, fy = (j & 1) ? -0.15005 : 0.15005;
      l[0] = fx - xr[0 * (N + 1) + k];
      l[1] = fy - xr[1 * (N + 1) + k];
      l[2] = 0.0 - xr[2 * (N + 1) + k];
    } else {  // fsteps row k, offset_CoM = (0,0,-0.03) (MPC.cpp:21,438)
      l[0] = f3[0] - (xr[0 * (N + 1) + k] + 0.0);
      l[1] = f3[1] - (xr[1 * (N + 1) + k] + 0.0);
      l[2] = f3[2] - (xr[2 * (N + 1) + k] + -0.03);
    }
    const double S[9] = {0.0, -l[2], l[1], l[2], 0.0, -l[0], -l[1], l[0], 0.0};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int t = 0; t < 3; t++) {
        double acc = 0;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) acc += Ii[r * 3 + kk] * S[kk * 3 + t];
        Bang[r][t] = dt * acc;
      }
  } else if (first) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int t = 0; t < 3; t++) Bang[r][t] = 0.0;
  }
  if (act && in_gait) {
    const double sv = (f3[0] == 0.0) ? 1.0 : 0.0;  // 1 - gait[k][j]
#pragma unroll
    for (int t = 0; t < 3; t++) sfl[t] = sv;
  }
  // gait matrix for the getter (MPC.cpp:686-701): rows < len rewritten, row len zeroed
  {
    int* gg = a.gait + (size_t)b * a.N_gait * 4;
    if (first)
      for (int e = tid; e < a.N_gait * 4; e += T) gg[e] = 0;
    __syncthreads();
    if (act && in_table) gg[k * 4 + j] = (f3[0] == 0.0) ? 0 : 1;
    if (act && k == len && len < a.N_gait) gg[k * 4 + j] = 0;
    if (len == N && N < a.N_gait && tid < 4) gg[N * 4 + tid] = 0;
  }

  // bounds of the dynamics rows (MPC.cpp:476-486): u = -g - A x0 (first block) + D vec(xref[:,1:])
  double uD0[3];
  if (act) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int i = 3 * j + t;
      const double xk = xr[i * (N + 1) + k];
      const double xk1 = xr[i * (N + 1) + k + 1];
      double am = -xk;
      if (j < 2) am += -dt * xr[(i + 6) * (N + 1) + k];
      const double gterm = (i == 8) ? g8 : 0.0;
      uD0[t] = (k == 0) ? ((gterm + am) + xk1) : (gterm + (am + xk1));
    }
  } else {
#pragma unroll
    for (int t = 0; t < 3; t++) uD0[t] = 0.0;
  }

  // =========================== B. Ruiz equilibration (OSQP scale_data) ===========================
  double Dx0[3] = {1, 1, 1}, Df0[3] = {1, 1, 1}, Ed[3] = {1, 1, 1}, Es[3] = {1, 1, 1}, Ec[5] = {1, 1, 1, 1, 1};
  double cs = 1.0;
  double aB[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int t = 0; t < 3; t++) aB[r][t] = fabs(Bang[r][t]);
  const double inv_n = 1.0 / (double)(24 * N);
  for (int pass = 0; pass < 10; pass++) {
    double nX[3], nF[3], nD[3], nS[3], nC[5];
    double EdL[3], EdA[3], mDf[3], mBD[3];
    double EnV[3], En6V[3], DxpV[3], Dxp6V[3];
    nb_next<NW, !FULL>(Ed, EnV, En6V, L.sE, k, j, lane, has_next);
    nb_prev<NW, !FULL>(Dx0, DxpV, Dxp6V, L.sE, k, j, lane, has_prev);
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const double En = EnV[t], En6 = En6V[t], Dxp = DxpV[t], Dxp6 = Dxp6V[t];
      EdL[t] = quad_bcast<2>(Ed[t]);
      EdA[t] = quad_bcast<3>(Ed[t]);
      mDf[t] = quad_max(Df0[t]);
      double mb = fmax(fmax(aB[t][0] * Df0[0], aB[t][1] * Df0[1]), aB[t][2] * Df0[2]);
      mBD[t] = quad_max(mb);
      // column of X_k[3j+t]
      double v = fabs(cs * wX[t] * Dx0[t] * Dx0[t]);
      v = fmax(v, Ed[t] * Dx0[t]);
      if (has_next) {
        v = fmax(v, En * Dx0[t]);
        if (j >= 2) v = fmax(v, dt * En6 * Dx0[t]);
      }
      nX[t] = v;
      // dynamics row (k, 3j+t)
      double r_ = Ed[t] * Dx0[t];
      if (has_prev) {
        r_ = fmax(r_, Ed[t] * Dxp);
        if (j < 2) r_ = fmax(r_, dt * Ed[t] * Dxp6);
      }
      nD[t] = r_;
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
      if (j == 2) nD[t] = fmax(nD[t], Ed[t] * dtm * mDf[t]);
      if (j == 3) nD[t] = fmax(nD[t], Ed[t] * mBD[t]);
      double v = fabs(cs * wF * Df0[t] * Df0[t]);
      v = fmax(v, dtm * EdL[t] * Df0[t]);
#pragma unroll
      for (int r = 0; r < 3; r++) v = fmax(v, aB[r][t] * EdA[r] * Df0[t]);
      v = fmax(v, sfl[t] * Es[t] * Df0[t]);
      nF[t] = v;
      nS[t] = sfl[t] * Es[t] * Df0[t];
    }
    nF[0] = fmax(nF[0], fmax(Ec[0], Ec[1]) * Df0[0]);
    nF[1] = fmax(nF[1], fmax(Ec[2], Ec[3]) * Df0[1]);
    nF[2] = fmax(nF[2], fmax(fmax(fmax(mu * Ec[0], mu * Ec[1]), fmax(mu * Ec[2], mu * Ec[3])), Ec[4]) * Df0[2]);
    nC[0] = Ec[0] * fmax(Df0[0], mu * Df0[2]);
    nC[1] = Ec[1] * fmax(Df0[0], mu * Df0[2]);
    nC[2] = Ec[2] * fmax(Df0[1], mu * Df0[2]);
    nC[3] = Ec[3] * fmax(Df0[1], mu * Df0[2]);
    nC[4] = Ec[4] * Df0[2];
    double colsum = 0.0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      // rsqrt (1-2 ulp) instead of OSQP's 1.0 / sqrt(): the equilibration is a preconditioner, the iterates and the
      // iteration counts still agree with the oracle (tests + soak), and 140 square roots and divisions per solve go
      Dx0[t] *= rsqrt(limit_scaling(nX[t]));
      Df0[t] *= rsqrt(limit_scaling(nF[t]));
      Ed[t] *= rsqrt(limit_scaling(nD[t]));
      Es[t] *= rsqrt(limit_scaling(nS[t]));
      colsum += fabs(cs * wX[t] * Dx0[t] * Dx0[t]) + fabs(cs * wF * Df0[t] * Df0[t]);
    }
#pragma unroll
    for (int c = 0; c < 5; c++) Ec[c] *= rsqrt(limit_scaling(nC[c]));
    // cost normalisation: c_temp = max(mean ||P cols||inf, ||q||inf -> 1 because q = 0)
    double ct = block_sum<NW>(act ? colsum : 0.0, L.sRed, wv, lane) * inv_n;
    ct = fmax(ct, 1.0);
    ct = limit_scaling(ct);
    cs *= 1.0 / ct;
  }
  const double cinv = 1.0 / cs;
  double iDx[3], iDf[3];
#pragma unroll
  for (int t = 0; t < 3; t++) { iDx[t] = 1.0 / Dx0[t]; iDf[t] = 1.0 / Df0[t]; }
  // The loop carries OSQP's scaled iterates in a form that needs no scaling products per iteration (same recursion):
  //   xh   = D xbar                         (the unscaled primal iterate)
  //   eta  = E ybar                         for every row;  theta = eta - zeta_u for the dynamics rows
  //   zeta = rho_row E zbar                 for the cone rows (the dynamics rows sit on their bound zeta_u = Omega u after
  //                                         the first projection, the force-enable rows on 0)
  // with Omega = rho_row E^2, so that A'(rho z - y) = A_hat' (zeta - eta), zeta+ = clip(alpha Omega A_hat x~ +
  // (1-alpha) zeta + eta, Omega l, Omega u) and eta+ = (alpha Omega A_hat x~ + (1-alpha) zeta + eta) - zeta+.
  double xhX[3], xhF[3], thD[3], etS[3], zeC[5], etC[5];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    xhX[t] = resumed ? xX[t] : Dx0[t] * xX[t];
    xhF[t] = resumed ? xF[t] : Df0[t] * xF[t];
    etS[t] = resumed ? yS[t] : Es[t] * yS[t];
  }
#pragma unroll
  for (int c = 0; c < 5; c++) etC[c] = resumed ? yC[c] : Ec[c] * yC[c];
  // per-lane constants of the current rho (set with every factorisation)
  double aOD[3], kD[3], zeU[3], aOS[3], aOC[5], sDx2[3], sDf2[3], lz4 = 0.0;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    aOD[t] = kD[t] = zeU[t] = aOS[t] = thD[t] = 0.0;
    sDx2[t] = sigma * iDx[t] * iDx[t]; sDf2[t] = sigma * iDf[t] * iDf[t];
  }
#pragma unroll
  for (int c = 0; c < 5; c++) aOC[c] = zeC[c] = 0.0;
  const double aVel = (j >= 2) ? alpha : 0.0;
  double wD[3];  // zeta - eta of the dynamics rows, as the next right-hand side needs it (first iteration: warm start)
  // wD of the next horizon step (same foot / foot - 2): exchanged as soon as wD is final, one iteration ahead of its use
  // in the right-hand side, so that the cross-lane round trip overlaps the rest of the update
  double wnV[3] = {0.0, 0.0, 0.0}, wn6V[3] = {0.0, 0.0, 0.0};
  double rho_used = 1.0;  // rho of the current factorisation and loop constants

  // factor data (per lane: D_j^-1 of the foot's force block, two rows of the step's K^-1, 3 rows of Delta_k^-1)
  // Delta^-1 rows are read once per iteration: pinned in accumulation registers (AccD) so that the register allocator
  // keeps the architectural ones for values the loop touches more often
  double Dinv[6], Kr[2][6];
  AccD Di[3][12];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int c = 0; c < 12; c++) Di[t][c].set(0.0);
#pragma unroll
  for (int c = 0; c < 6; c++) Dinv[c] = Kr[0][c] = Kr[1][c] = 0.0;
  const int kr1 = (j < 2) ? j + 4 : j;  // lanes 0,1 of a quad also take rows 4,5 of K^-1 (lanes 2,3: a discarded duplicate)

  // =========================== C/D. factor + ADMM loop (OSQP osqp_solve) ===========================
  bool need_factor = true;
  int iter = 0, status = kStatusUnsolved, rho_updates = 0;
  double pri_res = 0.0, dua_res = 0.0, last_np = 0.0, last_nd = 0.0;
  const int max_iter = 4000;
  rho = fmin(fmax(rho, kRhoMin), kRhoMax);
  {  // loop variables from the warm start (OSQP's scaled x, z, y of the previous solve, used as they are)
    // (PRE, resumed: rho is the value the solve was parked with, possibly just adapted; the loop constants and zeta_u belong
    // to the rho of the parked factorisation, kept in the cost-scale slot: the factor block below re-bases them exactly as
    // the uninterrupted solve does at this iteration)
    rho_used = resumed ? ST(kStC) : rho;
    const double rho_eq0 = kRhoEqOverIneq * rho_used;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      zeU[t] = (rho_eq0 * Ed[t] * Ed[t]) * uD0[t];
      const double eta = Ed[t] * yD[t], ze0 = rho_eq0 * Ed[t] * zD[t];
      // first iteration: zbar is the warm start, not yet the bound: the right-hand side takes zeta0 - eta0, and eta is
      // shifted so that the relaxed update sees (1-alpha) zeta0 where later iterations see (1-alpha) zeta_u
      const double th0 = (eta + (1.0 - alpha) * (ze0 - zeU[t])) - zeU[t];
      thD[t] = resumed ? yD[t] : th0;
      wD[t] = act ? (resumed ? -yD[t] : ze0 - eta) : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 5; c++) zeC[c] = resumed ? zC[c] : rho * Ec[c] * zC[c];
    if (resumed) rho_updates = __hip_atomic_load(&a.rho_updates[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // PRE: a solve parked with an unchanged rho resumes with the loop constants it had: the re-basing of theta / wD in the
  // factor block must then be the identity.  As written it is not: `zu_new - zeU` is contracted into one fma, which returns
  // the rounding error of the product instead of 0 (harmless in a fresh solve, where the uninterrupted run does the same)
  bool same_rho_resume = resumed && (rho == rho_used);
  // values only the factorisation, the termination check and the exit read: parked in accumulation registers, so that
  // they do not compete with the loop's own values for the architectural ones
  AccD pEd[3], pEs[3], pEc[5], pIDx[3], pIDf[3], pUD0[3], pWX[3], pCs;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    pEd[t].set(Ed[t]); pEs[t].set(Es[t]); pIDx[t].set(iDx[t]); pIDf[t].set(iDf[t]); pUD0[t].set(uD0[t]); pWX[t].set(wX[t]);
  }
#pragma unroll
  for (int c = 0; c < 5; c++) pEc[c].set(Ec[c]);
  pCs.set(cs);
#define QRW_UNPARK()                                                                                        \
  double Ed[3], Es[3], Ec[5], iDx[3], iDf[3], uD0[3], wX[3];                                                \
  _Pragma("unroll") for (int t = 0; t < 3; t++) {                                                           \
    Ed[t] = pEd[t].get(); Es[t] = pEs[t].get(); iDx[t] = pIDx[t].get(); iDf[t] = pIDf[t].get();              \
    uD0[t] = pUD0[t].get(); wX[t] = pWX[t].get();                                                           \
  }                                                                                                         \
  _Pragma("unroll") for (int c = 0; c < 5; c++) Ec[c] = pEc[c].get();                                       \
  const double cs = pCs.get();

  PH(8);
  // (PRE: from here on it_resume is the iteration this workgroup's CURRENT time slice began at -- it moves when a slice ends
  // without a taker being granted and the solve goes on in place; no second variable: the N = 32 kernels have no scalar to spare)
  for (iter = it_resume + 1; iter <= max_iter; iter++) {
    PH(9);
    if (need_factor) {
      need_factor = false;
      QRW_UNPARK()
      const double rho_eq = kRhoEqOverIneq * rho;
      double omD[3], omS[3], omC[5];
#pragma unroll
      for (int t = 0; t < 3; t++) { omD[t] = rho_eq * Ed[t] * Ed[t]; omS[t] = rho_eq * Es[t] * Es[t]; }
#pragma unroll
      for (int c = 0; c < 5; c++) omC[c] = rho * Ec[c] * Ec[c];
      {  // loop constants of this rho; eta is independent of rho, zeta scales with it (no-ops on the first pass)
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const double zu_new = omD[t] * uD0[t];
          double shift = zu_new - zeU[t];
          if constexpr (PRE) shift = same_rho_resume ? 0.0 : shift;
          thD[t] -= shift;
          wD[t] = act ? wD[t] + shift : 0.0;
          zeU[t] = zu_new;
          aOD[t] = (j >= 2) ? 0.0 : alpha * omD[t];  // velocity rows take alpha dK instead (see the update)
          kD[t] = alpha * zu_new;
          aOS[t] = alpha * omS[t] * sfl[t];
        }
        const double rs = rho / rho_used;
#pragma unroll
        for (int c = 0; c < 5; c++) {
          zeC[c] *= rs;
          aOC[c] = alpha * omC[c];
        }
        rho_used = rho;
        same_rho_resume = false;
        lz4 = omC[4] * -25.0;  // f_z <= 25 (MPC.cpp:293-300); the other cone rows have l = -inf
      }
      double omL[3], omA[3], Kinv[6][6];
#pragma unroll
      for (int t = 0; t < 3; t++) { omL[t] = quad_bcast<2>(omD[t]); omA[t] = quad_bcast<3>(omD[t]); }
      force_block_factor(act, Bang, sfl, iDf, omL, omA, omS, omC, cs, wF, sigma, dtm, mu, Dinv, Kinv);
      wg_sync();
      if (act) {
        if (j == 0) {
#pragma unroll
          for (int c = 0; c < 6; c++)
#pragma unroll
            for (int c2 = 0; c2 < 6; c2++) L.sW[k * kWSz + c * 6 + c2] = Kinv[c][c2];
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
          L.sOm[k * 12 + 3 * j + t] = omD[t];
          L.sDg[k * 12 + 3 * j + t] = cs * wX[t] + sigma * iDx[t] * iDx[t];
        }
      }
      wg_sync();
#pragma unroll
      for (int c = 0; c < 6; c++) { Kr[0][c] = L.sW[k * kWSz + j * 6 + c]; Kr[1][c] = L.sW[k * kWSz + kr1 * 6 + c]; }
      {
        double DiV[3][12];
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
          for (int c = 0; c < 12; c++) DiV[t][c] = Di[t][c].get();
        chain_factorize<T>(L, N, dt, tid, k, j, DiV);
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
          for (int c = 0; c < 12; c++) Di[t][c].set(DiV[t][c]);
      }
      // wD was re-based above
      nb_next<NW, !FULL>(wD, wnV, wn6V, L.sE, k, j, lane, has_next);
    PH(0);
    }  // need_factor

    // ---- 1. hatted right-hand side r = sigma x / D + A' E (rho z - y)      (OSQP compute_rhs + KKT reduction)
    double wC[5];
#pragma unroll
    for (int c = 0; c < 5; c++) wC[c] = zeC[c] - etC[c];
    double rX[3], rF[3], coneT[3];
    cone_apply_t(wC, mu, coneT);
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const double wn = wnV[t], wn6 = wn6V[t];
      const double v = mN * wn + mN6 * wn6 - wD[t];
      rX[t] = sDx2[t] * xhX[t] + v;
    }
    {
      double wL[3], wA[3];
#pragma unroll
      for (int t = 0; t < 3; t++) { wL[t] = quad_bcast<2>(wD[t]); wA[t] = quad_bcast<3>(wD[t]); }
#pragma unroll
      for (int t = 0; t < 3; t++) {
        double v = dtm * wL[t] - sfl[t] * etS[t] + coneT[t];
#pragma unroll
        for (int r = 0; r < 3; r++) v += Bang[r][t] * wA[r];
        rF[t] = sDf2[t] * xhF[t] + v;
      }
    }
    PH(1);
    // ---- 2. eliminate forces: r_X[v] -= g_k - g_{k+1}, g_k = Gbar F^-1 r_f,k = -K^-1 B D^-1 r_f,k
    double tF[3], yK[6];  // D^-1 r_f and K^-1 B D^-1 r_f: kept for the back-substitution
    {
      tF[0] = Dinv[0] * rF[0] + Dinv[1] * rF[1] + Dinv[2] * rF[2];
      tF[1] = Dinv[1] * rF[0] + Dinv[3] * rF[1] + Dinv[4] * rF[2];
      tF[2] = Dinv[2] * rF[0] + Dinv[4] * rF[1] + Dinv[5] * rF[2];
      double sK[6];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        sK[t] = dtm * quad_sum(act ? tF[t] : 0.0);
        sK[3 + t] = quad_sum(act ? Bang[t][0] * tF[0] + Bang[t][1] * tF[1] + Bang[t][2] * tF[2] : 0.0);
      }
      kinv_apply(Kr, sK, yK);
      double gsV[3], gnV[3], gdum[3];
#pragma unroll
      for (int t = 0; t < 3; t++) gsV[t] = -((j == 3) ? yK[3 + t] : yK[t]);
      nb_next<NW, !FULL>(gsV, gnV, gdum, L.sE, k, j, lane, has_next);
#pragma unroll
      for (int t = 0; t < 3; t++) rX[t] += mGN * gnV[t] - mG * gsV[t];
    }
    PH(2);
    // ---- 3. block-tridiagonal solve (twisted block LDL', sweeps on the FP64 VALU with DPP row broadcasts)
    {
      if (act) {
#pragma unroll
        for (int t = 0; t < 3; t++) L.sX[kx * 12 + 3 * j + t] = rX[t];
      }
      wg_sync();
      // forward sweeps of the twisted factorisation (both chains at once), in place in sX
      if (wv == 0) chain_forward<FULL ? 16 * NW : 0>(L.sN, L.sX, L.sDump, N, lane);
      wg_sync();
      PH(3);
      {  // v_k = Delta_k^-1 u_k (each quad its own step, in parallel)
        double u[12], v[3];
#pragma unroll
        for (int c = 0; c < 12; c++) u[c] = L.sX[kx * 12 + c];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          double s_ = 0.0, dr[12];
          AccD::get12(Di[t], dr);
#pragma unroll
          for (int c = 0; c < 12; c++) s_ += dr[c] * u[c];
          v[t] = s_;
        }
        // (no workgroup barrier: a quad reads and overwrites only its own step's slot, and the LDS executes a wavefront's
        // operations in order -- the compiler alone must not move the stores above the loads)
        asm volatile("" ::: "memory");
        if (act) {
#pragma unroll
          for (int t = 0; t < 3; t++) L.sX[kx * 12 + 3 * j + t] = v[t];
        }
      }
      wg_sync();
      PH(4);
      // backward sweeps from the root outwards (transposed reads of the same slots)
      if (wv == 0) chain_backward<FULL ? 16 * NW : 0>(L.sN, L.sX, L.sDump, N, lane);
      wg_sync();
      PH(5);
    }
    // ---- 4. back-substitute forces, apply A, update the iterates
    double dV[6], fh[3], xh[3], xpi[3], xpi6[3], dK[6];
    {
      const double* xc = &L.sX[kx * 12];
      const double* xp = &L.sX[kpx * 12];
#pragma unroll
      for (int c = 0; c < 6; c++) dV[c] = xc[6 + c] - mP * xp[6 + c];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        xh[t] = xc[3 * j + t];
        xpi[t] = xp[3 * j + t];
        xpi6[t] = xp[3 * ((j + 2) & 3) + t];
      }
      // f = F^-1 (r_f - Gbar' dV) = D^-1 r_f - D^-1 B' (K^-1 B D^-1 r_f - K^-1 dV)
      double wB[3];
      kinv_apply(Kr, dV, dK);
#pragma unroll
      for (int c = 0; c < 6; c++) dK[c] = yK[c] - dK[c];
#pragma unroll
      for (int t = 0; t < 3; t++) wB[t] = dtm * dK[t] + Bang[0][t] * dK[3] + Bang[1][t] * dK[4] + Bang[2][t] * dK[5];
      fh[0] = tF[0] - (Dinv[0] * wB[0] + Dinv[1] * wB[1] + Dinv[2] * wB[2]);
      fh[1] = tF[1] - (Dinv[1] * wB[0] + Dinv[3] * wB[1] + Dinv[4] * wB[2]);
      fh[2] = tF[2] - (Dinv[2] * wB[0] + Dinv[4] * wB[1] + Dinv[5] * wB[2]);
#pragma unroll
      for (int t = 0; t < 3; t++) fh[t] = act ? fh[t] : 0.0;
    }
    {
      double cv[5];
      cone_apply(fh, mu, cv);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        // dynamics row of A_hat x~.  Position rows (feet 0,1): x_{k-1} + dt v_{k-1} - x_k.  Velocity rows (feet 2,3):
        // v_{k-1} + B f - v_k, and with f from the elimination B f = dV + Omega^-1 (K^-1 B D^-1 r_f - K^-1 dV), so the
        // row is Omega^-1 dK and alpha Omega (row) = alpha dK: no sum over the feet is needed (aOD is 0 on these lanes)
        const double v = mP * xpi[t] + mP6 * xpi6[t] - xh[t];
        const double dsel = (j == 3) ? dK[3 + t] : dK[t];
        xhX[t] = alpha * xh[t] + (1.0 - alpha) * xhX[t];
        xhF[t] = alpha * fh[t] + (1.0 - alpha) * xhF[t];
        // equality rows: z is projected onto [u,u], so zeta stays zeta_u and only theta = eta - zeta_u moves
        thD[t] = fma(aVel, dsel, fma(aOD[t], v, thD[t])) - kD[t];
        wD[t] = act ? -thD[t] : 0.0;
        etS[t] = fma(aOS[t], fh[t], etS[t]);  // force-enable rows: z is identically 0 (l = u = 0)
      }
      nb_next<NW, !FULL>(wD, wnV, wn6V, L.sE, k, j, lane, has_next);
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const double s_ = fma(aOC[c], cv[c], (1.0 - alpha) * zeC[c]) + etC[c];
        double zn = (c == 4) ? fmax(s_, lz4) : s_;
        zn = fmin(zn, 0.0);
        etC[c] = s_ - zn;
        zeC[c] = zn;
      }
    }

    PH(6);
    // ---- 5. termination / adaptive rho (OSQP update_info, check_termination, adapt_rho)
    const bool check = (iter % 25 == 0);
    if (check) {
      PH(6);  // (diagnostic builds) everything up to here is the update phase; the check itself is accounted to slot 7
      QRW_UNPARK()
      const double cinv = fast_rcp(cs);
      // inverse scalings are only needed here (every 25 iterations): recomputed instead of held in registers
      double Dx[3], Df[3], iEd[3], iEs[3], iEc[5];
#pragma unroll
      for (int t = 0; t < 3; t++) { Dx[t] = fast_rcp(iDx[t]); Df[t] = fast_rcp(iDf[t]); iEd[t] = fast_rcp(Ed[t]); iEs[t] = fast_rcp(Es[t]); }
#pragma unroll
      for (int c = 0; c < 5; c++) iEc[c] = fast_rcp(Ec[c]);
      // OSQP's scaled z of the rows that can move (the dynamics rows sit on their bound, the force-enable rows on 0)
      double zD[3], zC[5];
#pragma unroll
      for (int t = 0; t < 3; t++) zD[t] = Ed[t] * uD0[t];
#pragma unroll
      for (int c = 0; c < 5; c++) zC[c] = zeC[c] * (iEc[c] * fast_rcp(rho_used));
      wg_sync();
      if (act) {
#pragma unroll
        for (int t = 0; t < 3; t++) L.sX[kx * 12 + 3 * j + t] = xhX[t];
      }
      wg_sync();
      double pres = 0.0, nz = 0.0, nax = 0.0, pres_s = 0.0, nz_s = 0.0, nax_s = 0.0;
      {
        double pl[3], pa[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          pl[t] = quad_sum(act ? xhF[t] : 0.0);
          pa[t] = quad_sum(act ? Bang[t][0] * xhF[0] + Bang[t][1] * xhF[1] + Bang[t][2] * xhF[2] : 0.0);
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const int i = 3 * j + t;
          double v = -xhX[t];
          if (has_prev) {
            v += L.sX[kpx * 12 + i];
            if (j < 2) v += dt * L.sX[kpx * 12 + i + 6];
          }
          if (j == 2) v += dtm * pl[t];
          if (j == 3) v += pa[t];
          const double axs = Ed[t] * v;           // (A_s x)_i
          const double rs = axs - zD[t];
          pres_s = fmax(pres_s, fabs(rs)); nz_s = fmax(nz_s, fabs(zD[t])); nax_s = fmax(nax_s, fabs(axs));
          pres = fmax(pres, fabs(iEd[t] * rs)); nz = fmax(nz, fabs(iEd[t] * zD[t])); nax = fmax(nax, fabs(iEd[t] * axs));
          const double axS = Es[t] * sfl[t] * xhF[t];  // z_S = 0
          pres_s = fmax(pres_s, fabs(axS)); nax_s = fmax(nax_s, fabs(axS));
          pres = fmax(pres, fabs(iEs[t] * axS)); nax = fmax(nax, fabs(iEs[t] * axS));
        }
        double cv[5];
        cone_apply(xhF, mu, cv);
#pragma unroll
        for (int c = 0; c < 5; c++) {
          const double axs = Ec[c] * cv[c];
          const double rs = axs - zC[c];
          pres_s = fmax(pres_s, fabs(rs)); nz_s = fmax(nz_s, fabs(zC[c])); nax_s = fmax(nax_s, fabs(axs));
          pres = fmax(pres, fabs(iEc[c] * rs)); nz = fmax(nz, fabs(iEc[c] * zC[c])); nax = fmax(nax, fabs(iEc[c] * axs));
        }
      }
      if (!act) { pres = nz = nax = 0.0; pres_s = nz_s = nax_s = 0.0; }
      pres = block_max<NW>(pres, L.sRed, wv, lane); nz = block_max<NW>(nz, L.sRed, wv, lane);
      nax = block_max<NW>(nax, L.sRed, wv, lane);
      pri_res = pres;
#ifdef QRW_TRACE_RES
      if (tid == 0 && a.prof) a.prof[(size_t)b * kMpcProfItems + 64 + (iter / 25 - 1)] = pres / (eps_abs + eps_rel * fmax(nz, nax));
#endif
      last_np = fmax(nz, nax);
      const bool pri_ok = pri_res < eps_abs + eps_rel * last_np;
      // The dual residual is only needed to terminate (primal side passed), for the rho adaptation every 200
      // iterations and for the final approximate test at max_iter: skipped otherwise (never observable then).
      const bool need_dual = pri_ok || (iter % 200 == 0) || (iter == max_iter) || (pri_res > kOsqpInfty);
      // dual side: D^-1 (P_s x + A_s' y) = c P xh + A' (E y)
      double dres = 0.0, naty = 0.0, npx = 0.0, dres_s = 0.0, naty_s = 0.0, npx_s = 0.0;
      if (need_dual) {
        double eD[3], eS[3], eC[5], cT[3];
#pragma unroll
        for (int t = 0; t < 3; t++) { eD[t] = act ? thD[t] + zeU[t] : 0.0; eS[t] = etS[t]; }
#pragma unroll
        for (int c = 0; c < 5; c++) eC[c] = etC[c];
        cone_apply_t(eC, mu, cT);
        double eL[3], eA[3];
#pragma unroll
        for (int t = 0; t < 3; t++) { eL[t] = quad_bcast<2>(eD[t]); eA[t] = quad_bcast<3>(eD[t]); }
        double enV[3], en6V[3];
        nb_next<NW, !FULL>(eD, enV, en6V, L.sE, k, j, lane, has_next);
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const double en = enV[t], en6 = en6V[t];
          double aty = -eD[t];
          if (has_next) {
            aty += en;
            if (j >= 2) aty += dt * en6;
          }
          const double px = cs * wX[t] * xhX[t];
          dres = fmax(dres, fabs(px + aty)); naty = fmax(naty, fabs(aty)); npx = fmax(npx, fabs(px));
          dres_s = fmax(dres_s, fabs(Dx[t] * (px + aty))); naty_s = fmax(naty_s, fabs(Dx[t] * aty));
          npx_s = fmax(npx_s, fabs(Dx[t] * px));
          double atf = dtm * eL[t] + sfl[t] * eS[t] + cT[t];
#pragma unroll
          for (int r = 0; r < 3; r++) atf += Bang[r][t] * eA[r];
          const double pf = cs * wF * xhF[t];
          dres = fmax(dres, fabs(pf + atf)); naty = fmax(naty, fabs(atf)); npx = fmax(npx, fabs(pf));
          dres_s = fmax(dres_s, fabs(Df[t] * (pf + atf))); naty_s = fmax(naty_s, fabs(Df[t] * atf));
          npx_s = fmax(npx_s, fabs(Df[t] * pf));
        }
        if (!act) { dres = naty = npx = 0.0; dres_s = naty_s = npx_s = 0.0; }
        dres = block_max<NW>(dres, L.sRed, wv, lane); naty = block_max<NW>(naty, L.sRed, wv, lane);
        npx = block_max<NW>(npx, L.sRed, wv, lane);
        dua_res = cinv * dres;
        last_nd = cinv * fmax(naty, npx);
      }
      bool done = false;
      if (need_dual) {
        if (pri_res > kOsqpInfty || dua_res > kOsqpInfty) {
          status = kStatusNonCvx;
          done = true;
        } else if (pri_ok && dua_res < eps_abs + eps_rel * last_nd) {
          // is_primal_infeasible / is_dual_infeasible can never fire for this QP: the cone rows have
          // l = -inf (their u'dy+ + l'dy- sum is NaN in OSQP's arithmetic) and q = 0 (q'dx = 0).
          status = kStatusSolved;
          done = true;
        }
      }
      if (done) break;
      if (iter % 200 == 0) {  // adapt_rho on the SCALED residuals (compute_rho_estimate)
        pres_s = block_max<NW>(pres_s, L.sRed, wv, lane); nz_s = block_max<NW>(nz_s, L.sRed, wv, lane);
        nax_s = block_max<NW>(nax_s, L.sRed, wv, lane); dres_s = block_max<NW>(dres_s, L.sRed, wv, lane);
        naty_s = block_max<NW>(naty_s, L.sRed, wv, lane); npx_s = block_max<NW>(npx_s, L.sRed, wv, lane);
        const double pn = pres_s / (fmax(nz_s, nax_s) + 1e-10);
        const double dn = dres_s / (fmax(naty_s, npx_s) + 1e-10);
        double rho_new = rho * sqrt(pn / (dn + 1e-10));
        rho_new = fmin(fmax(rho_new, kRhoMin), kRhoMax);
        if (rho_new > rho * 5.0 || rho_new < rho / 5.0) {
          rho = rho_new;  // osqp_update_rho: clip, refresh rho_vec, refactor
          rho_updates++;
          need_factor = true;
        }
#ifdef QRW_TRACE_RES  // diagnostics: how far from termination the solve is at every adaptive-rho test (scripts/gpu_res_trace.py)
        if (tid == 0 && a.prof && iter <= 4000) {
          double* tr = a.prof + (size_t)b * kMpcProfItems + (iter / 200 - 1) * 3;
          tr[0] = pri_res / (eps_abs + eps_rel * last_np); tr[1] = dua_res / (eps_abs + eps_rel * last_nd); tr[2] = rho;
        }
#endif
        if constexpr (PRE) {  // end of this workgroup's time slice (after the rho test, before iteration iter + 1): park the solve
          // here if a taker workgroup is granted for it, go on with the next slice in place otherwise (pre_next_task above)
          if (iter - it_resume >= a.pre_chunk && iter < max_iter) {
            if (pre_may_park<NW>(a, &L.sBal[0], tid)) parked = true;
            else it_resume = iter;
          }
          if (tid == 0) {  // how far from termination (the queue's priority level is predicted from two consecutive tests, at the park)
            double* pr = &L.sPre[(iter / 200 & 1) * 4];
            pr[0] = pri_res; pr[1] = last_np; pr[2] = dua_res; pr[3] = last_nd;
          }
        }
      }
      PH(7);
      if constexpr (PRE) {
        if (parked) break;
      }
    }
  }
  PH(7);
#ifdef QRW_PROFILE_PHASES
  if (tid == 0 && a.prof) for (int i = 0; i < 10; i++) a.prof[(size_t)b * kMpcProfItems + i] = (double)ph_acc[i];
#endif
  if (iter > max_iter) iter = max_iter;
  if (status == kStatusUnsolved && !parked) {
    // max_iter reached (4000 % 25 == 0: the residuals of the last iterate were just computed): OSQP re-checks with
    // 10x tolerances, check_termination(work, 1)
    const bool pok = pri_res < 10.0 * eps_abs + 10.0 * eps_rel * last_np;
    const bool dok = dua_res < 10.0 * eps_abs + 10.0 * eps_rel * last_nd;
    status = (pok && dok) ? kStatusSolvedInaccurate : kStatusMaxIter;
  }

  // =========================== E. results + persistent state ===========================
  {
  QRW_UNPARK()
  const bool has_sol = (status != kStatusNonCvx);
  if (act && !parked) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int i = 3 * j + t;
      const double sx = has_sol ? xhX[t] + xr[i * (N + 1) + k + 1] : nan("");
      const double sf = has_sol ? xhF[t] : nan("");
      outp[i * N + k] = sx;           // retrieve_result, MPC.cpp:573
      outp[(12 + i) * N + k] = sf;    // MPC.cpp:574
    }
  }
  // back to OSQP's scaled iterates for the persistent state (PRE, parked: the loop variables as they are, bit for bit)
#pragma unroll
  for (int t = 0; t < 3; t++) {
    xX[t] = parked ? xhX[t] : xhX[t] * iDx[t]; xF[t] = parked ? xhF[t] : xhF[t] * iDf[t];
    zD[t] = Ed[t] * uD0[t];
    yD[t] = parked ? thD[t] : (thD[t] + zeU[t]) * (1.0 / Ed[t]);
    yS[t] = parked ? etS[t] : etS[t] * (1.0 / Es[t]);
  }
#pragma unroll
  for (int c = 0; c < 5; c++) {
    zC[c] = parked ? zeC[c] : zeC[c] * (1.0 / (rho_used * Ec[c]));
    yC[c] = parked ? etC[c] : etC[c] * (1.0 / Ec[c]);
  }
  if (!has_sol) {  // store_solution(): cold start after a failed solve
#pragma unroll
    for (int t = 0; t < 3; t++) xX[t] = xF[t] = zD[t] = yD[t] = yS[t] = 0.0;
#pragma unroll
    for (int c = 0; c < 5; c++) zC[c] = yC[c] = 0.0;
  }
  ST(kStRho) = rho;
  ST(kStC) = parked ? rho_used : cs;  // (parked: the rho of the factorisation the loop variables belong to)
#pragma unroll
  for (int t = 0; t < 3; t++) {
    ST(kStXX + t) = xX[t]; ST(kStXF + t) = xF[t]; ST(kStZD + t) = zD[t];
    ST(kStYD + t) = yD[t]; ST(kStYS + t) = yS[t]; ST(kStS + t) = sfl[t];
    ST(kStDX + t) = 1.0 / iDx[t]; ST(kStDF + t) = 1.0 / iDf[t]; ST(kStED + t) = Ed[t]; ST(kStES + t) = Es[t];
#pragma unroll
    for (int r = 0; r < 3; r++) ST(kStB + r * 3 + t) = Bang[r][t];
  }
#pragma unroll
  for (int c = 0; c < 5; c++) { ST(kStZC + c) = zC[c]; ST(kStYC + c) = yC[c]; ST(kStEC + c) = Ec[c]; }
  if (PRE && parked) {
    if (tid == 0) {
      a.pause_it[b] = iter;
      a.rho_updates[b] = rho_updates;
    }
  } else if (tid == 0) {
    if constexpr (PRE) a.pause_it[b] = 0;
    a.flags[b] = 1;
    a.iters[b] = iter;
    a.status[b] = status;
    a.rho_out[b] = rho;
    a.pri[b] = pri_res;
    a.dua[b] = dua_res;
    a.rho_updates[b] = rho_updates;
    if (SEQ && a.seq_iters) a.seq_iters[io] = iter;
#ifdef QRW_SEQ_STATS  // diagnostics: duration of this task in 100 MHz ticks (a.prof holds 10 doubles per instance: K <= 5)
    if (a.prof && a.seq_K <= 5) {
      a.prof[io * 2] = (double)(__builtin_amdgcn_s_memrealtime() - task_t0);
      a.prof[io * 2 + 1] = (double)iter;
    }
#endif
  }
  }
  }  // set up
  if constexpr (SEQ) seq_finish_task<NW>(a, b, seq_s, tid);
  if constexpr (PRE) {
    int level = 0;
    if (parked && tid == 0) {  // (registers are free here: the divisions and logarithms stay out of the ADMM loop)
      const int it = a.pause_it[b];
      const double* p1 = &L.sPre[(it / 200 & 1) * 4];
      const double* p0 = &L.sPre[((it / 200 & 1) ^ 1) * 4];
      const double r1 = fmax(p1[0] / (1e-6 + 1e-6 * p1[1]), p1[2] / (1e-6 + 1e-6 * p1[3]));
      const double r0 = (p0[1] > 0.0 || p0[3] > 0.0) ? fmax(p0[0] / (1e-6 + 1e-6 * p0[1]), p0[2] / (1e-6 + 1e-6 * p0[3])) : 0.0;
      level = pre_level(a, it, it_resume, r1, r0);
    }
    pre_end_chunk<NW>(a, b, parked, level, tid);
  }
  }
#undef QRW_UNPARK
#undef ST
}

// Counting sort of the instances by decreasing iteration count (multiples of 25, at most 4000) -> order[].
// ONE wavefront: while other solves are resident (a second stream group, the asynchronous mode) every SIMD is owned by a
// 512-register solve wavefront; the 16-wavefront workgroup this used to be waited for a whole compute unit to drain
// (measured: up to 4 ms in the two-stream-group run, and the same with 4 wavefronts: a workgroup's wavefronts are spread
// over the SIMDs), one wavefront takes the first SIMD that frees.  To stay short alone it reads 16 instances per lane and
// round (four 16-byte loads in flight per array) in a compact loop: fully unrolled code, run once, was bound by
// instruction fetch (39 us against 19 us for batch 4096).
namespace {
__device__ __forceinline__ int order_key(float e) {
  const int key = (int)(e * 0.04f);
  return key < 0 ? 0 : (key > 160 ? 160 : key);
}
__device__ __forceinline__ float order_ema(float e0, float it) { return (e0 == 0.0f) ? it : e0 + (it - e0) * 0.125f; }
}  // namespace

__global__ __launch_bounds__(64) void mpc_order_kernel(const int* __restrict__ iters, float* __restrict__ ema,
                                                       int* __restrict__ order, int B) {
  // key = exponential moving average (1/8) of the instance's iteration counts: a better predictor of the next solve
  // than the last count alone (simulated makespan, scripts/gpu_lpt_sim.py: 1.12x instead of 1.17x of the lower bound)
  constexpr int U = 4, R = 64 * 4 * U;  // instances per round
  __shared__ int hist[162];
  __shared__ int offs[162];
  const int lane = threadIdx.x;
  for (int i = lane; i < 162; i += 64) hist[i] = 0;
  __syncthreads();
  const int Bv = B - B % R;  // whole rounds, vector loads (hipMalloc'd arrays: 16-byte aligned); the rest one by one
  for (int base = 0; base < Bv; base += R) {
    int4 it[U];
    float4 e[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int b = base + (u * 64 + lane) * 4;
      it[u] = *reinterpret_cast<const int4*>(iters + b);
      e[u] = *reinterpret_cast<const float4*>(ema + b);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int b = base + (u * 64 + lane) * 4;
      e[u].x = order_ema(e[u].x, (float)it[u].x);
      e[u].y = order_ema(e[u].y, (float)it[u].y);
      e[u].z = order_ema(e[u].z, (float)it[u].z);
      e[u].w = order_ema(e[u].w, (float)it[u].w);
      *reinterpret_cast<float4*>(ema + b) = e[u];
      atomicAdd(&hist[order_key(e[u].x)], 1);
      atomicAdd(&hist[order_key(e[u].y)], 1);
      atomicAdd(&hist[order_key(e[u].z)], 1);
      atomicAdd(&hist[order_key(e[u].w)], 1);
    }
  }
  for (int b = Bv + lane; b < B; b += 64) {
    const float e = order_ema(ema[b], (float)iters[b]);
    ema[b] = e;
    atomicAdd(&hist[order_key(e)], 1);
  }
  __syncthreads();
  // exclusive suffix sums over the 161 keys (largest key first): three keys per lane, wave scan of the lane totals
  {
    const int k0 = 160 - 3 * lane;  // this lane's keys k0, k0-1, k0-2 (lanes 0..53 cover 160..-1)
    int h[3], tot = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) { h[j] = (k0 - j >= 0) ? hist[k0 - j] : 0; tot += h[j]; }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    int run = incl - tot;
#pragma unroll
    for (int j = 0; j < 3; j++) { if (k0 - j >= 0) offs[k0 - j] = run; run += h[j]; }
  }
  __syncthreads();
  for (int base = 0; base < Bv; base += R) {
    float4 e[U];
#pragma unroll
    for (int u = 0; u < U; u++) e[u] = *reinterpret_cast<const float4*>(ema + base + (u * 64 + lane) * 4);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int b = base + (u * 64 + lane) * 4;
      order[atomicAdd(&offs[order_key(e[u].x)], 1)] = b;
      order[atomicAdd(&offs[order_key(e[u].y)], 1)] = b + 1;
      order[atomicAdd(&offs[order_key(e[u].z)], 1)] = b + 2;
      order[atomicAdd(&offs[order_key(e[u].w)], 1)] = b + 3;
    }
  }
  for (int b = Bv + lane; b < B; b += 64) order[atomicAdd(&offs[order_key(ema[b])], 1)] = b;
}

int mpc_order_launch(const int* iters, float* ema, int* order, int B, hipStream_t stream) {
  hipLaunchKernelGGL(mpc_order_kernel, dim3(1), dim3(64), 0, stream, iters, ema, order, B);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// preemptive launch (N > 16 only): B * pre_cmax workgroups, the caller has reset pre_queue (-1) and pre_ctr (0) on the stream
int mpc_preemptive_launch(const MpcArgs& a_in, int resident_slots, hipStream_t stream) {
  if (a_in.N <= 16 || a_in.N > kMpcMaxN || !a_in.pre_queue || !a_in.pre_ctr || !a_in.pause_it || a_in.pre_chunk < 200 || a_in.pre_cmax < 1 ||
      a_in.pre_levels < 1 || a_in.pre_levels > kPreMaxLevels || a_in.pre_bin < 1) return -1;
  MpcArgs a = a_in;
  auto go = [&](unsigned blocks) {
    if (a.N == 32) hipLaunchKernelGGL((mpc_solve_kernel<2, true, false, true>), dim3(blocks), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((mpc_solve_kernel<2, false, false, true>), dim3(blocks), dim3(128), 0, stream, a);
  };
  const unsigned takers = (unsigned)a.B * (unsigned)(a.pre_cmax - 1);
  if (a.B > resident_slots || takers == 0u) {
    // one grid: the takers (index >= B) start when index-dealt workgroups end, i.e. when first slices have been parked
    a.pre_block0 = 0;
    go((unsigned)a.B + takers);
  } else {
    // the whole batch is resident at once: takers of the same grid would all start, find nothing parked yet and leave.  Two
    // grids on the stream instead: every first slice, then the takers (each solve longer than a slice is parked once, resumed
    // by a taker and -- the other takers having left -- finished in place)
    a.pre_block0 = 0;
    go((unsigned)a.B);
    a.pre_block0 = a.B;
    go(takers);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// behind a time-sliced launch, on its stream: the launch's error word, if set, into a host-mapped word (sticky: the host clears it)
__global__ void mpc_pre_error_flush_kernel(const unsigned* pre_ctr, unsigned* host_word) {
  const unsigned code = pre_ctr[kPreErr];
  if (code != 0u) __hip_atomic_store(host_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int mpc_pre_error_flush(const unsigned* pre_ctr, unsigned* host_word, hipStream_t stream) {
  hipLaunchKernelGGL(mpc_pre_error_flush_kernel, dim3(1), dim3(1), 0, stream, pre_ctr, host_word);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int mpc_launch(const MpcArgs& a, hipStream_t stream) {
  if (a.N < 1 || a.N > kMpcMaxN) return -1;
  if (a.N == 16) hipLaunchKernelGGL((mpc_solve_kernel<1, true, false>), dim3(a.B), dim3(64), 0, stream, a);
  else if (a.N < 16) hipLaunchKernelGGL((mpc_solve_kernel<1, false, false>), dim3(a.B), dim3(64), 0, stream, a);
  else if (a.N == 32) hipLaunchKernelGGL((mpc_solve_kernel<2, true, false>), dim3(a.B), dim3(128), 0, stream, a);
  else hipLaunchKernelGGL((mpc_solve_kernel<2, false, false>), dim3(a.B), dim3(128), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// queues of a sequence.  Ranks in the handle's longest-first order decide the level of an instance (cut[] below; everything
// in the lowest level when there is no order yet).  Call 0 of the first
// `groups` ranks is dealt by workgroup index (first[]), the other call-0 tasks are queued in rank order.  One block.
__global__ void mpc_seq_init_kernel(int* queue, unsigned* qctr, int* level, int* first, const int* order, int B, int K, int groups) {
  __shared__ unsigned n_lvl[kSeqLevels], base[kSeqLevels], n_q[kSeqLevels], n_dealt[kSeqLevels];
  const int cut[kSeqLevels] = {B / 16, B / 4, B / 2, B};  // eight finer levels measured slower (748 k against 768 k solves/s)
  auto level_of = [&](int rank) {
    if (!order) return kSeqLevels - 1;
    for (int l = 0; l < kSeqLevels; l++) if (rank < cut[l]) return l;
    return kSeqLevels - 1;
  };
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int l = 0; l < kSeqLevels; l++) {
      const int lo = (l == 0) ? 0 : cut[l - 1];
      n_lvl[l] = order ? (unsigned)(cut[l] - lo) : (l == kSeqLevels - 1 ? (unsigned)B : 0u);
      base[l] = run;
      run += n_lvl[l] * (unsigned)K;
      n_q[l] = 0;
      n_dealt[l] = 0;
    }
  }
  __syncthreads();
  for (long i = threadIdx.x; i < (long)B * K; i += blockDim.x) queue[i] = -1;
  __syncthreads();
  for (int r = threadIdx.x; r < B; r += blockDim.x) {
    const int b = order ? order[r] : r;
    const int l = level_of(r);
    level[b] = l;
    if (r < groups) first[r] = b;  // call 0 of instance b goes straight to workgroup r
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // the remaining call-0 tasks, in rank order, into their levels' queues (serial: B - groups entries)
    for (int r = 0; r < B; r++) {
      const int l = level_of(r);
      if (r < groups) n_dealt[l]++;
      else queue[base[l] + n_q[l]++] = order ? order[r] : r;
    }
    for (int l = 0; l < kSeqLevels; l++) {
      qctr[l * kSeqStride] = 0u;
      qctr[l * kSeqStride + 1] = n_q[l];
      qctr[kSeqErr + 2 + l] = n_lvl[l] * (unsigned)K - n_dealt[l];
      qctr[kSeqErr + 16 + l] = base[l];
    }
    qctr[kSeqErr] = 0u;
    qctr[kSeqErr + 1] = 0u;
    qctr[kSeqProgress] = 0u;
  }
}

// one workgroup per task: K * B workgroups, of which the device keeps `a.seq_groups` resident at a time
int mpc_sequence_launch(const MpcArgs& a, hipStream_t stream) {
  if (a.N < 1 || a.N > kMpcMaxN || a.seq_K < 1 || !a.queue || !a.qctr || !a.seq_hot || !a.seq_first || a.seq_groups < 1) return -1;
  hipLaunchKernelGGL(mpc_seq_init_kernel, dim3(1), dim3(1024), 0, stream, a.queue, a.qctr, a.seq_hot, a.seq_first, a.order, a.B, a.seq_K,
                     a.seq_groups);
  const unsigned tasks = (unsigned)a.seq_K * (unsigned)a.B;
  if (a.N == 16) hipLaunchKernelGGL((mpc_solve_kernel<1, true, true>), dim3(tasks), dim3(64), 0, stream, a);
  else if (a.N < 16) hipLaunchKernelGGL((mpc_solve_kernel<1, false, true>), dim3(tasks), dim3(64), 0, stream, a);
  else if (a.N == 32) hipLaunchKernelGGL((mpc_solve_kernel<2, true, true>), dim3(tasks), dim3(128), 0, stream, a);
  else hipLaunchKernelGGL((mpc_solve_kernel<2, false, true>), dim3(tasks), dim3(128), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace qrw

// ---- self-test of what the solver's linear algebra relies on: v_fmac_f64_dpp row_newbcast semantics, the twisted
// sweeps of chain_sweep.h in the production LDS layout and the in-register Gauss-Jordan inverse -------------------------
namespace qrw {
__global__ void sweeps_selftest_kernel(const double* M, const double* r, const double* G, double* out, double* ginv) {
  __shared__ __attribute__((aligned(16))) double sN[chain_lds_doubles(16, 8)];
  __shared__ double sX[18 * 12];
  __shared__ double sDump[10 * 12];
  const int lane = threadIdx.x;
  for (int e = lane; e < chain_lds_doubles(16, 8); e += 64) sN[e] = M[e];
  for (int e = lane; e < 18 * 12; e += 64) sX[e] = r[e];
  __syncthreads();
  chain_forward<16>(sN, sX, sDump, 16, lane);
  __syncthreads();
  chain_backward<16>(sN, sX, sDump, 16, lane);
  __syncthreads();
  for (int e = lane; e < 18 * 12; e += 64) out[e] = sX[e];
  // two 12x12 inverses at once (DPP rows 0 and 1), rows 2-3 shadow them
  const int i = ((lane & 15) < 12) ? (lane & 15) : 11, which = (lane >> 4) & 1;
  double m[12];
#pragma unroll
  for (int c = 0; c < 12; c++) m[c] = G[which * 144 + i * 12 + c];
  gj_invert12(m, i);
  if (lane < 32 && (lane & 15) < 12)
#pragma unroll
    for (int c = 0; c < 12; c++) ginv[which * 144 + i * 12 + c] = m[c];
}

namespace {
// device buffers and the stream of a self-test: every allocation checked, released on every path; a private non-blocking
// stream, so that a self-test neither waits for nor stalls what other streams of the process have queued
struct SelfTestBuf {
  void* p = nullptr;
  ~SelfTestBuf() { if (p) hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes) == hipSuccess; }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};
struct SelfTestStream {
  hipStream_t s = nullptr;
  ~SelfTestStream() { if (s) hipStreamDestroy(s); }
  bool create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess; }
};
}  // namespace

int sweeps_selftest(double* max_err) {
  constexpr int N = 16, m = N / 2;
  std::vector<double> hMv(chain_lds_doubles(16, 8), 0.0), hrv(18 * 12), houtv(18 * 12), hGv(288), hGiv(288);
  double *hM = hMv.data(), *hr = hrv.data(), *hout = houtv.data(), *hG = hGv.data(), *hGi = hGiv.data();
  for (int s = 0; s < N - 1; s++)
    for (int i = 0; i < 12; i++)
      for (int c = 0; c < 12; c++) hM[chain_slot(s, m) + c * kCol + i] = 0.25 * sin(0.37 * (s * 144 + i * 12 + c) + 1.0);
  for (int e = 0; e < 18 * 12; e++) hr[e] = 0.0;
  for (int k = 0; k < N; k++)
    for (int i = 0; i < 12; i++) hr[chain_pos(k, m, N) * 12 + i] = cos(0.11 * (k * 12 + i));
  for (int w = 0; w < 2; w++)  // diagonally dominant test matrices
    for (int i = 0; i < 12; i++)
      for (int c = 0; c < 12; c++) hG[w * 144 + i * 12 + c] = ((i == c) ? 4.0 + w : 0.0) + 0.3 * sin(1.7 * (w * 144 + i * 12 + c));
  // host evaluation of the same recursions (forward, then backward on its result)
  auto Mat = [&](int slot, int i, int c) { return hM[chain_slot(slot, m) + c * kCol + i]; };
  double u[N][12], t[12];
  for (int k = 0; k < N; k++) for (int i = 0; i < 12; i++) u[k][i] = hr[chain_pos(k, m, N) * 12 + i];
  for (int k = 1; k < m; k++) for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(k - 1, i, c) * u[k - 1][c]; u[k][i] = s; }
  for (int k = N - 2; k > m; k--) { for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(m + N - 2 - k, i, c) * u[k + 1][c]; t[i] = s; } for (int i = 0; i < 12; i++) u[k][i] = t[i]; }
  for (int i = 0; i < 12; i++) { double s = u[m][i], s2 = 0; for (int c = 0; c < 12; c++) { s += Mat(m - 1, i, c) * u[m - 1][c]; s2 += Mat(N - 2, i, c) * u[m + 1][c]; } t[i] = s + s2; }
  for (int i = 0; i < 12; i++) u[m][i] = t[i];
  for (int k = m - 1; k >= 0; k--) { for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(k, c, i) * u[k + 1][c]; t[i] = s; } for (int i = 0; i < 12; i++) u[k][i] = t[i]; }
  for (int k = m + 1; k < N; k++) { for (int i = 0; i < 12; i++) { double s = u[k][i]; for (int c = 0; c < 12; c++) s += Mat(m + N - 2 - (k - 1), c, i) * u[k - 1][c]; t[i] = s; } for (int i = 0; i < 12; i++) u[k][i] = t[i]; }
  SelfTestBuf bM, br, bG, bout, bGi;
  SelfTestStream st;
  const size_t nM = hMv.size() * sizeof(double), nr = hrv.size() * sizeof(double), nG = hGv.size() * sizeof(double);
  if (!bM.alloc(nM) || !br.alloc(nr) || !bG.alloc(nG) || !bout.alloc(nr) || !bGi.alloc(nG) || !st.create()) return -1;
  hipMemcpyAsync(bM.p, hM, nM, hipMemcpyHostToDevice, st.s); hipMemcpyAsync(br.p, hr, nr, hipMemcpyHostToDevice, st.s);
  hipMemcpyAsync(bG.p, hG, nG, hipMemcpyHostToDevice, st.s);
  hipLaunchKernelGGL(sweeps_selftest_kernel, dim3(1), dim3(64), 0, st.s, bM.as<double>(), br.as<double>(), bG.as<double>(), bout.as<double>(),
                     bGi.as<double>());
  hipMemcpyAsync(hout, bout.p, nr, hipMemcpyDeviceToHost, st.s); hipMemcpyAsync(hGi, bGi.p, nG, hipMemcpyDeviceToHost, st.s);
  if (hipStreamSynchronize(st.s) != hipSuccess) return -2;
  double me = 0.0;
  for (int k = 0; k < N; k++) for (int i = 0; i < 12; i++) me = fmax(me, fabs(hout[chain_pos(k, m, N) * 12 + i] - u[k][i]) / 8.0);
  for (int w = 0; w < 2; w++)  // G * G^-1 = I
    for (int i = 0; i < 12; i++)
      for (int c = 0; c < 12; c++) {
        double s = 0.0;
        for (int q = 0; q < 12; q++) s += hG[w * 144 + i * 12 + q] * hGi[w * 144 + q * 12 + c];
        me = fmax(me, fabs(s - ((i == c) ? 1.0 : 0.0)));
      }
  if (max_err) *max_err = me;
  return (me < 1e-12) ? 0 : 1;
}
}  // namespace qrw
