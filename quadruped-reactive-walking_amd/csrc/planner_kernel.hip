// Batched planners that feed the MPC / WBC hot path — gfx950 (MI355X).  SURVEY.md §8(f) ranks 1-2.
//
// Replaces, per instance, the planner calls of one control iteration (/root/reference/scripts/Controller.py:222-236):
//   Gait::updateGait                       src/Gait.cpp:184-260   (changeGait, rollGait)
//   FootstepPlanner::updateFootsteps       src/FootstepPlanner.cpp:51-230
//   FootTrajectoryGenerator::update        src/FootTrajectoryGenerator.cpp:41-151
//   StatePlanner::computeReferenceStates   src/StatePlanner.cpp:21-61
// and writes the MPC inputs (xref, fsteps), the contact schedule (gait) and the WBC foot goals straight into
// HBM in the layouts the solver kernels read, so a control step needs no host round trip.
//
// Mapping: one thread per robot instance (the planners are a few thousand scalar operations with data-dependent
// control flow).  Persistent planner state is item-major, ps[item][instance], so the threads of a wavefront read
// and write consecutive addresses; inside the kernel the gait matrices are bit masks and the foot trajectories
// statically indexed register arrays (see below), the state is only touched at the kernel's edges.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "controller_glue.h"
#include "qrw_device.h"
#include "qrw_kernels.h"

namespace qrw {

// -DQRW_PROFILE_PRE (diagnostic build, scripts/gpu_loop_only.py): clocks per phase of control_pre_kernel, summed over its
// wavefronts and printed to stderr every 50th launch (the figures quoted in DESIGN.md 4.3, docs/HISTORY.md 4.4)
#ifdef QRW_PROFILE_PRE
__device__ unsigned long long qrw_pre_prof[16];
#define PP(k_) do { if (threadIdx.x == 0 && pp_last != 0) { const long long t__ = clock64(); atomicAdd(&qrw_pre_prof[k_], (unsigned long long)(t__ - pp_last)); pp_last = t__; } } while (0)
#else
#define PP(k_) do { } while (0)
#endif

namespace {

struct PS {  // accessor of one instance's planner state (item-major)
  double* base;
  size_t stride;  // = batch
  __device__ __forceinline__ double& operator()(int item) const { return base[(size_t)item * stride]; }
};

struct Lay {  // item offsets for a given N_gait
  int past, cur, des, cf, fs, tgt, otgt, t0s, tsw, ax, ay, pos, vel, acc, fttgt, feet, nfeet, newphase, isstatic, remain,
      qstatic, rz, total;
};
__host__ __device__ inline Lay make_layout(int Ng) {
  Lay L;
  int o = 0;
  // gait matrices: four 64-bit column masks each, kept bit-for-bit in double-sized state items (12 items instead of
  // 3 * N_gait * 4 doubles: one round trip to HBM instead of fifteen)
  L.past = o; o += 4;
  L.cur = o; o += 4;
  L.des = o; o += 4;
  L.cf = o; o += 12;
  L.fs = o; o += Ng * 12;
  L.tgt = o; o += 12;
  L.otgt = o; o += 12;
  L.t0s = o; o += 4;
  L.tsw = o; o += 4;
  L.ax = o; o += 24;
  L.ay = o; o += 24;
  L.pos = o; o += 12;
  L.vel = o; o += 12;
  L.acc = o; o += 12;
  L.fttgt = o; o += 12;
  L.feet = o; o += 4;
  L.nfeet = o; o += 1;
  L.newphase = o; o += 1;
  L.isstatic = o; o += 1;
  L.remain = o; o += 1;
  L.qstatic = o; o += 7;
  L.rz = o; o += 2;  // cos / sin of FootstepPlanner::Rz as its last updateFootsteps left it (getRz, FootstepPlanner.cpp:214,236)
  L.total = o;
  return L;
}

// ---------------------------------------------------------------------------------------------------------
// The reference keeps the three gait matrices (past / current / desired, N_gait x 4 entries that are only ever
// 0 or 1) as dense double matrices and rolls them with chains of row swaps; run literally against HBM that is
// thousands of dependent memory round trips per control iteration.  Here a matrix is four 64-bit column masks in
// registers and in the persistent state (bit i = row i; the host getters unpack them), so
// "row i is all zero", "roll rows 0..n" and the phase-duration scans are a few integer instructions.  The
// footstep table is produced row by row from registers and written once; the foot trajectory state is held in
// statically indexed register arrays (the swing-feet list becomes a 4-bit mask).
struct Gm {
  unsigned long long c[4];
};
__device__ __forceinline__ unsigned long long gm_any(const Gm& m) { return m.c[0] | m.c[1] | m.c[2] | m.c[3]; }
__device__ __forceinline__ bool rz(const Gm& m, int i) { return ((gm_any(m) >> i) & 1ull) == 0ull; }
__device__ __forceinline__ bool gbit(const Gm& m, int i, int j) {
  const unsigned long long col = (j == 0) ? m.c[0] : (j == 1) ? m.c[1] : (j == 2) ? m.c[2] : m.c[3];
  return ((col >> i) & 1ull) != 0ull;
}
__device__ __forceinline__ unsigned long long lowmask(int n) { return (n >= 64) ? ~0ull : ((1ull << n) - 1ull); }
// rows 0..n-1: new[r] = old[r+1] (r < n-1), new[n-1] = old[0]   (what "swap(0,1), swap(1,2), ..., swap(n-2,n-1)" does)
__device__ __forceinline__ void rot_up(Gm& m, int n) {
  if (n < 2) return;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const unsigned long long lo = m.c[c] & lowmask(n), hi = m.c[c] & ~lowmask(n);
    m.c[c] = hi | (lo >> 1) | ((lo & 1ull) << (n - 1));
  }
}
// rows 0..n-1: new[0] = old[n-1], new[r] = old[r-1]              ("swap(n-1,n-2), ..., swap(1,0)")
__device__ __forceinline__ void rot_down(Gm& m, int n) {
  if (n < 2) return;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const unsigned long long lo = m.c[c] & lowmask(n), hi = m.c[c] & ~lowmask(n);
    m.c[c] = hi | ((lo << 1) & lowmask(n)) | ((lo >> (n - 1)) & 1ull);
  }
}
__device__ __forceinline__ void rows_fill(Gm& m, int r0, int n, bool a, bool b, bool c, bool d) {
  const unsigned long long f = lowmask(n) << r0;
  if (a) m.c[0] |= f;
  if (b) m.c[1] |= f;
  if (c) m.c[2] |= f;
  if (d) m.c[3] |= f;
}
__device__ __forceinline__ void gm_load(const PS& s, int item, Gm& m) {
#pragma unroll
  for (int c = 0; c < 4; c++) m.c[c] = (unsigned long long)__double_as_longlong(s(item + c));
}
__device__ __forceinline__ void gm_store(const PS& s, int item, const Gm& m) {
#pragma unroll
  for (int c = 0; c < 4; c++) s(item + c) = __longlong_as_double((long long)m.c[c]);
}

// desired gait of one period (src/Gait.cpp:38-108); code 5 = walk (exists in the reference but is not reachable there)
__device__ void create_desired(Gm& des, const PlannerArgs& a, int code) {
  des.c[0] = des.c[1] = des.c[2] = des.c[3] = 0ull;
  const int Nh = (int)lround(0.5 * a.T_gait / a.dt_mpc);
  if (code == 1) { rows_fill(des, 0, Nh, 1, 0, 1, 0); rows_fill(des, Nh, Nh, 0, 1, 0, 1); }
  else if (code == 2) { rows_fill(des, 0, Nh, 1, 1, 0, 0); rows_fill(des, Nh, Nh, 0, 0, 1, 1); }
  else if (code == 3) { rows_fill(des, 0, Nh, 1, 0, 0, 1); rows_fill(des, Nh, Nh, 0, 1, 1, 0); }
  else if (code == 4) { rows_fill(des, 0, (int)lround(a.T_gait / a.dt_mpc), 1, 1, 1, 1); }
  else if (code == 5) {
    const int Nq = (int)lround(0.25 * a.T_gait / a.dt_mpc);
    rows_fill(des, 0, Nq, 0, 1, 1, 1); rows_fill(des, Nq, Nq, 1, 0, 1, 1);
    rows_fill(des, 2 * Nq, Nq, 1, 1, 0, 1); rows_fill(des, 3 * Nq, Nq, 1, 1, 1, 0);
  }
}

// Gait::initialize (src/Gait.cpp:19-36) + create_gait_f (:110-139)
__device__ void gait_init(Gm& past, Gm& cur, Gm& des, const PlannerArgs& a) {
  past.c[0] = past.c[1] = past.c[2] = past.c[3] = 0ull;
  cur = past;
  create_desired(des, a, 3);
  int i = 0;
  for (int j = 0; j < a.n_steps; j++) {
#pragma unroll
    for (int c = 0; c < 4; c++) cur.c[c] |= ((des.c[c] >> i) & 1ull) << j;
    i++;
    if (rz(des, i)) i = 0;
  }
  int index = 1;
  while (!rz(des, index)) index++;
  for (int k = 0; k < i; k++) rot_up(des, index);
}

// Gait::getPhaseDuration (src/Gait.cpp:141-182); also leaves remainingTime_.  The reference walks the rows one by one (forwards
// through the current gait and on into the desired one, backwards through the current one and on into the past one); on column
// masks each walk is "count the consecutive rows whose bit equals `value`" = a count of trailing / leading ones: no loop (round 4:
// the loops' trip counts differ per lane, a wavefront paid the longest one, ~1 k instructions per call).
__device__ __forceinline__ int trailing_ones(unsigned long long x) { return (~x == 0ull) ? 64 : (__ffsll((long long)~x) - 1); }
__device__ double phase_duration(const Gm& past, const Gm& cur, const Gm& des, const PlannerArgs& a, int i, int j, bool value,
                                 double& remain) {
  const unsigned long long ccur = (j == 0) ? cur.c[0] : (j == 1) ? cur.c[1] : (j == 2) ? cur.c[2] : cur.c[3];
  const unsigned long long cdes = (j == 0) ? des.c[0] : (j == 1) ? des.c[1] : (j == 2) ? des.c[2] : des.c[3];
  const unsigned long long cpast = (j == 0) ? past.c[0] : (j == 1) ? past.c[1] : (j == 2) ? past.c[2] : past.c[3];
  const unsigned long long vcur = value ? ccur : ~ccur, vdes = value ? cdes : ~cdes, vpast = value ? cpast : ~cpast;
  const unsigned long long nzc = gm_any(cur);
  // forwards: rows i + 1, i + 2, ... while the row is not all zero and the foot's entry equals `value` (:147-152)
  const unsigned long long fwd = (i + 1 < 64) ? ((vcur & nzc) >> (i + 1)) : 0ull;
  const int run = trailing_ones(fwd);
  int cnt = 1 + run;
  const int inext = i + run + 1;
  if (inext >= 64 || ((nzc >> inext) & 1ull) == 0ull) cnt += trailing_ones(vdes & gm_any(des));  // the walk ran into the zero row: on into the desired gait (:154-162)
  remain = (double)cnt;  // remainingTime_ (:164)
  // backwards from the ORIGINAL row: rows i - 1, i - 2, ... while the entry equals `value` (no zero-row test, :167-171)
  const unsigned long long miss = ~vcur & lowmask(i);  // rows below i whose entry differs
  const int runb = (miss == 0ull) ? i : (i - 1 - (63 - __clzll((long long)miss)));
  cnt += runb;
  if (runb == i) cnt += trailing_ones(vpast & gm_any(past));  // reached row 0: on into the past gait (:173-180)
  return (double)cnt * a.dt_mpc;
}

// Gait::updateGait = changeGait + rollGait (src/Gait.cpp:184-260); returns whether the gait matrices were rolled
__device__ void gait_update(Gm& past, Gm& cur, Gm& des, const PlannerArgs& a, int k, int code, double& newphase) {
  if (code >= 1 && code <= 5) create_desired(des, a, code);
  if (k % a.k_mpc != 0) return;
  rot_down(past, a.n_steps + 1);
  bool differ = false;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    past.c[c] = (past.c[c] & ~1ull) | (cur.c[c] & 1ull);
    differ = differ || ((cur.c[c] & 1ull) != ((cur.c[c] >> 1) & 1ull));
  }
  newphase = differ ? 1.0 : 0.0;
  int index = 1;
  while (!rz(cur, index)) index++;
  rot_up(cur, index);
#pragma unroll
  for (int c = 0; c < 4; c++) cur.c[c] = (cur.c[c] & ~(1ull << (index - 1))) | ((des.c[c] & 1ull) << (index - 1));
  index = 1;
  while (!rz(des, index)) index++;
  rot_up(des, index);
}

// pinocchio::rpy::matrixToRpy of the rotation of quaternion (x, y, z, w) [third-party definition restated]
__device__ void quat_to_rpy(const double* q, double rpy[3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
               tyz = tz * y, tzz = tz * z;
  const double R00 = 1 - (tyy + tzz), R01 = txy - twz, R10 = txy + twz, R11 = 1 - (txx + tzz), R20 = txz - twy,
               R21 = tyz + twx, R22 = 1 - (txx + tyy);
  const double m = sqrt(R21 * R21 + R22 * R22);
  const double p = atan2(-R20, m);
  double r, yw;
  if (fabs(fabs(p) - M_PI / 2) < 0.001) { r = 0.0; yw = -atan2(R01, R11); }
  else { yw = atan2(R10, R00); r = atan2(R21, R22); }
  rpy[0] = r; rpy[1] = p; rpy[2] = yw;
}

// StatePlanner::computeReferenceStates (src/StatePlanner.cpp:21-61): writes xref[12][N+1] of this instance
__device__ void state_compute(const PlannerArgs& a, const double* q7, const double* rpy, const double* v6, const double* vref6,
                              double z_average, double* X) {
  const int n = a.n_steps, ld = n + 1;
  X[0 * ld] = 0.0; X[1 * ld] = 0.0; X[2 * ld] = q7[2];
  X[3 * ld] = rpy[0]; X[4 * ld] = rpy[1]; X[5 * ld] = 0.0;
  for (int i = 0; i < 3; i++) { X[(6 + i) * ld] = v6[i]; X[(9 + i) * ld] = v6[3 + i]; }
  const double T_mpc = a.T_mpc;
  const int n_out = (a.xref_steps > 0 && a.xref_steps < n) ? a.xref_steps : n;  // a caller that does not solve this iteration
  for (int i = 0; i < n_out; i++) {
    const double dtv = (n == 1 || i == n - 1) ? T_mpc : a.dt_mpc + i * ((T_mpc - a.dt_mpc) / (n - 1));  // LinSpaced
    // one sine and one cosine per step: the reference evaluates sin / cos of this same product three times each (:39-56)
    const double yaw = vref6[5] * dtv;
    const double sy = sin(yaw), cy = cos(yaw);
    double x, y;
    if (vref6[5] != 0) {
      x = (vref6[0] * sy + vref6[1] * (cy - 1.0)) / vref6[5];
      y = (vref6[1] * sy - vref6[0] * (cy - 1.0)) / vref6[5];
    } else {
      x = vref6[0] * dtv;
      y = vref6[1] * dtv;
    }
    X[0 * ld + 1 + i] = x + 0.0;
    X[1 * ld + 1 + i] = y + 0.0;
    X[2 * ld + 1 + i] = a.h_ref + z_average;
    X[3 * ld + 1 + i] = 0.0;
    X[4 * ld + 1 + i] = 0.0;
    X[5 * ld + 1 + i] = yaw;
    X[6 * ld + 1 + i] = vref6[0] * cy - vref6[1] * sy;
    X[7 * ld + 1 + i] = vref6[0] * sy + vref6[1] * cy;
    X[8 * ld + 1 + i] = 0.0; X[9 * ld + 1 + i] = 0.0; X[10 * ld + 1 + i] = 0.0;
    X[11 * ld + 1 + i] = vref6[5];
  }
}

#define FSI(i, r, c) (L.fs + ((i)*3 + (r)) * 4 + (c))

// x^N by repeated multiplication in a fixed order (the reference's std::pow(x, n) with small integer n)
template <int N>
__device__ __forceinline__ double ipow(double x) {
  double r = x;
#pragma unroll
  for (int i = 1; i < N; i++) r *= x;
  return r;
}

// Foot trajectory state of one instance in registers (statically indexed: loops over the four feet are unrolled)
struct FootTraj {
  double t0s[4], tsw[4], ax[6][4], ay[6][4], pos[3][4], vel[3][4], acc[3][4];
};

// FootTrajectoryGenerator::updateFootPosition (src/FootTrajectoryGenerator.cpp:41-106) for foot J
template <int J>
__device__ __forceinline__ void update_foot_position(FootTraj& f, const PS& s, const Lay& L, const PlannerArgs& a, const double tf[3]) {
  constexpr int j = J;
  const double ddx0 = f.acc[0][j], ddy0 = f.acc[1][j];
  const double dx0 = f.vel[0][j], dy0 = f.vel[1][j];
  const double x0 = f.pos[0][j], y0 = f.pos[1][j];
  const double t = f.t0s[j], d = f.tsw[j], dt = a.dt_wbc;
#define P(x_, n_) ipow<n_>(x_)
  if (t < d - a.lock_time) {
    const double den1 = (2 * P((t - d), 2) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
    const double den2 = (2 * (P(t, 2) - 2 * t * d + P(d, 2)) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
#pragma unroll
    for (int ax = 0; ax < 2; ax++) {
      const double dd0 = ax ? ddy0 : ddx0, d0 = ax ? dy0 : dx0, p0 = ax ? y0 : x0, tg = tf[ax];
      double(&A)[6][4] = ax ? f.ay : f.ax;
      A[0][j] = (dd0 * P(t, 2) - 2 * dd0 * t * d - 6 * d0 * t + dd0 * P(d, 2) + 6 * d0 * d + 12 * p0 - 12 * tg) / den1;
      A[1][j] = (30 * t * tg - 30 * t * p0 - 30 * d * p0 + 30 * d * tg - 2 * P(t, 3) * dd0 - 3 * P(d, 3) * dd0 +
                 14 * P(t, 2) * d0 - 16 * P(d, 2) * d0 + 2 * t * d * d0 + 4 * t * P(d, 2) * dd0 + P(t, 2) * d * dd0) / den1;
      A[2][j] = (P(t, 4) * dd0 + 3 * P(d, 4) * dd0 - 8 * P(t, 3) * d0 + 12 * P(d, 3) * d0 + 20 * P(t, 2) * p0 -
                 20 * P(t, 2) * tg + 20 * P(d, 2) * p0 - 20 * P(d, 2) * tg + 80 * t * d * p0 - 80 * t * d * tg +
                 4 * P(t, 3) * d * dd0 + 28 * t * P(d, 2) * d0 - 32 * P(t, 2) * d * d0 - 8 * P(t, 2) * P(d, 2) * dd0) / den1;
      A[3][j] = -(P(d, 5) * dd0 + 4 * t * P(d, 4) * dd0 + 3 * P(t, 4) * d * dd0 + 36 * t * P(d, 3) * d0 -
                  24 * P(t, 3) * d * d0 + 60 * t * P(d, 2) * p0 + 60 * P(t, 2) * d * p0 - 60 * t * P(d, 2) * tg -
                  60 * P(t, 2) * d * tg - 8 * P(t, 2) * P(d, 3) * dd0 - 12 * P(t, 2) * P(d, 2) * d0) / den2;
      A[4][j] = -(2 * P(d, 5) * d0 - 2 * t * P(d, 5) * dd0 - 10 * t * P(d, 4) * d0 + P(t, 2) * P(d, 4) * dd0 +
                  4 * P(t, 3) * P(d, 3) * dd0 - 3 * P(t, 4) * P(d, 2) * dd0 - 16 * P(t, 2) * P(d, 3) * d0 +
                  24 * P(t, 3) * P(d, 2) * d0 - 60 * P(t, 2) * P(d, 2) * p0 + 60 * P(t, 2) * P(d, 2) * tg) / den1;
      A[5][j] = (2 * tg * P(t, 5) - dd0 * P(t, 4) * P(d, 3) - 10 * tg * P(t, 4) * d + 2 * dd0 * P(t, 3) * P(d, 4) +
                 8 * d0 * P(t, 3) * P(d, 3) + 20 * tg * P(t, 3) * P(d, 2) - dd0 * P(t, 2) * P(d, 5) -
                 10 * d0 * P(t, 2) * P(d, 4) - 20 * p0 * P(t, 2) * P(d, 3) + 2 * d0 * t * P(d, 5) +
                 10 * p0 * t * P(d, 4) - 2 * p0 * P(d, 5)) / den2;
    }
    s(L.fttgt + j) = tf[0];
    s(L.fttgt + 4 + j) = tf[1];
  }
  const double dz = (P((d / 2), 3) * P((d - d / 2), 3));
  const double Az0 = -a.max_height / dz, Az1 = (3 * d * a.max_height) / dz, Az2 = -(3 * P(d, 2) * a.max_height) / dz,
               Az3 = (P(d, 3) * a.max_height) / dz;
  const double ev = t + dt;
  if (t < 0.0 || t > d) {
    f.pos[0][j] = x0; f.pos[1][j] = y0;
    f.vel[0][j] = 0.0; f.vel[1][j] = 0.0;
    f.acc[0][j] = 0.0; f.acc[1][j] = 0.0;
  } else {
#pragma unroll
    for (int ax = 0; ax < 2; ax++) {
      double(&A)[6][4] = ax ? f.ay : f.ax;
      const double A0 = A[0][j], A1 = A[1][j], A2 = A[2][j], A3 = A[3][j], A4 = A[4][j], A5 = A[5][j];
      f.pos[ax][j] = A5 + A4 * ev + A3 * P(ev, 2) + A2 * P(ev, 3) + A1 * P(ev, 4) + A0 * P(ev, 5);
      f.vel[ax][j] = A4 + 2 * A3 * ev + 3 * A2 * P(ev, 2) + 4 * A1 * P(ev, 3) + 5 * A0 * P(ev, 4);
      f.acc[ax][j] = 2 * A3 + 3 * 2 * A2 * ev + 4 * 3 * A1 * P(ev, 2) + 5 * 4 * A0 * P(ev, 3);
    }
  }
  f.vel[2][j] = 3 * Az3 * P(ev, 2) + 4 * Az2 * P(ev, 3) + 5 * Az1 * P(ev, 4) + 6 * Az0 * P(ev, 5);
  f.acc[2][j] = 2 * 3 * Az3 * ev + 3 * 4 * Az2 * P(ev, 2) + 4 * 5 * Az1 * P(ev, 3) + 5 * 6 * Az0 * P(ev, 4);
  f.pos[2][j] = Az3 * P(ev, 3) + Az2 * P(ev, 4) + Az1 * P(ev, 5) + Az0 * P(ev, 6);
#undef P
}

__device__ void traj_load(FootTraj& f, const PS& s, const Lay& L) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    f.t0s[j] = s(L.t0s + j); f.tsw[j] = s(L.tsw + j);
#pragma unroll
    for (int r = 0; r < 6; r++) { f.ax[r][j] = s(L.ax + r * 4 + j); f.ay[r][j] = s(L.ay + r * 4 + j); }
#pragma unroll
    for (int r = 0; r < 3; r++) { f.pos[r][j] = s(L.pos + r * 4 + j); f.vel[r][j] = s(L.vel + r * 4 + j); f.acc[r][j] = s(L.acc + r * 4 + j); }
  }
}
__device__ void traj_store(const FootTraj& f, const PS& s, const Lay& L) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    s(L.t0s + j) = f.t0s[j]; s(L.tsw + j) = f.tsw[j];
#pragma unroll
    for (int r = 0; r < 6; r++) { s(L.ax + r * 4 + j) = f.ax[r][j]; s(L.ay + r * 4 + j) = f.ay[r][j]; }
#pragma unroll
    for (int r = 0; r < 3; r++) { s(L.pos + r * 4 + j) = f.pos[r][j]; s(L.vel + r * 4 + j) = f.vel[r][j]; s(L.acc + r * 4 + j) = f.acc[r][j]; }
  }
}

}  // namespace

// One thread per instance; see the note above on the register representation.
__device__ __forceinline__ void planner_body(const PlannerArgs& a, int b, long long& pp_last) {
  const Lay L = make_layout(a.N_gait);
  const int Ng = a.N_gait;
  PS s;
  s.base = a.ps + b;
  s.stride = (size_t)a.B;

  Gm past, cur, des;
  double newphase = 0.0, remain = 0.0;
  const bool init = (a.mode & kPlanInit) != 0;
  const bool uses_gait = (a.mode & (kPlanGait | kPlanFootsteps | kPlanTraj | kPlanOutputs)) != 0;
  if (init) {
    gait_init(past, cur, des, a);
    for (int e = 0; e < 12; e++) {
      s(L.cf + e) = a.shoulders[e]; s(L.tgt + e) = a.shoulders[e]; s(L.otgt + e) = a.shoulders[e];
      s(L.fttgt + e) = a.init_target[e]; s(L.pos + e) = a.init_pos[e]; s(L.vel + e) = 0.0; s(L.acc + e) = 0.0;
    }
    for (int e = 0; e < Ng * 12; e++) s(L.fs + e) = 0.0;
    for (int e = 0; e < 24; e++) { s(L.ax + e) = 0.0; s(L.ay + e) = 0.0; }
    for (int e = 0; e < 4; e++) { s(L.t0s + e) = 0.0; s(L.tsw + e) = 0.0; s(L.feet + e) = 0.0; }
    for (int e = 0; e < 7; e++) s(L.qstatic + e) = 0.0;
    s(L.newphase) = 0.0; s(L.isstatic) = 0.0; s(L.remain) = 0.0; s(L.nfeet) = 0.0;
    s(L.rz) = 0.0; s(L.rz + 1) = 0.0;  // Rz = 0 but (2,2) = 1 until the first updateFootsteps (FootstepPlanner.cpp:10,48)
  } else if (uses_gait) {
    gm_load(s, L.past, past);
    gm_load(s, L.cur, cur);
    gm_load(s, L.des, des);
    newphase = s(L.newphase);
    remain = s(L.remain);
  }
  bool gait_dirty = init, remain_dirty = false;
  // Operands of the later phases are read here, ahead of the first store: a load the compiler has to keep behind a
  // possibly aliasing store is one more HBM round trip on this single thread's critical path.
  double cf[3][4], fs1[3][4], otgt_z[4];
  if (a.mode & kPlanFootsteps) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) { cf[r][j] = s(L.cf + r * 4 + j); fs1[r][j] = s(FSI(1, r, j)); }
#pragma unroll
    for (int f = 0; f < 4; f++) otgt_z[f] = s(L.otgt + 8 + f);
  }
  FootTraj ft;
  double nfeet_s = 0.0, feet_s[4] = {0.0, 0.0, 0.0, 0.0};
  bool traj_loaded = false;
  PP(2);

  double q7[7] = {0, 0, 0, 0, 0, 0, 1}, hv[6] = {0, 0, 0, 0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
  if (a.q7) for (int i = 0; i < 7; i++) q7[i] = a.q7[(size_t)b * a.q_ld + i];
  if (a.hv) for (int i = 0; i < 6; i++) hv[i] = a.hv[(size_t)b * 6 + i];
  if (a.vref) for (int i = 0; i < 6; i++) vr[i] = a.vref[(size_t)b * 6 + i];
  const int code = a.code ? a.code[b] : a.code_scalar;
  const int k = a.k;
  double rpy[3] = {0.0, 0.0, 0.0};  // of the base quaternion: the footstep targets and the reference state both use it
  if ((a.mode & kPlanFootsteps) || ((a.mode & kPlanState) && a.xref)) quat_to_rpy(q7 + 3, rpy);

  // ---- Gait::updateGait
  if (a.mode & kPlanGait) {
    s(L.isstatic) = (code == 4) ? 1.0 : 0.0;
    if (code == 4)
      for (int i = 0; i < 7; i++) s(L.qstatic + i) = q7[i];
    gait_update(past, cur, des, a, k, code, newphase);
    if (k % a.k_mpc == 0) s(L.newphase) = newphase;
    gait_dirty = true;
  }

  PP(3);
  // ---- FootstepPlanner::updateFootsteps (src/FootstepPlanner.cpp:51-74) with computeTargetFootstep (:204-221),
  // computeFootsteps (:76-156), computeNextFootstep (:158-186), updateTargetFootsteps (:188-202), updateNewContact (:223-232)
  double otgt[12];
  bool have_otgt = false;
  if (a.mode & kPlanFootsteps) {
    const double* b_v = hv;
    const double* b_vref = vr;
    if (a.refresh != 0 && newphase != 0.0) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (gbit(cur, 0, j)) {
#pragma unroll
          for (int r = 0; r < 3; r++) cf[r][j] = fs1[r][j];
        }
    }
    {
      const double ry = a.dt_wbc * b_vref[5];
      const double c = cos(ry), sn = sin(ry);
      const double dpx = a.dt_wbc * b_vref[0], dpy = a.dt_wbc * b_vref[1];
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (gbit(cur, 0, j)) {
          const double x = cf[0][j] - dpx, y = cf[1][j] - dpy;
          cf[0][j] = c * x + sn * y;
          cf[1][j] = -sn * x + c * y;
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) s(L.cf + r * 4 + j) = cf[r][j];
    PP(4);
    // the table, row by row: `row` is row i-1 while row i is being built
    double row[3][4], tg[2][4];
    bool found[4] = {false, false, false, false};
    double* fo = (a.fsteps) ? a.fsteps + (size_t)b * Ng * 12 : nullptr;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 3; r++) row[r][j] = gbit(cur, 0, j) ? cf[r][j] : 0.0;
    const double w = b_vref[5];
    double dtc_prev = a.dt_wbc * a.k_footsteps;
    const double cross0 = b_v[1] * b_vref[5] - b_v[2] * b_vref[4], cross1 = b_v[2] * b_vref[3] - b_v[0] * b_vref[5];
    bool live = true;  // rows up to the first all-zero gait row are built, the rest of the table is zero
    for (int i = 0; i < Ng; i++) {
      if (i > 0) {
        live = live && !rz(cur, i);
        double nrow[3][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool gp = gbit(cur, i - 1, j), gc = gbit(cur, i, j);
#pragma unroll
          for (int r = 0; r < 3; r++) nrow[r][j] = (live && gp && gc) ? row[r][j] : 0.0;
        }
        if (live) {
          // rows in which a foot lands: one sine / cosine of the previous row's yaw for all of its feet (the reference
          // evaluates them of this same product for dx / dy and again per landing foot, src/FootstepPlanner.cpp:101-150)
          bool lands = false;
#pragma unroll
          for (int j = 0; j < 4; j++) lands = lands || (!gbit(cur, i - 1, j) && gbit(cur, i, j));
          double c = 1.0, sn = 0.0;
          if (lands) { const double yawp = w * dtc_prev; c = cos(yawp); sn = sin(yawp); }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const bool gp = gbit(cur, i - 1, j), gc = gbit(cur, i, j);
            if (!gp && gc) {
              double dxp, dyp;
              if (w != 0) {
                dxp = (b_v[0] * sn + b_v[1] * (c - 1.0)) / w;
                dyp = (b_v[1] * sn - b_v[0] * (c - 1.0)) / w;
              } else {
                dxp = b_v[0] * dtc_prev;
                dyp = b_v[1] * dtc_prev;
              }
              const double t_stance = phase_duration(past, cur, des, a, i, j, true, remain);
              remain_dirty = true;
              double nf[3];
              const double cr[3] = {cross0, cross1, 0.0};
#pragma unroll
              for (int r = 0; r < 3; r++) {
                double v = t_stance * 0.5 * b_v[r];
                v += a.k_feedback * (b_v[r] - b_vref[r]);
                v += 0.5 * sqrt(a.h_ref / a.g) * cr[r];
                nf[r] = v;
              }
              nf[0] = fmax(fmin(nf[0], a.L), -a.L);
              nf[1] = fmax(fmin(nf[1], a.L), -a.L);
              nf[0] += a.shoulders[0 * 4 + j];
              nf[1] += a.shoulders[1 * 4 + j];
              nf[2] = 0.0;
              nrow[0][j] = (c * nf[0] - sn * nf[1] + 0.0 * nf[2]) + dxp;
              nrow[1][j] = (sn * nf[0] + c * nf[1] + 0.0 * nf[2]) + dyp;
              nrow[2][j] = (0.0 * nf[0] + 0.0 * nf[1] + 1.0 * nf[2]) + 0.0;
            }
          }
          dtc_prev = dtc_prev + a.dt_mpc;  // dt_cum(i) = dt_cum(i-1) + dt for a non-zero row i
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 3; r++) row[r][j] = nrow[r][j];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
          s(FSI(i, r, j)) = row[r][j];
          if (fo) fo[i * 12 + 3 * j + r] = row[r][j];  // FootstepPlanner::getFootsteps / vectorToMatrix (:235-249)
        }
        if (!found[j] && (row[0][j] != 0.0 || i == Ng - 1)) {  // updateTargetFootsteps: first row with x != 0
          found[j] = true;
          tg[0][j] = row[0][j];
          tg[1][j] = row[1][j];
        }
      }
    }
    if (a.mode & kPlanTraj) {  // the foot-trajectory state: requested here so that it arrives during the target arithmetic
      traj_load(ft, s, L);
      nfeet_s = s(L.nfeet);
#pragma unroll
      for (int jj = 0; jj < 4; jj++) feet_s[jj] = s(L.feet + jj);
      traj_loaded = true;
    }
    PP(5);
    const double c = cos(rpy[2]), sn = sin(rpy[2]);
    s(L.rz) = c; s(L.rz + 1) = sn;
#pragma unroll
    for (int f = 0; f < 4; f++) {
      s(L.tgt + f) = tg[0][f];
      s(L.tgt + 4 + f) = tg[1][f];
      s(L.tgt + 8 + f) = 0.0;
      otgt[f] = (c * tg[0][f] - sn * tg[1][f]) + q7[0];
      otgt[4 + f] = (sn * tg[0][f] + c * tg[1][f]) + q7[1];
      s(L.otgt + f) = otgt[f];
      s(L.otgt + 4 + f) = otgt[4 + f];
    }
#pragma unroll
    for (int f = 0; f < 4; f++) otgt[8 + f] = otgt_z[f];
    have_otgt = true;
  } else if (a.fsteps && (a.mode & kPlanOutputs)) {
    double* o = a.fsteps + (size_t)b * Ng * 12;
    for (int i = 0; i < Ng; i++)
      for (int j = 0; j < 4; j++)
        for (int r = 0; r < 3; r++) o[i * 12 + 3 * j + r] = s(FSI(i, r, j));
  }

  PP(6);
  // ---- FootTrajectoryGenerator::update (src/FootTrajectoryGenerator.cpp:108-151)
  if (a.mode & kPlanTraj) {
    double tgt[12];
#pragma unroll
    for (int e = 0; e < 12; e++) tgt[e] = a.target_in ? a.target_in[(size_t)b * 12 + e] : (have_otgt ? otgt[e] : s(L.otgt + e));
    FootTraj& f = ft;
    if (!traj_loaded) {
      traj_load(f, s, L);
      nfeet_s = s(L.nfeet);
#pragma unroll
      for (int jj = 0; jj < 4; jj++) feet_s[jj] = s(L.feet + jj);
    }
    unsigned swing = 0;
    bool run = true;
    if ((k % a.k_mpc) == 0) {
      int nf = 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!gbit(cur, 0, i)) { s(L.feet + nf) = (double)i; nf++; swing |= 1u << i; }
      s(L.nfeet) = (double)nf;
      run = nf != 0;
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (swing & (1u << i)) {
          const double tsw = phase_duration(past, cur, des, a, 0, i, false, remain);
          remain_dirty = true;
          f.tsw[i] = tsw;
          const double value = tsw - (remain * a.k_mpc - ((k + 1) % a.k_mpc)) * a.dt_wbc - a.dt_wbc;
          f.t0s[i] = fmax(0.0, value);
        }
    } else {
      const int nf = (int)nfeet_s;
      run = nf != 0;
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
        if (jj < nf) swing |= 1u << (int)feet_s[jj];
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (swing & (1u << i)) f.t0s[i] = fmax(0.0, f.t0s[i] + a.dt_wbc);
    }
    if (run) {
      if (swing & 1u) { const double tf[3] = {tgt[0], tgt[4], tgt[8]}; update_foot_position<0>(f, s, L, a, tf); }
      if (swing & 2u) { const double tf[3] = {tgt[1], tgt[5], tgt[9]}; update_foot_position<1>(f, s, L, a, tf); }
      if (swing & 4u) { const double tf[3] = {tgt[2], tgt[6], tgt[10]}; update_foot_position<2>(f, s, L, a, tf); }
      if (swing & 8u) { const double tf[3] = {tgt[3], tgt[7], tgt[11]}; update_foot_position<3>(f, s, L, a, tf); }
    }
    traj_store(f, s, L);
    if (a.feet_pva) {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          a.feet_pva[(size_t)b * 36 + r * 4 + j] = f.pos[r][j];
          a.feet_pva[(size_t)b * 36 + 12 + r * 4 + j] = f.vel[r][j];
          a.feet_pva[(size_t)b * 36 + 24 + r * 4 + j] = f.acc[r][j];
        }
    }
  } else if (a.feet_pva && (a.mode & kPlanOutputs)) {
    for (int e = 0; e < 12; e++) {
      a.feet_pva[(size_t)b * 36 + e] = s(L.pos + e);
      a.feet_pva[(size_t)b * 36 + 12 + e] = s(L.vel + e);
      a.feet_pva[(size_t)b * 36 + 24 + e] = s(L.acc + e);
    }
  }
  PP(7);
  if ((a.mode & kPlanState) && a.xref) state_compute(a, q7, rpy, hv, vr, a.z_average, a.xref + (size_t)b * 12 * (a.n_steps + 1));

  PP(8);
  // ---- remaining outputs / state
  if (a.gait && (a.mode & (kPlanGait | kPlanOutputs))) {
    double* o = a.gait + (size_t)b * Ng * 4;
    for (int i = 0; i < Ng; i++)
#pragma unroll
      for (int c = 0; c < 4; c++) o[i * 4 + c] = gbit(cur, i, c) ? 1.0 : 0.0;
  }
  if (a.contacts && uses_gait)
#pragma unroll
    for (int c = 0; c < 4; c++) a.contacts[(size_t)b * 4 + c] = gbit(cur, 0, c) ? 1.0 : 0.0;
  if (a.target && (a.mode & (kPlanFootsteps | kPlanOutputs)))
    for (int e = 0; e < 12; e++) a.target[(size_t)b * 12 + e] = have_otgt ? otgt[e] : s(L.otgt + e);
  if (gait_dirty) {
    gm_store(s, L.past, past);
    gm_store(s, L.cur, cur);
    gm_store(s, L.des, des);
  }
  if (remain_dirty) s(L.remain) = remain;
  PP(9);
}

__global__ __launch_bounds__(64) void planner_kernel(PlannerArgs a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  long long pp_last = 0;  // phase accounting is for control_pre_kernel only
  planner_body(a, b, pp_last);
}

// Fused head of a control iteration (scripts/Controller.py:218-296): Controller.updateState, the four planners and --
// when the MPC result to use is already known (every iteration but the ones that solve) -- the WBC target assembly,
// one thread per instance, one launch.  The pieces hand over through their HBM operands (same thread, program order).
__global__ __launch_bounds__(64) void control_pre_kernel(ControllerArgs cu, PlannerArgs p, ControllerArgs cw, int with_wbc_inputs) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
#ifdef QRW_PROFILE_PRE
  long long pp_last = clock64();
  if (threadIdx.x == 0) atomicAdd(&qrw_pre_prof[0], 1ull);
#else
  long long pp_last = 0;
#endif
  glue::update_state(cu, b);
  PP(1);
  planner_body(p, b, pp_last);
  if (with_wbc_inputs) glue::wbc_inputs(cw, b);
  PP(10);
}


// ---- the same fused head with ONE QUAD PER INSTANCE (lane = 4 * instance + foot; 16 instances per wavefront, as wbc_kernel):
// the per-foot work -- footstep table column, foot trajectory, foot command of the WBC target assembly -- runs on the
// foot's own lane, the shared scalar work (state update, gait matrices as bit masks, yaw sines) redundantly on all four, the
// stores are dealt to the lanes, and the three pieces hand over through REGISTERS instead of through their HBM operands.
// One thread per instance was 64 wavefronts of pure latency on 1024 SIMDs (57 % of the wave cycles parked at s_waitcnt,
// DESIGN.md 4.3, docs/HISTORY.md 4.4); per lane this does about a third of the instructions and a quarter of the dependent HBM round trips.
// Arithmetic, expression by expression, is that of update_state / planner_body / wbc_inputs above (parity: the fused and
// the separate control loops are both checked against the chained CPU oracles, tests/test_gpu_controller.py).
namespace {
// FootTrajectoryGenerator::updateFootPosition for ONE foot held in scalars (the expressions of update_foot_position<J>)
struct FootLane {
  double t0s, tsw, ax[6], ay[6], pos[3], vel[3], acc[3];
};
__device__ __forceinline__ void update_foot_lane(FootLane& f, const PlannerArgs& a, const double tf[3], double& fttgt0, double& fttgt1,
                                                 bool& fttgt_dirty) {
  const double ddx0 = f.acc[0], ddy0 = f.acc[1];
  const double dx0 = f.vel[0], dy0 = f.vel[1];
  const double x0 = f.pos[0], y0 = f.pos[1];
  const double t = f.t0s, d = f.tsw, dt = a.dt_wbc;
#define P(x_, n_) ipow<n_>(x_)
  if (t < d - a.lock_time) {
    const double den1 = (2 * P((t - d), 2) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
    const double den2 = (2 * (P(t, 2) - 2 * t * d + P(d, 2)) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
#pragma unroll
    for (int ax = 0; ax < 2; ax++) {
      const double dd0 = ax ? ddy0 : ddx0, d0 = ax ? dy0 : dx0, p0 = ax ? y0 : x0, tg = tf[ax];
      double* A = ax ? f.ay : f.ax;
      A[0] = (dd0 * P(t, 2) - 2 * dd0 * t * d - 6 * d0 * t + dd0 * P(d, 2) + 6 * d0 * d + 12 * p0 - 12 * tg) / den1;
      A[1] = (30 * t * tg - 30 * t * p0 - 30 * d * p0 + 30 * d * tg - 2 * P(t, 3) * dd0 - 3 * P(d, 3) * dd0 +
              14 * P(t, 2) * d0 - 16 * P(d, 2) * d0 + 2 * t * d * d0 + 4 * t * P(d, 2) * dd0 + P(t, 2) * d * dd0) / den1;
      A[2] = (P(t, 4) * dd0 + 3 * P(d, 4) * dd0 - 8 * P(t, 3) * d0 + 12 * P(d, 3) * d0 + 20 * P(t, 2) * p0 -
              20 * P(t, 2) * tg + 20 * P(d, 2) * p0 - 20 * P(d, 2) * tg + 80 * t * d * p0 - 80 * t * d * tg +
              4 * P(t, 3) * d * dd0 + 28 * t * P(d, 2) * d0 - 32 * P(t, 2) * d * d0 - 8 * P(t, 2) * P(d, 2) * dd0) / den1;
      A[3] = -(P(d, 5) * dd0 + 4 * t * P(d, 4) * dd0 + 3 * P(t, 4) * d * dd0 + 36 * t * P(d, 3) * d0 -
               24 * P(t, 3) * d * d0 + 60 * t * P(d, 2) * p0 + 60 * P(t, 2) * d * p0 - 60 * t * P(d, 2) * tg -
               60 * P(t, 2) * d * tg - 8 * P(t, 2) * P(d, 3) * dd0 - 12 * P(t, 2) * P(d, 2) * d0) / den2;
      A[4] = -(2 * P(d, 5) * d0 - 2 * t * P(d, 5) * dd0 - 10 * t * P(d, 4) * d0 + P(t, 2) * P(d, 4) * dd0 +
               4 * P(t, 3) * P(d, 3) * dd0 - 3 * P(t, 4) * P(d, 2) * dd0 - 16 * P(t, 2) * P(d, 3) * d0 +
               24 * P(t, 3) * P(d, 2) * d0 - 60 * P(t, 2) * P(d, 2) * p0 + 60 * P(t, 2) * P(d, 2) * tg) / den1;
      A[5] = (2 * tg * P(t, 5) - dd0 * P(t, 4) * P(d, 3) - 10 * tg * P(t, 4) * d + 2 * dd0 * P(t, 3) * P(d, 4) +
              8 * d0 * P(t, 3) * P(d, 3) + 20 * tg * P(t, 3) * P(d, 2) - dd0 * P(t, 2) * P(d, 5) -
              10 * d0 * P(t, 2) * P(d, 4) - 20 * p0 * P(t, 2) * P(d, 3) + 2 * d0 * t * P(d, 5) +
              10 * p0 * t * P(d, 4) - 2 * p0 * P(d, 5)) / den2;
    }
    fttgt0 = tf[0];
    fttgt1 = tf[1];
    fttgt_dirty = true;
  }
  const double dz = (P((d / 2), 3) * P((d - d / 2), 3));
  const double Az0 = -a.max_height / dz, Az1 = (3 * d * a.max_height) / dz, Az2 = -(3 * P(d, 2) * a.max_height) / dz,
               Az3 = (P(d, 3) * a.max_height) / dz;
  const double ev = t + dt;
  if (t < 0.0 || t > d) {
    f.pos[0] = x0; f.pos[1] = y0;
    f.vel[0] = 0.0; f.vel[1] = 0.0;
    f.acc[0] = 0.0; f.acc[1] = 0.0;
  } else {
#pragma unroll
    for (int ax = 0; ax < 2; ax++) {
      const double* A = ax ? f.ay : f.ax;
      const double A0 = A[0], A1 = A[1], A2 = A[2], A3 = A[3], A4 = A[4], A5 = A[5];
      f.pos[ax] = A5 + A4 * ev + A3 * P(ev, 2) + A2 * P(ev, 3) + A1 * P(ev, 4) + A0 * P(ev, 5);
      f.vel[ax] = A4 + 2 * A3 * ev + 3 * A2 * P(ev, 2) + 4 * A1 * P(ev, 3) + 5 * A0 * P(ev, 4);
      f.acc[ax] = 2 * A3 + 3 * 2 * A2 * ev + 4 * 3 * A1 * P(ev, 2) + 5 * 4 * A0 * P(ev, 3);
    }
  }
  f.vel[2] = 3 * Az3 * P(ev, 2) + 4 * Az2 * P(ev, 3) + 5 * Az1 * P(ev, 4) + 6 * Az0 * P(ev, 5);
  f.acc[2] = 2 * 3 * Az3 * ev + 3 * 4 * Az2 * P(ev, 2) + 4 * 5 * Az1 * P(ev, 3) + 5 * 6 * Az0 * P(ev, 4);
  f.pos[2] = Az3 * P(ev, 3) + Az2 * P(ev, 4) + Az1 * P(ev, 5) + Az0 * P(ev, 6);
#undef P
}
__device__ __forceinline__ double quadmax_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true));
  v = fmax(v, o);
  lo = __double2loint(v); hi = __double2hiint(v);
  o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true));
  return fmax(v, o);
}
}  // namespace

__global__ __launch_bounds__(64) void control_pre_quad_kernel(ControllerArgs cu, PlannerArgs a, ControllerArgs cw, int with_wbc_inputs) {
  kernarg_warm<(2 * sizeof(ControllerArgs) + sizeof(PlannerArgs) + 4 + 63) / 64>();
  const int tix = blockIdx.x * 64 + threadIdx.x;
  const int b = tix >> 2, j = tix & 3;
  if (b >= a.B) return;  // whole quads leave together
  const int Ng = a.N_gait, k = a.k;
  const Lay L = make_layout(Ng);
  PS s;
  s.base = a.ps + b;
  s.stride = (size_t)a.B;
  const glue::CS cs = glue::state_of(cu, b);
  const double dtw = cu.dt_wbc;
#ifdef QRW_PROFILE_PRE
  long long pp_last = clock64();
  if (threadIdx.x == 0) atomicAdd(&qrw_pre_prof[0], 1ull);
#endif

  // ================= operands of all three pieces, ahead of the first store =================
  double jv[6], vf6[6], rpy_in[3], vf_mine[3], qf_mine[3];
  {
    const double* pj = cu.in0 + (size_t)b * 6;
    const double* pq = cu.in1 + (size_t)b * 19;
    const double* pv = cu.in2 + (size_t)b * 18;
    const double* pr = cu.in3 + (size_t)b * 3;
#pragma unroll
    for (int i = 0; i < 6; i++) { jv[i] = pj[i]; vf6[i] = pv[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { rpy_in[i] = pr[i]; vf_mine[i] = pv[6 + 3 * j + i]; qf_mine[i] = pq[7 + 3 * j + i]; }
  }
  const double qf2 = cu.in1[(size_t)b * 19 + 2];
  const double yaw0 = cs(glue::cYAW), qx0 = cs(glue::cQX), qy0 = cs(glue::cQY);
  Gm past, cur, des;
  gm_load(s, L.past, past);
  gm_load(s, L.cur, cur);
  gm_load(s, L.des, des);
  double newphase = s(L.newphase), remain = s(L.remain);
  double cf[3], fs1[3];
#pragma unroll
  for (int r = 0; r < 3; r++) { cf[r] = s(L.cf + r * 4 + j); fs1[r] = s(FSI(1, r, j)); }
  const double otgt_z = s(L.otgt + 8 + j);
  FootLane ft;
  ft.t0s = s(L.t0s + j); ft.tsw = s(L.tsw + j);
#pragma unroll
  for (int r = 0; r < 6; r++) { ft.ax[r] = s(L.ax + r * 4 + j); ft.ay[r] = s(L.ay + r * 4 + j); }
#pragma unroll
  for (int r = 0; r < 3; r++) { ft.pos[r] = s(L.pos + r * 4 + j); ft.vel[r] = s(L.vel + r * 4 + j); ft.acc[r] = s(L.acc + r * 4 + j); }
  const double nfeet_s = s(L.nfeet);
  double feet_s[4];
#pragma unroll
  for (int jj = 0; jj < 4; jj++) feet_s[jj] = s(L.feet + jj);
  const int code = a.code ? a.code[b] : a.code_scalar;
  // WBC target assembly: the previous commands of this foot, the PD references of this foot's joints, the MPC forces of this foot
  double pcmd[3] = {0, 0, 0}, vcmd[3] = {0, 0, 0}, qdes[3] = {0, 0, 0}, vdes[3] = {0, 0, 0}, xf_f[3] = {0, 0, 0};
  if (with_wbc_inputs) {
    const int N = cw.n_steps;
    const double* xf = cw.in0 + (size_t)b * 24 * N;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      pcmd[r] = cs(glue::cPCMD + r * 4 + j); vcmd[r] = cs(glue::cVCMD + r * 4 + j);
      qdes[r] = cs(glue::cQDES + 3 * j + r); vdes[r] = cs(glue::cVDES + 3 * j + r);
      xf_f[r] = xf[(12 + 3 * j + r) * N];
    }
  }

  PP(2);
  // ================= Controller.updateState (scripts/Controller.py:381-426) =================
  // Eight sine / cosine pairs of this block and of the footstep planner's head have arguments that are known here: each lane of
  // the quad evaluates two of them (its own argument, the same library routine: the same bits) and the quad shares the results --
  // two evaluations per lane instead of eight (round 4: the state update was 13 k of the kernel's 73 k clocks, nearly all trig)
  const double yaw = yaw0 + jv[5] * dtw;
  const double ang_a = (j == 0) ? yaw0 : (j == 1) ? rpy_in[0] / 2. : (j == 2) ? rpy_in[1] / 2. : yaw / 2.;
  const double ang_b = (j == 0) ? rpy_in[0] : (j == 1) ? rpy_in[1] : (j == 2) ? yaw : a.dt_wbc * jv[5];
  const double sin_a = sin(ang_a), cos_a = cos(ang_a), sin_b = sin(ang_b), cos_b = cos(ang_b);
  const double c0 = quad_bcast<0>(cos_a), s0 = quad_bcast<0>(sin_a);
  const double qx = qx0 + (c0 * jv[0] + -s0 * jv[1]) * dtw;
  const double qy = qy0 + (s0 * jv[0] + c0 * jv[1]) * dtw;
  double q7[7];
  q7[0] = qx; q7[1] = qy; q7[2] = qf2;
  {
    const double sr = quad_bcast<1>(sin_a), cr = quad_bcast<1>(cos_a), sp = quad_bcast<2>(sin_a), cp = quad_bcast<2>(cos_a),
                 sy = quad_bcast<3>(sin_a), cy = quad_bcast<3>(cos_a);
    q7[3] = sr * cp * cy - cr * sp * sy;
    q7[4] = cr * sp * cy + sr * cp * sy;
    q7[5] = cr * cp * sy - sr * sp * cy;
    q7[6] = cr * cp * cy + sr * sp * sy;
  }
  double hv[6];
  {
    const double cr = quad_bcast<0>(cos_b), sr = quad_bcast<0>(sin_b), cp = quad_bcast<1>(cos_b), sp = quad_bcast<1>(sin_b);
    const double R[9] = {cp, sp * sr, sp * cr, 0.0, cr, -sr, -sp, cp * sr, cp * cr};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      hv[r] = R[r * 3] * vf6[0] + R[r * 3 + 1] * vf6[1] + R[r * 3 + 2] * vf6[2];
      hv[3 + r] = R[r * 3] * vf6[3] + R[r * 3 + 1] * vf6[4] + R[r * 3 + 2] * vf6[5];
    }
  }
  const double cyw = quad_bcast<2>(cos_b), syw = quad_bcast<2>(sin_b);
  {
    double* q = cu.out0 + (size_t)b * 19;
    double* v = cu.out1 + (size_t)b * 18;
#pragma unroll
    for (int i = 0; i < 3; i++) { q[7 + 3 * j + i] = qf_mine[i]; v[6 + 3 * j + i] = vf_mine[i]; }
    if (j == 0) {
#pragma unroll
      for (int i = 0; i < 7; i++) q[i] = q7[i];
      cs(glue::cQX) = qx; cs(glue::cQY) = qy; cs(glue::cYAW) = yaw;
    } else if (j == 1) {
#pragma unroll
      for (int i = 0; i < 6; i++) { v[i] = vf6[i]; cs(glue::cVREF + i) = jv[i]; }
    } else if (j == 2) {
      double* o = cu.out2 + (size_t)b * 6;
#pragma unroll
      for (int i = 0; i < 6; i++) o[i] = hv[i];
      if (cu.out3) {
#pragma unroll
        for (int i = 0; i < 6; i++) cu.out3[(size_t)b * 6 + i] = jv[i];
      }
    } else if (cu.out4) {
      double* o = cu.out4 + (size_t)b * 12;
      o[0] = cyw; o[1] = -syw; o[2] = 0; o[3] = syw; o[4] = cyw; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = 1;
      o[9] = qx; o[10] = qy; o[11] = 0.0;
    }
  }
  const double* vr = jv;  // the planners' reference velocity is the joystick's (Controller.py:222-236 via updateState)
  double rpy[3];
  quat_to_rpy(q7 + 3, rpy);

  PP(1);
  // ================= Gait::updateGait (all four lanes hold the whole matrices as bit masks) =================
  if (j == 0) {
    s(L.isstatic) = (code == 4) ? 1.0 : 0.0;
    if (code == 4)
      for (int i = 0; i < 7; i++) s(L.qstatic + i) = q7[i];
  }
  gait_update(past, cur, des, a, k, code, newphase);
  if (j == 0 && k % a.k_mpc == 0) s(L.newphase) = newphase;

  PP(3);
  // ================= FootstepPlanner::updateFootsteps, this lane's foot =================
  double last_call = -1.0;  // order key of this lane's last getPhaseDuration call (table: 4 i + j, trajectory: 4 N_gait + j)
  const bool ct0 = gbit(cur, 0, j);
  if (a.refresh != 0 && newphase != 0.0 && ct0) {
#pragma unroll
    for (int r = 0; r < 3; r++) cf[r] = fs1[r];
  }
  {
    const double c = quad_bcast<3>(cos_b), sn = quad_bcast<3>(sin_b);  // of dt_wbc * v_ref(5), the rotation of the last time step
    const double dpx = a.dt_wbc * vr[0], dpy = a.dt_wbc * vr[1];
    if (ct0) {
      const double x = cf[0] - dpx, y = cf[1] - dpy;
      cf[0] = c * x + sn * y;
      cf[1] = -sn * x + c * y;
    }
  }
#pragma unroll
  for (int r = 0; r < 3; r++) s(L.cf + r * 4 + j) = cf[r];
  double row[3], tg[2] = {0.0, 0.0};
  bool found = false;
  double* fo = (a.fsteps) ? a.fsteps + (size_t)b * Ng * 12 : nullptr;
#pragma unroll
  for (int r = 0; r < 3; r++) row[r] = ct0 ? cf[r] : 0.0;
  const double w = vr[5];
  const double cross0 = hv[1] * vr[5] - hv[2] * vr[4], cross1 = hv[2] * vr[3] - hv[0] * vr[5];
  PP(4);
  // The table row by row (src/FootstepPlanner.cpp:76-152) is cheap -- a stance foot keeps its row, a swing foot has zeros -- except
  // at a TOUCH-DOWN (swing -> stance): yaw sine / cosine, the displacement integrals, getPhaseDuration, the Raibert terms.  A
  // foot touches down once or twice in the table, but at a different row on every lane, so a wavefront that walks the rows
  // together executes the expensive branch at nearly every row (round 4: 25.7 k of the kernel's 73 k clocks).  Here each lane
  // first finds ITS touch-down rows with mask arithmetic, the wavefront then computes "the k-th touch-down of every lane" together
  // (once or twice in all), and the row loop only selects and stores.  Same arithmetic per touch-down, same call order of
  // getPhaseDuration per foot (remainingTime_, `last_call`).  More than kTdMax touch-downs on some lane: the row-wise form.
  constexpr int kTdMax = 4;
  // this foot's shoulder by selects on the kernel arguments (scalar registers): indexed with the lane's foot number the array is
  // fetched from the kernel-argument segment in memory instead -- two dependent loads per touch-down, each waiting for every
  // store issued before it (stores count in vmcnt on gfx9): 8 of the table's 16.7 k clocks
  const double sh0 = (j == 0) ? a.shoulders[0] : (j == 1) ? a.shoulders[1] : (j == 2) ? a.shoulders[2] : a.shoulders[3];
  const double sh1 = (j == 0) ? a.shoulders[4] : (j == 1) ? a.shoulders[5] : (j == 2) ? a.shoulders[6] : a.shoulders[7];
  const unsigned long long colj = (j == 0) ? cur.c[0] : (j == 1) ? cur.c[1] : (j == 2) ? cur.c[2] : cur.c[3];
  const unsigned long long anyrow = gm_any(cur) | 1ull;                 // (row 0 is not tested by the reference's walk)
  const int zrow = (~anyrow == 0ull) ? 64 : (__ffsll((long long)~anyrow) - 1);  // first all-zero row >= 1: the walk stops there
  const unsigned long long livem = lowmask(zrow < Ng ? zrow : Ng);
  unsigned long long tdm = colj & ~(colj << 1) & livem & ~1ull;        // rows i >= 1 with gait(i-1, j) = 0, gait(i, j) = 1
  const bool rowwise = __any(__popcll(tdm) > kTdMax);
  double tdv[kTdMax][3];
#pragma unroll
  for (int q = 0; q < kTdMax; q++) tdv[q][0] = tdv[q][1] = tdv[q][2] = 0.0;
  if (!rowwise) {
    unsigned long long left = tdm;
    double dtc = a.dt_wbc * a.k_footsteps;  // dt_cum of the row before the touch-down: one addition per row, as the reference
    int irow = 1;
#pragma unroll
    for (int q = 0; q < kTdMax; q++) {
      if (!__any(left != 0ull)) break;
      const bool have = left != 0ull;
      const int i = have ? (__ffsll((long long)left) - 1) : irow;
      left &= left - 1ull;
      for (; irow < i; irow++) dtc = dtc + a.dt_mpc;  // (rows 1 .. i - 1 are live: each adds dt)
      if (have) {
        const double yawp = w * dtc;
        const double c = cos(yawp), sn = sin(yawp);
        double dxp, dyp;
        if (w != 0) {
          dxp = (hv[0] * sn + hv[1] * (c - 1.0)) / w;
          dyp = (hv[1] * sn - hv[0] * (c - 1.0)) / w;
        } else {
          dxp = hv[0] * dtc;
          dyp = hv[1] * dtc;
        }
        const double t_stance = phase_duration(past, cur, des, a, i, j, true, remain);
        last_call = (double)(4 * i + j);
        double nf[3];
        const double cr[3] = {cross0, cross1, 0.0};
#pragma unroll
        for (int r = 0; r < 3; r++) {
          double v = t_stance * 0.5 * hv[r];
          v += a.k_feedback * (hv[r] - vr[r]);
          v += 0.5 * sqrt(a.h_ref / a.g) * cr[r];
          nf[r] = v;
        }
        nf[0] = fmax(fmin(nf[0], a.L), -a.L);
        nf[1] = fmax(fmin(nf[1], a.L), -a.L);
        nf[0] += sh0;
        nf[1] += sh1;
        nf[2] = 0.0;
        tdv[q][0] = (c * nf[0] - sn * nf[1] + 0.0 * nf[2]) + dxp;
        tdv[q][1] = (sn * nf[0] + c * nf[1] + 0.0 * nf[2]) + dyp;
        tdv[q][2] = (0.0 * nf[0] + 0.0 * nf[1] + 1.0 * nf[2]) + 0.0;
      }
    }
    const unsigned long long stm = colj & livem;            // rows in stance (and live)
    const unsigned long long carrym = stm & (colj << 1);    // ... that continue a stance
    // running pointers (one 64-bit add per row instead of a 64-bit multiply-add per store), no branch in the loop body
    // On an iteration that does not solve (no fsteps output, xref_steps = 1: qrw_control_pre without the MPC's inputs) nobody reads
    // the table before the next solving iteration rewrites it -- except row 1, which updateNewContact takes at the next gait
    // change (src/FootstepPlanner.cpp:225-230): rows 0 and 1 of the state copy are kept current, the rest waits (the target
    // footsteps come from the registers).  54 of 60 stores fewer on nine iterations of ten.
    const bool full_table = (fo != nullptr) || (a.xref_steps == 0);
    const size_t row_stride = 12 * s.stride;
    double* p0 = &s(FSI(0, 0, j));
    double* p1 = &s(FSI(0, 1, j));
    double* p2 = &s(FSI(0, 2, j));
    double* pf = fo ? fo + 3 * j : nullptr;
    for (int i = 0; i < Ng; i++) {
      const bool td = i > 0 && ((tdm >> i) & 1ull) != 0ull, keep = i == 0 || ((carrym >> i) & 1ull) != 0ull;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        row[r] = td ? tdv[0][r] : (keep ? row[r] : 0.0);
        // the touch-down values are consumed in order: shift the next one up (static indices only: register arrays)
        tdv[0][r] = td ? tdv[1][r] : tdv[0][r];
        tdv[1][r] = td ? tdv[2][r] : tdv[1][r];
        tdv[2][r] = td ? tdv[3][r] : tdv[2][r];
      }
      if (full_table || i < 2) { *p0 = row[0]; *p1 = row[1]; *p2 = row[2]; }
      p0 += row_stride; p1 += row_stride; p2 += row_stride;
      if (pf) { pf[0] = row[0]; pf[1] = row[1]; pf[2] = row[2]; pf += 12; }
      const bool take = !found && (row[0] != 0.0 || i == Ng - 1);
      tg[0] = take ? row[0] : tg[0];
      tg[1] = take ? row[1] : tg[1];
      found = found || take;
    }
  } else {
  double dtc_prev = a.dt_wbc * a.k_footsteps;
  bool live = true;
  for (int i = 0; i < Ng; i++) {
    if (i > 0) {
      live = live && !rz(cur, i);
      const bool gp = gbit(cur, i - 1, j), gc = gbit(cur, i, j);
      double nrow[3];
#pragma unroll
      for (int r = 0; r < 3; r++) nrow[r] = (live && gp && gc) ? row[r] : 0.0;
      if (live) {
        if (!gp && gc) {
          const double yawp = w * dtc_prev;
          const double c = cos(yawp), sn = sin(yawp);
          double dxp, dyp;
          if (w != 0) {
            dxp = (hv[0] * sn + hv[1] * (c - 1.0)) / w;
            dyp = (hv[1] * sn - hv[0] * (c - 1.0)) / w;
          } else {
            dxp = hv[0] * dtc_prev;
            dyp = hv[1] * dtc_prev;
          }
          const double t_stance = phase_duration(past, cur, des, a, i, j, true, remain);
          last_call = (double)(4 * i + j);
          double nf[3];
          const double cr[3] = {cross0, cross1, 0.0};
#pragma unroll
          for (int r = 0; r < 3; r++) {
            double v = t_stance * 0.5 * hv[r];
            v += a.k_feedback * (hv[r] - vr[r]);
            v += 0.5 * sqrt(a.h_ref / a.g) * cr[r];
            nf[r] = v;
          }
          nf[0] = fmax(fmin(nf[0], a.L), -a.L);
          nf[1] = fmax(fmin(nf[1], a.L), -a.L);
          nf[0] += sh0;
          nf[1] += sh1;
          nf[2] = 0.0;
          nrow[0] = (c * nf[0] - sn * nf[1] + 0.0 * nf[2]) + dxp;
          nrow[1] = (sn * nf[0] + c * nf[1] + 0.0 * nf[2]) + dyp;
          nrow[2] = (0.0 * nf[0] + 0.0 * nf[1] + 1.0 * nf[2]) + 0.0;
        }
        dtc_prev = dtc_prev + a.dt_mpc;
      }
#pragma unroll
      for (int r = 0; r < 3; r++) row[r] = nrow[r];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      s(FSI(i, r, j)) = row[r];
      if (fo) fo[i * 12 + 3 * j + r] = row[r];
    }
    if (!found && (row[0] != 0.0 || i == Ng - 1)) {
      found = true;
      tg[0] = row[0];
      tg[1] = row[1];
    }
  }
  }
  PP(5);
  double otgt[3];
  {
    const double c = cos(rpy[2]), sn = sin(rpy[2]);
    if (j == 0) { s(L.rz) = c; s(L.rz + 1) = sn; }
    s(L.tgt + j) = tg[0];
    s(L.tgt + 4 + j) = tg[1];
    s(L.tgt + 8 + j) = 0.0;
    otgt[0] = (c * tg[0] - sn * tg[1]) + q7[0];
    otgt[1] = (sn * tg[0] + c * tg[1]) + q7[1];
    otgt[2] = otgt_z;
    s(L.otgt + j) = otgt[0];
    s(L.otgt + 4 + j) = otgt[1];
  }

  PP(6);
  // ================= FootTrajectoryGenerator::update, this lane's foot =================
  {
    unsigned swing = 0;
    bool run;
    if ((k % a.k_mpc) == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!gbit(cur, 0, i)) swing |= 1u << i;
      const int nf = __popc(swing);
      if (swing & (1u << j)) s(L.feet + __popc(swing & ((1u << j) - 1u))) = (double)j;
      if (j == 0) s(L.nfeet) = (double)nf;
      run = nf != 0;
      if (swing & (1u << j)) {
        const double tsw = phase_duration(past, cur, des, a, 0, j, false, remain);
        last_call = (double)(4 * Ng + j);
        ft.tsw = tsw;
        const double value = tsw - (remain * a.k_mpc - ((k + 1) % a.k_mpc)) * a.dt_wbc - a.dt_wbc;
        ft.t0s = fmax(0.0, value);
      }
    } else {
      const int nf = (int)nfeet_s;
      run = nf != 0;
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
        if (jj < nf) swing |= 1u << (int)feet_s[jj];
      if (swing & (1u << j)) ft.t0s = fmax(0.0, ft.t0s + a.dt_wbc);
    }
    double ftt0 = 0.0, ftt1 = 0.0;
    bool ftt_dirty = false;
    if (run && (swing & (1u << j))) update_foot_lane(ft, a, otgt, ftt0, ftt1, ftt_dirty);
    if (ftt_dirty) { s(L.fttgt + j) = ftt0; s(L.fttgt + 4 + j) = ftt1; }
    s(L.t0s + j) = ft.t0s; s(L.tsw + j) = ft.tsw;
#pragma unroll
    for (int r = 0; r < 6; r++) { s(L.ax + r * 4 + j) = ft.ax[r]; s(L.ay + r * 4 + j) = ft.ay[r]; }
#pragma unroll
    for (int r = 0; r < 3; r++) { s(L.pos + r * 4 + j) = ft.pos[r]; s(L.vel + r * 4 + j) = ft.vel[r]; s(L.acc + r * 4 + j) = ft.acc[r]; }
    if (a.feet_pva) {
#pragma unroll
      for (int r = 0; r < 3; r++) {
        a.feet_pva[(size_t)b * 36 + r * 4 + j] = ft.pos[r];
        a.feet_pva[(size_t)b * 36 + 12 + r * 4 + j] = ft.vel[r];
        a.feet_pva[(size_t)b * 36 + 24 + r * 4 + j] = ft.acc[r];
      }
    }
  }
  // remainingTime_ is what the LAST getPhaseDuration call of the sequential program left (src/Gait.cpp:141-182)
  {
    const double mx = quadmax_d(last_call);
    if (last_call >= 0.0 && last_call == mx) s(L.remain) = remain;
  }

  PP(7);
  // ================= StatePlanner::computeReferenceStates: the horizon steps dealt to the four lanes =================
  double xr1_6 = 0.0, xr1_7 = 0.0;  // rows 6, 7 of column 1 (the WBC target assembly reads them back)
  if (a.xref) {
    double* X = a.xref + (size_t)b * 12 * (a.n_steps + 1);
    const int n = a.n_steps, ld = n + 1;
    {  // column 0: three rows per lane
      const double col0[12] = {0.0, 0.0, q7[2], rpy[0], rpy[1], 0.0, hv[0], hv[1], hv[2], hv[3], hv[4], hv[5]};
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int e = 3 * j + r;
        double v = col0[0];
#pragma unroll
        for (int q = 1; q < 12; q++) v = (e == q) ? col0[q] : v;
        X[e * ld] = v;
      }
    }
    const double T_mpc = a.T_mpc;
    const int n_out = (a.xref_steps > 0 && a.xref_steps < n) ? a.xref_steps : n;
    for (int i = j; i < n_out; i += 4) {
      const double dtv = (n == 1 || i == n - 1) ? T_mpc : a.dt_mpc + i * ((T_mpc - a.dt_mpc) / (n - 1));
      const double yw = vr[5] * dtv;
      const double sy = sin(yw), cy = cos(yw);
      double x, y;
      if (vr[5] != 0) {
        x = (vr[0] * sy + vr[1] * (cy - 1.0)) / vr[5];
        y = (vr[1] * sy - vr[0] * (cy - 1.0)) / vr[5];
      } else {
        x = vr[0] * dtv;
        y = vr[1] * dtv;
      }
      const double v6 = vr[0] * cy - vr[1] * sy, v7 = vr[0] * sy + vr[1] * cy;
      X[0 * ld + 1 + i] = x + 0.0;
      X[1 * ld + 1 + i] = y + 0.0;
      X[2 * ld + 1 + i] = a.h_ref + a.z_average;
      X[3 * ld + 1 + i] = 0.0;
      X[4 * ld + 1 + i] = 0.0;
      X[5 * ld + 1 + i] = yw;
      X[6 * ld + 1 + i] = v6;
      X[7 * ld + 1 + i] = v7;
      X[8 * ld + 1 + i] = 0.0; X[9 * ld + 1 + i] = 0.0; X[10 * ld + 1 + i] = 0.0;
      X[11 * ld + 1 + i] = vr[5];
      if (i == 0) { xr1_6 = v6; xr1_7 = v7; }
    }
  }

  PP(8);
  // ================= remaining planner outputs / state =================
  if (a.gait) {
    double* o = a.gait + (size_t)b * Ng * 4;
    for (int i = 0; i < Ng; i++) o[i * 4 + j] = gbit(cur, i, j) ? 1.0 : 0.0;
  }
  if (a.contacts) a.contacts[(size_t)b * 4 + j] = gbit(cur, 0, j) ? 1.0 : 0.0;
  if (a.target) {
#pragma unroll
    for (int r = 0; r < 3; r++) a.target[(size_t)b * 12 + r * 4 + j] = otgt[r];
  }
  {
    const unsigned long long pc = (j == 0) ? past.c[0] : (j == 1) ? past.c[1] : (j == 2) ? past.c[2] : past.c[3];
    const unsigned long long cc = (j == 0) ? cur.c[0] : (j == 1) ? cur.c[1] : (j == 2) ? cur.c[2] : cur.c[3];
    const unsigned long long dc = (j == 0) ? des.c[0] : (j == 1) ? des.c[1] : (j == 2) ? des.c[2] : des.c[3];
    s(L.past + j) = __longlong_as_double((long long)pc);
    s(L.cur + j) = __longlong_as_double((long long)cc);
    s(L.des + j) = __longlong_as_double((long long)dc);
  }

  PP(9);
  // ================= WBC target assembly (scripts/Controller.py:258-296), this lane's foot =================
  if (with_wbc_inputs) {
    const double h_ref = cw.h_ref;
    double* xw = cw.out0 ? cw.out0 + (size_t)b * 24 : nullptr;
    if (xw) {
#pragma unroll
      for (int r = 0; r < 3; r++) xw[12 + 3 * j + r] = xf_f[r];
      if (j == 0) {  // lane 0 computed horizon step 1 of xref (rows 6..11 of column 1: v6, v7, 0, 0, 0, w_ref)
        xw[0] = dtw * xr1_6;
        xw[1] = dtw * xr1_7;
        xw[2] = h_ref; xw[3] = 0.0; xw[4] = 0.0;
        xw[5] = dtw * vr[5];
        xw[6] = xr1_6; xw[7] = xr1_7; xw[8] = 0.0; xw[9] = 0.0; xw[10] = 0.0; xw[11] = vr[5];
      }
    }
    double* qw = cw.out1 + (size_t)b * 19;
    double* bv = cw.out2 + (size_t)b * 18;
#pragma unroll
    for (int r = 0; r < 3; r++) { qw[7 + 3 * j + r] = qdes[r]; bv[6 + 3 * j + r] = vdes[r]; }
    if (j == 0) {
#pragma unroll
      for (int i = 0; i < 7; i++) qw[i] = (i == 2) ? h_ref : (i == 6) ? 1.0 : 0.0;
    } else if (j == 1) {
#pragma unroll
      for (int i = 0; i < 6; i++) bv[i] = jv[i];
    }
    if (cw.out3) {
#pragma unroll
      for (int r = 0; r < 3; r++) cw.out3[(size_t)b * 12 + 3 * j + r] = xf_f[r];
    }
    const double wv[3] = {jv[3], jv[4], jv[5]};
    const double vl[3] = {jv[0], jv[1], jv[2]};
    double* fc = cw.out4 + (size_t)b * 12;
    const size_t pl = (size_t)cw.B * 12;
    double wxp[3], wxwxp[3], wxv[3];
    glue::cross3(wv, pcmd, wxp);
    glue::cross3(wv, wxp, wxwxp);
    glue::cross3(wv, vcmd, wxv);
    const double ra[3] = {cyw * ft.acc[0] + syw * ft.acc[1], -syw * ft.acc[0] + cyw * ft.acc[1], ft.acc[2]};
    const double rv[3] = {cyw * ft.vel[0] + syw * ft.vel[1], -syw * ft.vel[0] + cyw * ft.vel[1], ft.vel[2]};
    const double dp[3] = {ft.pos[0] - 0.0 - qx, ft.pos[1] - 0.0 - qy, ft.pos[2] - h_ref - 0.0};
    const double rp[3] = {cyw * dp[0] + syw * dp[1], -syw * dp[0] + cyw * dp[1], dp[2]};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double an = ra[r] - wxwxp[r] - 2 * wxv[r];
      const double vn = (rv[r] - vl[r]) - wxp[r];
      fc[2 * pl + r * 4 + j] = an;
      fc[pl + r * 4 + j] = vn;
      fc[r * 4 + j] = rp[r];
      cs(glue::cVCMD + r * 4 + j) = vn;
      cs(glue::cPCMD + r * 4 + j) = rp[r];
    }
  }
  PP(10);
}

#undef FSI

#ifdef QRW_PROFILE_PRE
static void pre_prof_dump() {  // diagnostic build only: synchronises the device
  unsigned long long h[16];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(qrw_pre_prof), sizeof(h)) != hipSuccess || h[0] == 0) return;
  const char* nm[11] = {"", "update_state", "planner loads", "gait update", "footsteps head", "footsteps table", "targets",
                        "foot trajectory", "state_compute(xref)", "outputs + state stores", "wbc_inputs"};
  fprintf(stderr, "control_pre phases (clk per wavefront, %llu wavefronts):", h[0]);
  for (int k = 1; k <= 10; k++) fprintf(stderr, "  %s %.0f", nm[k], (double)h[k] / (double)h[0]);
  fprintf(stderr, "\n");
}
#endif

int planner_state_items(int N_gait) { return make_layout(N_gait).total; }

int planner_item_offset(int N_gait, int which) {
  const Lay L = make_layout(N_gait);
  switch (which) {
    case 0: return L.past;
    case 1: return L.cur;
    case 2: return L.des;
    case 3: return L.newphase;
    case 4: return L.isstatic;
    case 5: return L.remain;
    case 6: return L.nfeet;
    case 7: return L.tgt;
    case 8: return L.otgt;
    case 9: return L.pos;
    case 10: return L.vel;
    case 11: return L.acc;
    case 12: return L.t0s;
    case 13: return L.tsw;
    case 14: return L.fs;
    case 15: return L.cf;
    case 16: return L.qstatic;
    case 17: return L.fttgt;
    case 18: return L.rz;
    default: return -1;
  }
}

int control_pre_launch(const ControllerArgs& cu, const PlannerArgs& p, const ControllerArgs& cw, int with_wbc_inputs,
                       hipStream_t stream) {
#ifdef QRW_PROFILE_PRE
  static int calls = 0;
  if (++calls % 50 == 0) pre_prof_dump();
#endif
  // one quad per instance (control_pre_quad_kernel); QRW_PRE_QUAD=0 selects the one-thread-per-instance form (A/B timing).
  // The quad form covers what qrw_control_pre asks for: all four planners on every call, no external trajectory targets.
  const char* qe = getenv("QRW_PRE_QUAD");  // read per call: the A/B scripts flip it between two handles of one process
  const bool quad = !(qe && atoi(qe) == 0);
  const int want = kPlanGait | kPlanFootsteps | kPlanTraj | kPlanState;
  if (quad && p.mode == want && !p.target_in && p.q_ld == 19 && p.q7 == cu.out0 && p.hv == cu.out2 && p.vref == cu.out3 &&
      (!with_wbc_inputs || (cw.in1 == p.xref && cw.in2 == p.feet_pva && cw.in3 == cu.out1)) && p.N_gait <= 64)
    hipLaunchKernelGGL(control_pre_quad_kernel, dim3((4 * p.B + 63) / 64), dim3(64), 0, stream, cu, p, cw, with_wbc_inputs);
  else
    hipLaunchKernelGGL(control_pre_kernel, dim3((p.B + 63) / 64), dim3(64), 0, stream, cu, p, cw, with_wbc_inputs);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int planner_launch(const PlannerArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(planner_kernel, dim3((a.B + 63) / 64), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace qrw
