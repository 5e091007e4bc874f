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
// control flow; they cost ~1 % of the MPC solve).  Persistent planner state is item-major, ps[item][instance],
// so the threads of a wavefront read and write consecutive addresses.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "qrw_kernels.h"

namespace qrw {

namespace {

struct PS {  // accessor of one instance's planner state (item-major)
  double* base;
  size_t stride;  // = batch
  __device__ __forceinline__ double& operator()(int item) const { return base[(size_t)item * stride]; }
};

struct Lay {  // item offsets for a given N_gait
  int past, cur, des, cf, fs, tgt, otgt, t0s, tsw, ax, ay, pos, vel, acc, fttgt, feet, nfeet, newphase, isstatic, remain,
      qstatic, total;
};
__host__ __device__ inline Lay make_layout(int Ng) {
  Lay L;
  int o = 0;
  L.past = o; o += Ng * 4;
  L.cur = o; o += Ng * 4;
  L.des = o; o += Ng * 4;
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
  L.total = o;
  return L;
}

__device__ __forceinline__ bool row_zero(const PS& s, int m, int i) {
  return s(m + i * 4) == 0.0 && s(m + i * 4 + 1) == 0.0 && s(m + i * 4 + 2) == 0.0 && s(m + i * 4 + 3) == 0.0;
}
__device__ __forceinline__ void row_swap(const PS& s, int m, int a, int b) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double t = s(m + a * 4 + j);
    s(m + a * 4 + j) = s(m + b * 4 + j);
    s(m + b * 4 + j) = t;
  }
}
__device__ __forceinline__ void rows_fill(const PS& s, int m, int r0, int n, double a, double b, double c, double d) {
  for (int r = r0; r < r0 + n; r++) { s(m + r * 4) = a; s(m + r * 4 + 1) = b; s(m + r * 4 + 2) = c; s(m + r * 4 + 3) = d; }
}

// desired gait of one period (src/Gait.cpp:38-108); code 5 = walk (exists in the reference but is not reachable there)
__device__ void create_desired(const PS& s, const Lay& L, const PlannerArgs& a, int code) {
  for (int e = 0; e < a.N_gait * 4; e++) s(L.des + e) = 0.0;
  const int Nh = (int)lround(0.5 * a.T_gait / a.dt_mpc);
  if (code == 1) { rows_fill(s, L.des, 0, Nh, 1, 0, 1, 0); rows_fill(s, L.des, Nh, Nh, 0, 1, 0, 1); }
  else if (code == 2) { rows_fill(s, L.des, 0, Nh, 1, 1, 0, 0); rows_fill(s, L.des, Nh, Nh, 0, 0, 1, 1); }
  else if (code == 3) { rows_fill(s, L.des, 0, Nh, 1, 0, 0, 1); rows_fill(s, L.des, Nh, Nh, 0, 1, 1, 0); }
  else if (code == 4) { rows_fill(s, L.des, 0, (int)lround(a.T_gait / a.dt_mpc), 1, 1, 1, 1); }
  else if (code == 5) {
    const int Nq = (int)lround(0.25 * a.T_gait / a.dt_mpc);
    rows_fill(s, L.des, 0, Nq, 0, 1, 1, 1); rows_fill(s, L.des, Nq, Nq, 1, 0, 1, 1);
    rows_fill(s, L.des, 2 * Nq, Nq, 1, 1, 0, 1); rows_fill(s, L.des, 3 * Nq, Nq, 1, 1, 1, 0);
  }
}

// Gait::initialize (src/Gait.cpp:19-36) + create_gait_f (:110-139)
__device__ void gait_init(const PS& s, const Lay& L, const PlannerArgs& a) {
  for (int e = 0; e < a.N_gait * 4; e++) { s(L.past + e) = 0.0; s(L.cur + e) = 0.0; }
  create_desired(s, L, a, 3);
  int i = 0;
  for (int j = 0; j < a.n_steps; j++) {
#pragma unroll
    for (int c = 0; c < 4; c++) s(L.cur + j * 4 + c) = s(L.des + i * 4 + c);
    i++;
    if (row_zero(s, L.des, i)) i = 0;
  }
  int index = 1;
  while (!row_zero(s, L.des, index)) index++;
  for (int k = 0; k < i; k++)
    for (int m = 0; m < index - 1; m++) row_swap(s, L.des, m, m + 1);
  s(L.newphase) = 0.0; s(L.isstatic) = 0.0; s(L.remain) = 0.0; s(L.nfeet) = 0.0;
}

// Gait::getPhaseDuration (src/Gait.cpp:141-182); also leaves remainingTime_
__device__ double phase_duration(const PS& s, const Lay& L, const PlannerArgs& a, int i, int j, double value) {
  double t_phase = 1;
  int b = i;
  while (!row_zero(s, L.cur, i + 1) && s(L.cur + (i + 1) * 4 + j) == value) { i++; t_phase++; }
  if (row_zero(s, L.cur, i + 1)) {
    int k = 0;
    while (!row_zero(s, L.des, k) && s(L.des + k * 4 + j) == value) { k++; t_phase++; }
  }
  s(L.remain) = t_phase;
  while (b > 0 && s(L.cur + (b - 1) * 4 + j) == value) { b--; t_phase++; }
  if (b == 0) {
    while (!row_zero(s, L.past, b) && s(L.past + b * 4 + j) == value) { b++; t_phase++; }
  }
  return t_phase * a.dt_mpc;
}

// Gait::updateGait = changeGait + rollGait (src/Gait.cpp:184-260)
__device__ void gait_update(const PS& s, const Lay& L, const PlannerArgs& a, int k, const double* q7, int code) {
  s(L.isstatic) = 0.0;
  if (code >= 1 && code <= 5) create_desired(s, L, a, code);
  if (code == 4) {
    for (int i = 0; i < 7; i++) s(L.qstatic + i) = q7[i];
    s(L.isstatic) = 1.0;
  }
  if (k % a.k_mpc != 0) return;
  for (int m = a.n_steps; m > 0; m--) row_swap(s, L.past, m, m - 1);
  bool differ = false;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    s(L.past + c) = s(L.cur + c);
    differ = differ || (s(L.cur + c) != s(L.cur + 4 + c));
  }
  s(L.newphase) = differ ? 1.0 : 0.0;
  int index = 1;
  while (!row_zero(s, L.cur, index)) { row_swap(s, L.cur, index - 1, index); index++; }
#pragma unroll
  for (int c = 0; c < 4; c++) s(L.cur + (index - 1) * 4 + c) = s(L.des + c);
  index = 1;
  while (!row_zero(s, L.des, index)) { row_swap(s, L.des, index - 1, index); index++; }
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
__device__ void state_compute(const PlannerArgs& a, const double* q7, const double* v6, const double* vref6,
                              double z_average, double* X) {
  const int n = a.n_steps, ld = n + 1;
  double rpy[3];
  quat_to_rpy(q7 + 3, rpy);
  X[0 * ld] = 0.0; X[1 * ld] = 0.0; X[2 * ld] = q7[2];
  X[3 * ld] = rpy[0]; X[4 * ld] = rpy[1]; X[5 * ld] = 0.0;
  for (int i = 0; i < 3; i++) { X[(6 + i) * ld] = v6[i]; X[(9 + i) * ld] = v6[3 + i]; }
  const double T_mpc = a.T_mpc;
  for (int i = 0; i < n; i++) {
    const double dtv = (n == 1 || i == n - 1) ? T_mpc : a.dt_mpc + i * ((T_mpc - a.dt_mpc) / (n - 1));  // LinSpaced
    double x, y;
    if (vref6[5] != 0) {
      x = (vref6[0] * sin(vref6[5] * dtv) + vref6[1] * (cos(vref6[5] * dtv) - 1.0)) / vref6[5];
      y = (vref6[1] * sin(vref6[5] * dtv) - vref6[0] * (cos(vref6[5] * dtv) - 1.0)) / vref6[5];
    } else {
      x = vref6[0] * dtv;
      y = vref6[1] * dtv;
    }
    X[0 * ld + 1 + i] = x + X[0 * ld];
    X[1 * ld + 1 + i] = y + X[1 * ld];
    X[2 * ld + 1 + i] = a.h_ref + z_average;
    X[3 * ld + 1 + i] = 0.0;
    X[4 * ld + 1 + i] = 0.0;
    const double yaw = vref6[5] * dtv;
    X[5 * ld + 1 + i] = yaw;
    X[6 * ld + 1 + i] = vref6[0] * cos(yaw) - vref6[1] * sin(yaw);
    X[7 * ld + 1 + i] = vref6[0] * sin(yaw) + vref6[1] * cos(yaw);
    X[8 * ld + 1 + i] = 0.0; X[9 * ld + 1 + i] = 0.0; X[10 * ld + 1 + i] = 0.0;
    X[11 * ld + 1 + i] = vref6[5];
  }
}

#define FSI(i, r, c) (L.fs + ((i)*3 + (r)) * 4 + (c))

// FootstepPlanner::updateFootsteps (src/FootstepPlanner.cpp:51-74) with computeTargetFootstep (:204-221),
// computeFootsteps (:76-156), computeNextFootstep (:158-186), updateTargetFootsteps (:188-202), updateNewContact (:223-232)
__device__ void footsteps_update(const PS& s, const Lay& L, const PlannerArgs& a, bool refresh, int k, const double* q7,
                                 const double* b_v, const double* b_vref) {
  const int Ng = a.N_gait;
  if (refresh && s(L.newphase) != 0.0)
    for (int i = 0; i < 4; i++)
      if (s(L.cur + i) == 1.0)
        for (int r = 0; r < 3; r++) s(L.cf + r * 4 + i) = s(FSI(1, r, i));
  {
    const double ry = a.dt_wbc * b_vref[5];
    const double c = cos(ry), sn = sin(ry);
    const double dpx = a.dt_wbc * b_vref[0], dpy = a.dt_wbc * b_vref[1];
    for (int j = 0; j < 4; j++)
      if (s(L.cur + j) == 1.0) {
        const double x = s(L.cf + j) - dpx, y = s(L.cf + 4 + j) - dpy;
        s(L.cf + j) = c * x + sn * y;
        s(L.cf + 4 + j) = -sn * x + c * y;
      }
  }
  for (int e = 0; e < Ng * 12; e++) s(L.fs + e) = 0.0;
  for (int j = 0; j < 4; j++)
    if (s(L.cur + j) == 1.0)
      for (int r = 0; r < 3; r++) s(FSI(0, r, j)) = s(L.cf + r * 4 + j);
  // running values of dt_cum / yaw / dx / dy for row i-1 (only consumed when a foot lands at row i)
  const double w = b_vref[5];
  double dtc_prev = a.dt_wbc * k;
  const double cross0 = b_v[1] * b_vref[5] - b_v[2] * b_vref[4], cross1 = b_v[2] * b_vref[3] - b_v[0] * b_vref[5];
  int i = 1;
  while (!row_zero(s, L.cur, i)) {
    for (int j = 0; j < 4; j++) {
      const double gp = s(L.cur + (i - 1) * 4 + j), gc = s(L.cur + i * 4 + j);
      if (gp * gc > 0) {
        for (int r = 0; r < 3; r++) s(FSI(i, r, j)) = s(FSI(i - 1, r, j));
      }
    }
    for (int j = 0; j < 4; j++) {
      const double gp = s(L.cur + (i - 1) * 4 + j), gc = s(L.cur + i * 4 + j);
      if ((1 - gp) * gc > 0) {
        double dxp, dyp;
        if (w != 0) {
          dxp = (b_v[0] * sin(w * dtc_prev) + b_v[1] * (cos(w * dtc_prev) - 1.0)) / w;
          dyp = (b_v[1] * sin(w * dtc_prev) - b_v[0] * (cos(w * dtc_prev) - 1.0)) / w;
        } else {
          dxp = b_v[0] * dtc_prev;
          dyp = b_v[1] * dtc_prev;
        }
        const double t_stance = phase_duration(s, L, a, i, j, 1.0);
        double nf[3];
        const double cr[3] = {cross0, cross1, 0.0};
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
        const double yawp = w * dtc_prev;
        const double c = cos(yawp), sn = sin(yawp);
        s(FSI(i, 0, j)) = (c * nf[0] - sn * nf[1] + 0.0 * nf[2]) + dxp;
        s(FSI(i, 1, j)) = (sn * nf[0] + c * nf[1] + 0.0 * nf[2]) + dyp;
        s(FSI(i, 2, j)) = (0.0 * nf[0] + 0.0 * nf[1] + 1.0 * nf[2]) + 0.0;
      }
    }
    dtc_prev = dtc_prev + a.dt_mpc;  // dt_cum(i) = dt_cum(i-1) + dt for a non-zero row i
    i++;
  }
  for (int f = 0; f < 4; f++) {
    int index = 0;
    while (index < Ng - 1 && s(FSI(index, 0, f)) == 0.0) index++;
    s(L.tgt + f) = s(FSI(index, 0, f));
    s(L.tgt + 4 + f) = s(FSI(index, 1, f));
    s(L.tgt + 8 + f) = 0.0;
  }
  double rpy[3];
  quat_to_rpy(q7 + 3, rpy);
  const double c = cos(rpy[2]), sn = sin(rpy[2]);
  for (int f = 0; f < 4; f++) {
    const double x = s(L.tgt + f), y = s(L.tgt + 4 + f);
    s(L.otgt + f) = (c * x - sn * y) + q7[0];
    s(L.otgt + 4 + f) = (sn * x + c * y) + q7[1];
  }
}

// FootTrajectoryGenerator::updateFootPosition (src/FootTrajectoryGenerator.cpp:41-106)
__device__ void update_foot_position(const PS& s, const Lay& L, const PlannerArgs& a, int j, const double tf[3]) {
  const double ddx0 = s(L.acc + j), ddy0 = s(L.acc + 4 + j);
  const double dx0 = s(L.vel + j), dy0 = s(L.vel + 4 + j);
  const double x0 = s(L.pos + j), y0 = s(L.pos + 4 + j);
  const double t = s(L.t0s + j), d = s(L.tsw + j), dt = a.dt_wbc;
#define P(x_, n_) pow((x_), (n_))
  if (t < d - a.lock_time) {
    const double den1 = (2 * P((t - d), 2) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
    const double den2 = (2 * (P(t, 2) - 2 * t * d + P(d, 2)) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
    for (int ax = 0; ax < 2; ax++) {
      const double dd0 = ax ? ddy0 : ddx0, d0 = ax ? dy0 : dx0, p0 = ax ? y0 : x0, tg = tf[ax];
      const int A = ax ? L.ay : L.ax;
      s(A + 0 * 4 + j) = (dd0 * P(t, 2) - 2 * dd0 * t * d - 6 * d0 * t + dd0 * P(d, 2) + 6 * d0 * d + 12 * p0 - 12 * tg) / den1;
      s(A + 1 * 4 + j) = (30 * t * tg - 30 * t * p0 - 30 * d * p0 + 30 * d * tg - 2 * P(t, 3) * dd0 - 3 * P(d, 3) * dd0 +
                          14 * P(t, 2) * d0 - 16 * P(d, 2) * d0 + 2 * t * d * d0 + 4 * t * P(d, 2) * dd0 + P(t, 2) * d * dd0) / den1;
      s(A + 2 * 4 + j) = (P(t, 4) * dd0 + 3 * P(d, 4) * dd0 - 8 * P(t, 3) * d0 + 12 * P(d, 3) * d0 + 20 * P(t, 2) * p0 -
                          20 * P(t, 2) * tg + 20 * P(d, 2) * p0 - 20 * P(d, 2) * tg + 80 * t * d * p0 - 80 * t * d * tg +
                          4 * P(t, 3) * d * dd0 + 28 * t * P(d, 2) * d0 - 32 * P(t, 2) * d * d0 - 8 * P(t, 2) * P(d, 2) * dd0) / den1;
      s(A + 3 * 4 + j) = -(P(d, 5) * dd0 + 4 * t * P(d, 4) * dd0 + 3 * P(t, 4) * d * dd0 + 36 * t * P(d, 3) * d0 -
                           24 * P(t, 3) * d * d0 + 60 * t * P(d, 2) * p0 + 60 * P(t, 2) * d * p0 - 60 * t * P(d, 2) * tg -
                           60 * P(t, 2) * d * tg - 8 * P(t, 2) * P(d, 3) * dd0 - 12 * P(t, 2) * P(d, 2) * d0) / den2;
      s(A + 4 * 4 + j) = -(2 * P(d, 5) * d0 - 2 * t * P(d, 5) * dd0 - 10 * t * P(d, 4) * d0 + P(t, 2) * P(d, 4) * dd0 +
                           4 * P(t, 3) * P(d, 3) * dd0 - 3 * P(t, 4) * P(d, 2) * dd0 - 16 * P(t, 2) * P(d, 3) * d0 +
                           24 * P(t, 3) * P(d, 2) * d0 - 60 * P(t, 2) * P(d, 2) * p0 + 60 * P(t, 2) * P(d, 2) * tg) / den1;
      s(A + 5 * 4 + j) = (2 * tg * P(t, 5) - dd0 * P(t, 4) * P(d, 3) - 10 * tg * P(t, 4) * d + 2 * dd0 * P(t, 3) * P(d, 4) +
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
    s(L.pos + j) = x0; s(L.pos + 4 + j) = y0;
    s(L.vel + j) = 0.0; s(L.vel + 4 + j) = 0.0;
    s(L.acc + j) = 0.0; s(L.acc + 4 + j) = 0.0;
  } else {
    for (int ax = 0; ax < 2; ax++) {
      const int A = ax ? L.ay : L.ax;
      const double A0 = s(A + j), A1 = s(A + 4 + j), A2 = s(A + 8 + j), A3 = s(A + 12 + j), A4 = s(A + 16 + j), A5 = s(A + 20 + j);
      s(L.pos + ax * 4 + j) = A5 + A4 * ev + A3 * P(ev, 2) + A2 * P(ev, 3) + A1 * P(ev, 4) + A0 * P(ev, 5);
      s(L.vel + ax * 4 + j) = A4 + 2 * A3 * ev + 3 * A2 * P(ev, 2) + 4 * A1 * P(ev, 3) + 5 * A0 * P(ev, 4);
      s(L.acc + ax * 4 + j) = 2 * A3 + 3 * 2 * A2 * ev + 4 * 3 * A1 * P(ev, 2) + 5 * 4 * A0 * P(ev, 3);
    }
  }
  s(L.vel + 8 + j) = 3 * Az3 * P(ev, 2) + 4 * Az2 * P(ev, 3) + 5 * Az1 * P(ev, 4) + 6 * Az0 * P(ev, 5);
  s(L.acc + 8 + j) = 2 * 3 * Az3 * ev + 3 * 4 * Az2 * P(ev, 2) + 4 * 5 * Az1 * P(ev, 3) + 5 * 6 * Az0 * P(ev, 4);
  s(L.pos + 8 + j) = Az3 * P(ev, 3) + Az2 * P(ev, 4) + Az1 * P(ev, 5) + Az0 * P(ev, 6);
#undef P
}

// FootTrajectoryGenerator::update (src/FootTrajectoryGenerator.cpp:108-151)
__device__ void traj_update(const PS& s, const Lay& L, const PlannerArgs& a, int k, const double tgt[12]) {
  if ((k % a.k_mpc) == 0) {
    int nf = 0;
    for (int i = 0; i < 4; i++)
      if (s(L.cur + i) == 0.0) { s(L.feet + nf) = (double)i; nf++; }
    s(L.nfeet) = (double)nf;
    if (nf == 0) return;
    for (int jj = 0; jj < nf; jj++) {
      const int i = (int)s(L.feet + jj);
      const double tsw = phase_duration(s, L, a, 0, i, 0.0);
      s(L.tsw + i) = tsw;
      const double value = tsw - (s(L.remain) * a.k_mpc - ((k + 1) % a.k_mpc)) * a.dt_wbc - a.dt_wbc;
      s(L.t0s + i) = fmax(0.0, value);
    }
  } else {
    const int nf = (int)s(L.nfeet);
    if (nf == 0) return;
    for (int jj = 0; jj < nf; jj++) {
      const int i = (int)s(L.feet + jj);
      s(L.t0s + i) = fmax(0.0, s(L.t0s + i) + a.dt_wbc);
    }
  }
  const int nf = (int)s(L.nfeet);
  for (int jj = 0; jj < nf; jj++) {
    const int i = (int)s(L.feet + jj);
    const double tf[3] = {tgt[i], tgt[4 + i], tgt[8 + i]};
    update_foot_position(s, L, a, i, tf);
  }
}

}  // namespace

__global__ __launch_bounds__(64) void planner_kernel(PlannerArgs a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const Lay L = make_layout(a.N_gait);
  PS s;
  s.base = a.ps + b;
  s.stride = (size_t)a.B;

  if (a.mode & kPlanInit) {
    gait_init(s, L, a);
    for (int e = 0; e < 12; e++) {
      s(L.cf + e) = a.shoulders[e]; s(L.tgt + e) = a.shoulders[e]; s(L.otgt + e) = a.shoulders[e];
      s(L.fttgt + e) = a.init_target[e]; s(L.pos + e) = a.init_pos[e]; s(L.vel + e) = 0.0; s(L.acc + e) = 0.0;
    }
    for (int e = 0; e < a.N_gait * 12; e++) s(L.fs + e) = 0.0;
    for (int e = 0; e < 24; e++) { s(L.ax + e) = 0.0; s(L.ay + e) = 0.0; }
    for (int e = 0; e < 4; e++) { s(L.t0s + e) = 0.0; s(L.tsw + e) = 0.0; s(L.feet + e) = 0.0; }
    for (int e = 0; e < 7; e++) s(L.qstatic + e) = 0.0;
  }
  double q7[7] = {0, 0, 0, 0, 0, 0, 1}, hv[6] = {0, 0, 0, 0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
  if (a.q7) for (int i = 0; i < 7; i++) q7[i] = a.q7[(size_t)b * 7 + i];
  if (a.hv) for (int i = 0; i < 6; i++) hv[i] = a.hv[(size_t)b * 6 + i];
  if (a.vref) for (int i = 0; i < 6; i++) vr[i] = a.vref[(size_t)b * 6 + i];
  const int code = a.code ? a.code[b] : a.code_scalar;
  const int k = a.k;

  if (a.mode & kPlanGait) gait_update(s, L, a, k, q7, code);
  if (a.mode & kPlanFootsteps) footsteps_update(s, L, a, a.refresh != 0, a.k_footsteps, q7, hv, vr);
  if (a.mode & kPlanTraj) {
    double tgt[12];
    for (int e = 0; e < 12; e++) tgt[e] = a.target_in ? a.target_in[(size_t)b * 12 + e] : s(L.otgt + e);
    traj_update(s, L, a, k, tgt);
  }
  if ((a.mode & kPlanState) && a.xref) state_compute(a, q7, hv, vr, a.z_average, a.xref + (size_t)b * 12 * (a.n_steps + 1));

  // ---- outputs in the layouts the solver kernels read
  if (a.fsteps && (a.mode & (kPlanFootsteps | kPlanOutputs))) {  // FootstepPlanner::getFootsteps / vectorToMatrix (:235-249)
    double* o = a.fsteps + (size_t)b * a.N_gait * 12;
    for (int i = 0; i < a.N_gait; i++)
      for (int j = 0; j < 4; j++)
        for (int r = 0; r < 3; r++) o[i * 12 + 3 * j + r] = s(FSI(i, r, j));
  }
  if (a.gait && (a.mode & (kPlanGait | kPlanOutputs))) {
    double* o = a.gait + (size_t)b * a.N_gait * 4;
    for (int e = 0; e < a.N_gait * 4; e++) o[e] = s(L.cur + e);
  }
  if (a.target && (a.mode & (kPlanFootsteps | kPlanOutputs)))
    for (int e = 0; e < 12; e++) a.target[(size_t)b * 12 + e] = s(L.otgt + e);
  if (a.feet_pva && (a.mode & (kPlanTraj | kPlanOutputs)))
    for (int e = 0; e < 12; e++) {
      a.feet_pva[(size_t)b * 36 + e] = s(L.pos + e);
      a.feet_pva[(size_t)b * 36 + 12 + e] = s(L.vel + e);
      a.feet_pva[(size_t)b * 36 + 24 + e] = s(L.acc + e);
    }
}
#undef FSI

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
    default: return -1;
  }
}

int planner_launch(const PlannerArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(planner_kernel, dim3((a.B + 63) / 64), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace qrw
