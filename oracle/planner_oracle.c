/*
 * oracle/planner_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see qrw_oracle.h).
 *
 * CPU restatement of the planners that produce the hot path's inputs:
 *   Gait                     /root/reference/src/Gait.cpp:19-260, include/qrw/Gait.hpp
 *   StatePlanner             src/StatePlanner.cpp:12-61
 *   FootstepPlanner          src/FootstepPlanner.cpp:22-249
 *   FootTrajectoryGenerator  src/FootTrajectoryGenerator.cpp:22-151
 * in the order scripts/Controller.py:222-236 calls them.  pinocchio::rpy::matrixToRpy (third-party,
 * absent) is restated from its published definition.  PARITY UNPINNED (no reference vectors).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "qrw_oracle.h"

struct planner_oracle {
  /* Gait */
  int N_gait, n_steps;
  double dt, T_gait, T_mpc;
  double *past, *cur, *des; /* N_gait x 4 */
  double remainingTime;
  int newPhase, is_static;
  double q_static[19];
  /* StatePlanner */
  double h_ref;
  double *xref; /* 12 x (n_steps+1) */
  /* FootstepPlanner */
  double dt_wbc, k_feedback, g, L;
  double shoulders[12], currentFootstep[12], targetFootstep[12], o_targetFootstep[12], nextFootstep[12]; /* 3x4 row-major */
  double Rz[9]; /* FootstepPlanner::Rz (3x3 row-major): zero but (2,2) = 1 after initialize (FootstepPlanner.cpp:10,48), then whatever
                 * the member was last assigned -- after updateFootsteps always computeTargetFootstep's rotation by the base yaw (:214) */
  double *footsteps; /* N_gait x 3 x 4 */
  double *dt_cum, *yaws, *dx, *dy;
  /* FootTrajectoryGenerator */
  int k_mpc;
  double maxHeight, lockTime;
  int feet[4], n_feet;
  double t0s[4], t_swing[4];
  double ft_target[12], Ax[24], Ay[24]; /* 3x4, 6x4, 6x4 row-major */
  double position[12], velocity[12], acceleration[12];
};

#define G(m, i, j) ((m)[(i)*4 + (j)])

static int row_zero(const double *m, int i) { return m[i * 4] == 0 && m[i * 4 + 1] == 0 && m[i * 4 + 2] == 0 && m[i * 4 + 3] == 0; }
static void row_swap(double *m, int a, int b) {
  for (int j = 0; j < 4; j++) { double t = m[a * 4 + j]; m[a * 4 + j] = m[b * 4 + j]; m[b * 4 + j] = t; }
}
static void fill_rows(double *m, int r0, int n, double a, double b, double c, double d) {
  for (int r = r0; r < r0 + n; r++) { m[r * 4] = a; m[r * 4 + 1] = b; m[r * 4 + 2] = c; m[r * 4 + 3] = d; }
}

/* Gait::create_* (src/Gait.cpp:38-108) */
static void create_trot(planner_oracle *o) {
  int N = (int)lround(0.5 * o->T_gait / o->dt);
  memset(o->des, 0, o->N_gait * 4 * sizeof(double));
  fill_rows(o->des, 0, N, 1, 0, 0, 1);
  fill_rows(o->des, N, N, 0, 1, 1, 0);
}
static void create_pacing(planner_oracle *o) {
  int N = (int)lround(0.5 * o->T_gait / o->dt);
  memset(o->des, 0, o->N_gait * 4 * sizeof(double));
  fill_rows(o->des, 0, N, 1, 0, 1, 0);
  fill_rows(o->des, N, N, 0, 1, 0, 1);
}
static void create_bounding(planner_oracle *o) {
  int N = (int)lround(0.5 * o->T_gait / o->dt);
  memset(o->des, 0, o->N_gait * 4 * sizeof(double));
  fill_rows(o->des, 0, N, 1, 1, 0, 0);
  fill_rows(o->des, N, N, 0, 0, 1, 1);
}
static void create_walk(planner_oracle *o) {
  int N = (int)lround(0.25 * o->T_gait / o->dt);
  memset(o->des, 0, o->N_gait * 4 * sizeof(double));
  fill_rows(o->des, 0, N, 0, 1, 1, 1);
  fill_rows(o->des, N, N, 1, 0, 1, 1);
  fill_rows(o->des, 2 * N, N, 1, 1, 0, 1);
  fill_rows(o->des, 3 * N, N, 1, 1, 1, 0);
}
static void create_static(planner_oracle *o) {
  int N = (int)lround(o->T_gait / o->dt);
  memset(o->des, 0, o->N_gait * 4 * sizeof(double));
  fill_rows(o->des, 0, N, 1, 1, 1, 1);
}
/* Gait::create_gait_f (src/Gait.cpp:110-139) */
static void create_gait_f(planner_oracle *o) {
  int i = 0;
  for (int j = 0; j < o->n_steps; j++) {
    memcpy(&o->cur[j * 4], &o->des[i * 4], 4 * sizeof(double));
    i++;
    if (row_zero(o->des, i)) i = 0;
  }
  int index = 1;
  while (!row_zero(o->des, index)) index++;
  for (int k = 0; k < i; k++)
    for (int m = 0; m < index - 1; m++) row_swap(o->des, m, m + 1);
}

/* Gait::getPhaseDuration (src/Gait.cpp:141-182) */
static double gait_phase_duration(planner_oracle *o, int i, int j, double value) {
  double t_phase = 1;
  int a = i;
  while ((!row_zero(o->cur, i + 1)) && (G(o->cur, i + 1, j) == value)) { i++; t_phase++; }
  if (row_zero(o->cur, i + 1)) {
    int k = 0;
    while ((!row_zero(o->des, k)) && (G(o->des, k, j) == value)) { k++; t_phase++; }
  }
  o->remainingTime = t_phase;
  while ((a > 0) && (G(o->cur, a - 1, j) == value)) { a--; t_phase++; }
  if (a == 0) {
    while ((!row_zero(o->past, a)) && (G(o->past, a, j) == value)) { a++; t_phase++; }
  }
  return t_phase * o->dt;
}

/* Gait::rollGait (src/Gait.cpp:221-260) */
static void roll_gait(planner_oracle *o) {
  for (int m = o->n_steps; m > 0; m--) row_swap(o->past, m, m - 1);
  memcpy(&o->past[0], &o->cur[0], 4 * sizeof(double));
  o->newPhase = memcmp(&o->cur[0], &o->cur[4], 4 * sizeof(double)) != 0; /* !row(0).isApprox(row(1)) on 0/1 rows */
  int index = 1;
  while (!row_zero(o->cur, index)) { row_swap(o->cur, index - 1, index); index++; }
  memcpy(&o->cur[(index - 1) * 4], &o->des[0], 4 * sizeof(double));
  index = 1;
  while (!row_zero(o->des, index)) { row_swap(o->des, index - 1, index); index++; }
}

/* Gait::updateGait / changeGait (src/Gait.cpp:184-219) */
void planner_oracle_gait_update(planner_oracle *o, int k, const double *q7, int code) {
  o->is_static = 0;
  if (code == 1) create_pacing(o);
  else if (code == 2) create_bounding(o);
  else if (code == 3) create_trot(o);
  else if (code == 4) {
    create_static(o);
    memcpy(o->q_static, q7, 7 * sizeof(double));
    o->is_static = 1;
  } else if (code == 5) {
    create_walk(o); /* create_walk exists in the reference (Gait.cpp:38-54) but no joystick code selects it; 5 is ours */
  }
  if (k % o->k_mpc == 0) roll_gait(o);
}

/* pinocchio::rpy::matrixToRpy on the rotation of quaternion (x,y,z,w) [third-party, restated] */
static void quat_to_rpy(const double *q, double rpy[3]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
         tzz = tz * z;
  double R00 = 1 - (tyy + tzz), R01 = txy - twz, R10 = txy + twz, R11 = 1 - (txx + tzz), R20 = txz - twy,
         R21 = tyz + twx, R22 = 1 - (txx + tyy);
  double m = sqrt(R21 * R21 + R22 * R22);
  double p = atan2(-R20, m), r, yw;
  if (fabs(fabs(p) - M_PI / 2) < 0.001) {
    r = 0.0;
    yw = -atan2(R01, R11);
  } else {
    yw = atan2(R10, R00);
    r = atan2(R21, R22);
  }
  rpy[0] = r; rpy[1] = p; rpy[2] = yw;
}

/* StatePlanner::computeReferenceStates (src/StatePlanner.cpp:21-61) */
void planner_oracle_state_compute(planner_oracle *o, const double *q7, const double *v6, const double *vref6,
                                  double z_average) {
  int n = o->n_steps, ld = n + 1;
  double rpy[3];
  quat_to_rpy(q7 + 3, rpy);
  double *X = o->xref;
  X[0 * ld] = 0.0; X[1 * ld] = 0.0; X[2 * ld] = q7[2];
  X[3 * ld] = rpy[0]; X[4 * ld] = rpy[1]; X[5 * ld] = 0.0;
  for (int i = 0; i < 3; i++) { X[(6 + i) * ld] = v6[i]; X[(9 + i) * ld] = v6[3 + i]; }
  for (int i = 0; i < n; i++) {
    /* dt_vector_ = LinSpaced(n_steps, dt, T_mpc) (StatePlanner.cpp:18) */
    double dtv = (n == 1) ? o->T_mpc : ((i == n - 1) ? o->T_mpc : o->dt + i * ((o->T_mpc - o->dt) / (n - 1)));
    if (vref6[5] != 0) {
      X[0 * ld + 1 + i] = (vref6[0] * sin(vref6[5] * dtv) + vref6[1] * (cos(vref6[5] * dtv) - 1.0)) / vref6[5];
      X[1 * ld + 1 + i] = (vref6[1] * sin(vref6[5] * dtv) - vref6[0] * (cos(vref6[5] * dtv) - 1.0)) / vref6[5];
    } else {
      X[0 * ld + 1 + i] = vref6[0] * dtv;
      X[1 * ld + 1 + i] = vref6[1] * dtv;
    }
    X[0 * ld + 1 + i] += X[0 * ld];
    X[1 * ld + 1 + i] += X[1 * ld];
    X[2 * ld + 1 + i] = o->h_ref + z_average;
    X[5 * ld + 1 + i] = vref6[5] * dtv;
    X[6 * ld + 1 + i] = vref6[0] * cos(X[5 * ld + 1 + i]) - vref6[1] * sin(X[5 * ld + 1 + i]);
    X[7 * ld + 1 + i] = vref6[0] * sin(X[5 * ld + 1 + i]) + vref6[1] * cos(X[5 * ld + 1 + i]);
    X[11 * ld + 1 + i] = vref6[5];
  }
}

#define FS(o, i, r, c) ((o)->footsteps[((i)*3 + (r)) * 4 + (c)])

/* FootstepPlanner::computeNextFootstep (src/FootstepPlanner.cpp:158-186) */
static void compute_next_footstep(planner_oracle *o, int i, int j, const double *b_v, const double *b_vref) {
  memset(o->nextFootstep, 0, sizeof(o->nextFootstep));
  double t_stance = gait_phase_duration(o, i, j, 1.0);
  double cross[3] = {b_v[1] * b_vref[5] - b_v[2] * b_vref[4], b_v[2] * b_vref[3] - b_v[0] * b_vref[5], 0.0};
  for (int r = 0; r < 3; r++) {
    double v = t_stance * 0.5 * b_v[r];
    v += o->k_feedback * (b_v[r] - b_vref[r]);
    v += 0.5 * sqrt(o->h_ref / o->g) * cross[r];
    o->nextFootstep[r * 4 + j] = v;
  }
  for (int r = 0; r < 2; r++) {
    o->nextFootstep[r * 4 + j] = fmin(o->nextFootstep[r * 4 + j], o->L);
    o->nextFootstep[r * 4 + j] = fmax(o->nextFootstep[r * 4 + j], -o->L);
  }
  for (int r = 0; r < 3; r++) o->nextFootstep[r * 4 + j] += o->shoulders[r * 4 + j];
  for (int c = 0; c < 4; c++) o->nextFootstep[2 * 4 + c] = 0.0;
}

/* FootstepPlanner::computeFootsteps (src/FootstepPlanner.cpp:76-156) */
static void compute_footsteps(planner_oracle *o, int k, const double *b_v, const double *b_vref) {
  int Ng = o->N_gait;
  memset(o->footsteps, 0, Ng * 12 * sizeof(double));
  const double *gait = o->cur;
  for (int j = 0; j < 4; j++)
    if (G(gait, 0, j) == 1.0)
      for (int r = 0; r < 3; r++) FS(o, 0, r, j) = o->currentFootstep[r * 4 + j];
  o->dt_cum[0] = o->dt_wbc * k;
  o->yaws[0] = b_vref[5] * o->dt_cum[0];
  for (int j = 1; j < Ng; j++) {
    o->dt_cum[j] = row_zero(gait, j) ? o->dt_cum[j - 1] : o->dt_cum[j - 1] + o->dt;
    o->yaws[j] = b_vref[5] * o->dt_cum[j];
  }
  if (b_vref[5] != 0) {
    for (int j = 0; j < Ng; j++) {
      o->dx[j] = (b_v[0] * sin(b_vref[5] * o->dt_cum[j]) + b_v[1] * (cos(b_vref[5] * o->dt_cum[j]) - 1.0)) / b_vref[5];
      o->dy[j] = (b_v[1] * sin(b_vref[5] * o->dt_cum[j]) - b_v[0] * (cos(b_vref[5] * o->dt_cum[j]) - 1.0)) / b_vref[5];
    }
  } else {
    for (int j = 0; j < Ng; j++) { o->dx[j] = b_v[0] * o->dt_cum[j]; o->dy[j] = b_v[1] * o->dt_cum[j]; }
  }
  int i = 1;
  while (!row_zero(gait, i)) {
    for (int j = 0; j < 4; j++)
      if (G(gait, i - 1, j) * G(gait, i, j) > 0)
        for (int r = 0; r < 3; r++) FS(o, i, r, j) = FS(o, i - 1, r, j);
    for (int j = 0; j < 4; j++)
      if ((1 - G(gait, i - 1, j)) * G(gait, i, j) > 0) {
        double q_dxdy[3] = {o->dx[i - 1], o->dy[i - 1], 0.0};
        compute_next_footstep(o, i, j, b_v, b_vref);
        double c = cos(o->yaws[i - 1]), s = sin(o->yaws[i - 1]);
        double nx = o->nextFootstep[0 * 4 + j], ny = o->nextFootstep[1 * 4 + j], nz = o->nextFootstep[2 * 4 + j];
        FS(o, i, 0, j) = (c * nx - s * ny + 0.0 * nz) + q_dxdy[0];
        FS(o, i, 1, j) = (s * nx + c * ny + 0.0 * nz) + q_dxdy[1];
        FS(o, i, 2, j) = (0.0 * nx + 0.0 * ny + 1.0 * nz) + q_dxdy[2];
      }
    i++;
  }
}

/* FootstepPlanner::updateFootsteps + computeTargetFootstep + updateTargetFootsteps + updateNewContact
 * (src/FootstepPlanner.cpp:51-74,188-230) */
void planner_oracle_footsteps_update(planner_oracle *o, int refresh, int k, const double *q7, const double *b_v,
                                     const double *b_vref, double *out_target3x4) {
  if (refresh && o->newPhase)
    for (int i = 0; i < 4; i++)
      if (G(o->cur, 0, i) == 1.0)
        for (int r = 0; r < 3; r++) o->currentFootstep[r * 4 + i] = FS(o, 1, r, i);
  double rotation_yaw = o->dt_wbc * b_vref[5];
  double c = cos(rotation_yaw), s = sin(rotation_yaw);
  double dpos[2] = {o->dt_wbc * b_vref[0], o->dt_wbc * b_vref[1]};
  for (int j = 0; j < 4; j++)
    if (G(o->cur, 0, j) == 1.0) {
      double x = o->currentFootstep[0 * 4 + j] - dpos[0], y = o->currentFootstep[1 * 4 + j] - dpos[1];
      o->currentFootstep[0 * 4 + j] = c * x + s * y; /* Rz << c, s, -s, c */
      o->currentFootstep[1 * 4 + j] = -s * x + c * y;
    }
  compute_footsteps(o, k, b_v, b_vref);
  for (int i = 0; i < 4; i++) { /* updateTargetFootsteps */
    int index = 0;
    while (index < o->N_gait - 1 && FS(o, index, 0, i) == 0.0) index++;
    o->targetFootstep[0 * 4 + i] = FS(o, index, 0, i);
    o->targetFootstep[1 * 4 + i] = FS(o, index, 1, i);
    o->targetFootstep[2 * 4 + i] = 0.0;
  }
  double rpy[3];
  quat_to_rpy(q7 + 3, rpy);
  c = cos(rpy[2]); s = sin(rpy[2]);
  /* Rz.topLeftCorner<2, 2>() << c, -s, s, c  (FootstepPlanner.cpp:214): the member's last assignment of an updateFootsteps call
   * (the earlier ones, :63 and :149, are overwritten before the call returns) */
  o->Rz[0] = c; o->Rz[1] = -s; o->Rz[3] = s; o->Rz[4] = c;
  for (int i = 0; i < 4; i++) {
    double x = o->targetFootstep[0 * 4 + i], y = o->targetFootstep[1 * 4 + i];
    o->o_targetFootstep[0 * 4 + i] = (c * x - s * y) + q7[0];
    o->o_targetFootstep[1 * 4 + i] = (s * x + c * y) + q7[1];
  }
  if (out_target3x4) memcpy(out_target3x4, o->o_targetFootstep, sizeof(o->o_targetFootstep));
}

/* FootTrajectoryGenerator::updateFootPosition (src/FootTrajectoryGenerator.cpp:41-106).
 * `Vector4 Az; Az(i, j)` indexes a 4x1 vector with a column index in the reference (out of bounds for j > 0);
 * the evident intent — four coefficients per foot — is what is restated. */
static void update_foot_position(planner_oracle *o, int j, const double *tf /* 3 */) {
  double ddx0 = o->acceleration[0 * 4 + j], ddy0 = o->acceleration[1 * 4 + j];
  double dx0 = o->velocity[0 * 4 + j], dy0 = o->velocity[1 * 4 + j];
  double x0 = o->position[0 * 4 + j], y0 = o->position[1 * 4 + j];
  double t = o->t0s[j], d = o->t_swing[j], dt = o->dt_wbc;
#define P(a, b) pow((a), (b))
  if (t < d - o->lockTime) {
    double den1 = (2 * P((t - d), 2) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
    double den2 = (2 * (P(t, 2) - 2 * t * d + P(d, 2)) * (P(t, 3) - 3 * P(t, 2) * d + 3 * t * P(d, 2) - P(d, 3)));
    for (int ax = 0; ax < 2; ax++) {
      double dd0 = ax ? ddy0 : ddx0, d0 = ax ? dy0 : dx0, p0 = ax ? y0 : x0, tg = tf[ax];
      double *A = ax ? o->Ay : o->Ax;
      A[0 * 4 + j] = (dd0 * P(t, 2) - 2 * dd0 * t * d - 6 * d0 * t + dd0 * P(d, 2) + 6 * d0 * d + 12 * p0 - 12 * tg) / den1;
      A[1 * 4 + j] = (30 * t * tg - 30 * t * p0 - 30 * d * p0 + 30 * d * tg - 2 * P(t, 3) * dd0 - 3 * P(d, 3) * dd0 +
                      14 * P(t, 2) * d0 - 16 * P(d, 2) * d0 + 2 * t * d * d0 + 4 * t * P(d, 2) * dd0 + P(t, 2) * d * dd0) / den1;
      A[2 * 4 + j] = (P(t, 4) * dd0 + 3 * P(d, 4) * dd0 - 8 * P(t, 3) * d0 + 12 * P(d, 3) * d0 + 20 * P(t, 2) * p0 -
                      20 * P(t, 2) * tg + 20 * P(d, 2) * p0 - 20 * P(d, 2) * tg + 80 * t * d * p0 - 80 * t * d * tg +
                      4 * P(t, 3) * d * dd0 + 28 * t * P(d, 2) * d0 - 32 * P(t, 2) * d * d0 - 8 * P(t, 2) * P(d, 2) * dd0) / den1;
      A[3 * 4 + j] = -(P(d, 5) * dd0 + 4 * t * P(d, 4) * dd0 + 3 * P(t, 4) * d * dd0 + 36 * t * P(d, 3) * d0 -
                       24 * P(t, 3) * d * d0 + 60 * t * P(d, 2) * p0 + 60 * P(t, 2) * d * p0 - 60 * t * P(d, 2) * tg -
                       60 * P(t, 2) * d * tg - 8 * P(t, 2) * P(d, 3) * dd0 - 12 * P(t, 2) * P(d, 2) * d0) / den2;
      A[4 * 4 + j] = -(2 * P(d, 5) * d0 - 2 * t * P(d, 5) * dd0 - 10 * t * P(d, 4) * d0 + P(t, 2) * P(d, 4) * dd0 +
                       4 * P(t, 3) * P(d, 3) * dd0 - 3 * P(t, 4) * P(d, 2) * dd0 - 16 * P(t, 2) * P(d, 3) * d0 +
                       24 * P(t, 3) * P(d, 2) * d0 - 60 * P(t, 2) * P(d, 2) * p0 + 60 * P(t, 2) * P(d, 2) * tg) / den1;
      A[5 * 4 + j] = (2 * tg * P(t, 5) - dd0 * P(t, 4) * P(d, 3) - 10 * tg * P(t, 4) * d + 2 * dd0 * P(t, 3) * P(d, 4) +
                      8 * d0 * P(t, 3) * P(d, 3) + 20 * tg * P(t, 3) * P(d, 2) - dd0 * P(t, 2) * P(d, 5) -
                      10 * d0 * P(t, 2) * P(d, 4) - 20 * p0 * P(t, 2) * P(d, 3) + 2 * d0 * t * P(d, 5) +
                      10 * p0 * t * P(d, 4) - 2 * p0 * P(d, 5)) / den2;
    }
    o->ft_target[0 * 4 + j] = tf[0];
    o->ft_target[1 * 4 + j] = tf[1];
  }
  double Az[4];
  double dz = (P((d / 2), 3) * P((d - d / 2), 3));
  Az[0] = -o->maxHeight / dz;
  Az[1] = (3 * d * o->maxHeight) / dz;
  Az[2] = -(3 * P(d, 2) * o->maxHeight) / dz;
  Az[3] = (P(d, 3) * o->maxHeight) / dz;
  double ev = t + dt;
  if (t < 0.0 || t > d) {
    o->position[0 * 4 + j] = x0; o->position[1 * 4 + j] = y0;
    o->velocity[0 * 4 + j] = 0.0; o->velocity[1 * 4 + j] = 0.0;
    o->acceleration[0 * 4 + j] = 0.0; o->acceleration[1 * 4 + j] = 0.0;
  } else {
    for (int ax = 0; ax < 2; ax++) {
      const double *A = ax ? o->Ay : o->Ax;
      o->position[ax * 4 + j] = A[5 * 4 + j] + A[4 * 4 + j] * ev + A[3 * 4 + j] * P(ev, 2) + A[2 * 4 + j] * P(ev, 3) +
                                A[1 * 4 + j] * P(ev, 4) + A[0 * 4 + j] * P(ev, 5);
      o->velocity[ax * 4 + j] = A[4 * 4 + j] + 2 * A[3 * 4 + j] * ev + 3 * A[2 * 4 + j] * P(ev, 2) +
                                4 * A[1 * 4 + j] * P(ev, 3) + 5 * A[0 * 4 + j] * P(ev, 4);
      o->acceleration[ax * 4 + j] = 2 * A[3 * 4 + j] + 3 * 2 * A[2 * 4 + j] * ev + 4 * 3 * A[1 * 4 + j] * P(ev, 2) +
                                    5 * 4 * A[0 * 4 + j] * P(ev, 3);
    }
  }
  o->velocity[2 * 4 + j] = 3 * Az[3] * P(ev, 2) + 4 * Az[2] * P(ev, 3) + 5 * Az[1] * P(ev, 4) + 6 * Az[0] * P(ev, 5);
  o->acceleration[2 * 4 + j] = 2 * 3 * Az[3] * ev + 3 * 4 * Az[2] * P(ev, 2) + 4 * 5 * Az[1] * P(ev, 3) + 5 * 6 * Az[0] * P(ev, 4);
  o->position[2 * 4 + j] = Az[3] * P(ev, 3) + Az[2] * P(ev, 4) + Az[1] * P(ev, 5) + Az[0] * P(ev, 6);
#undef P
}

/* FootTrajectoryGenerator::update (src/FootTrajectoryGenerator.cpp:108-151) */
void planner_oracle_traj_update(planner_oracle *o, int k, const double *target3x4) {
  if ((k % o->k_mpc) == 0) {
    o->n_feet = 0;
    for (int i = 0; i < 4; i++)
      if (G(o->cur, 0, i) == 0) o->feet[o->n_feet++] = i;
    if (o->n_feet == 0) return;
    for (int jj = 0; jj < o->n_feet; jj++) {
      int i = o->feet[jj];
      o->t_swing[i] = gait_phase_duration(o, 0, i, 0.0);
      double value = o->t_swing[i] - (o->remainingTime * o->k_mpc - ((k + 1) % o->k_mpc)) * o->dt_wbc - o->dt_wbc;
      o->t0s[i] = fmax(0.0, value);
    }
  } else {
    if (o->n_feet == 0) return;
    for (int jj = 0; jj < o->n_feet; jj++) {
      double value = o->t0s[o->feet[jj]] + o->dt_wbc;
      o->t0s[o->feet[jj]] = fmax(0.0, value);
    }
  }
  for (int jj = 0; jj < o->n_feet; jj++) {
    int i = o->feet[jj];
    double tf[3] = {target3x4[0 * 4 + i], target3x4[1 * 4 + i], target3x4[2 * 4 + i]};
    update_foot_position(o, i, tf);
  }
}

/* constructors + initialize() of the four classes (Gait.cpp:19-36, StatePlanner.cpp:12-19, FootstepPlanner.cpp:22-49,
 * FootTrajectoryGenerator.cpp:22-38), with the arguments scripts/Controller.py:119-137 passes */
planner_oracle *planner_oracle_create(double dt_mpc, double dt_wbc, double T_gait, double T_mpc, int N_gait, int k_mpc,
                                      double h_ref, const double *shoulders3x4, double max_height, double lock_time,
                                      const double *init_target3x4, const double *init_foot_pos3x4) {
  planner_oracle *o = (planner_oracle *)calloc(1, sizeof(*o));
  o->dt = dt_mpc; o->T_gait = T_gait; o->T_mpc = T_mpc; o->N_gait = N_gait; o->k_mpc = k_mpc;
  o->n_steps = (int)lround(T_mpc / dt_mpc);
  if ((o->n_steps > N_gait) || ((int)lround(T_gait / dt_mpc) > N_gait)) { free(o); return NULL; } /* Gait.cpp:30-31 throws */
  o->past = (double *)calloc(N_gait * 4, sizeof(double));
  o->cur = (double *)calloc(N_gait * 4, sizeof(double));
  o->des = (double *)calloc(N_gait * 4, sizeof(double));
  create_trot(o);
  create_gait_f(o);
  o->h_ref = h_ref;
  o->xref = (double *)calloc(12 * (o->n_steps + 1), sizeof(double));
  o->dt_wbc = dt_wbc; o->k_feedback = 0.03; o->g = 9.81; o->L = 0.155;
  memcpy(o->shoulders, shoulders3x4, 12 * sizeof(double));
  memcpy(o->currentFootstep, shoulders3x4, 12 * sizeof(double));
  memcpy(o->targetFootstep, shoulders3x4, 12 * sizeof(double));
  memcpy(o->o_targetFootstep, shoulders3x4, 12 * sizeof(double));
  memset(o->Rz, 0, sizeof(o->Rz)); o->Rz[8] = 1.0; /* FootstepPlanner.cpp:10,48 */
  o->footsteps = (double *)calloc(N_gait * 12, sizeof(double));
  o->dt_cum = (double *)calloc(N_gait, sizeof(double)); o->yaws = (double *)calloc(N_gait, sizeof(double));
  o->dx = (double *)calloc(N_gait, sizeof(double)); o->dy = (double *)calloc(N_gait, sizeof(double));
  o->maxHeight = max_height; o->lockTime = lock_time;
  memcpy(o->ft_target, init_target3x4, 12 * sizeof(double));
  memcpy(o->position, init_foot_pos3x4, 12 * sizeof(double));
  return o;
}
void planner_oracle_destroy(planner_oracle *o) {
  if (!o) return;
  free(o->past); free(o->cur); free(o->des); free(o->xref); free(o->footsteps);
  free(o->dt_cum); free(o->yaws); free(o->dx); free(o->dy);
  free(o);
}

/* the planner calls of one control iteration in the order of scripts/Controller.py:222-236 */
void planner_oracle_step(planner_oracle *o, int k, const double *q7, const double *h_v6, const double *vref6, int code) {
  double target[12];
  planner_oracle_gait_update(o, k, q7, code);
  planner_oracle_footsteps_update(o, (k % o->k_mpc == 0) && (k != 0), o->k_mpc - k % o->k_mpc, q7, h_v6, vref6, target);
  planner_oracle_traj_update(o, k, target);
  planner_oracle_state_compute(o, q7, h_v6, vref6, 0.0);
}

double planner_oracle_phase_duration(planner_oracle *o, int i, int j, double value) { return gait_phase_duration(o, i, j, value); }
void planner_oracle_get_gaits(const planner_oracle *o, double *past, double *cur, double *des) {
  if (past) memcpy(past, o->past, o->N_gait * 4 * sizeof(double));
  if (cur) memcpy(cur, o->cur, o->N_gait * 4 * sizeof(double));
  if (des) memcpy(des, o->des, o->N_gait * 4 * sizeof(double));
}
void planner_oracle_get_flags(const planner_oracle *o, double *out4) {
  out4[0] = o->newPhase; out4[1] = o->is_static; out4[2] = o->remainingTime; out4[3] = o->n_feet;
}
void planner_oracle_get_xref(const planner_oracle *o, double *xref) { memcpy(xref, o->xref, 12 * (o->n_steps + 1) * sizeof(double)); }
void planner_oracle_get_footsteps(const planner_oracle *o, double *fsteps_Ngx12, double *target3x4, double *o_target3x4) {
  if (fsteps_Ngx12)
    for (int i = 0; i < o->N_gait; i++)
      for (int j = 0; j < 4; j++)
        for (int r = 0; r < 3; r++) fsteps_Ngx12[i * 12 + 3 * j + r] = FS(o, i, r, j); /* vectorToMatrix :235-249 */
  if (target3x4) memcpy(target3x4, o->targetFootstep, sizeof(o->targetFootstep));
  if (o_target3x4) memcpy(o_target3x4, o->o_targetFootstep, sizeof(o->o_targetFootstep));
}
void planner_oracle_get_Rz(const planner_oracle *o, double *out9) { memcpy(out9, o->Rz, sizeof(o->Rz)); } /* FootstepPlanner::getRz :236 */
void planner_oracle_get_feet(const planner_oracle *o, double *pos, double *vel, double *acc, double *t0s, double *t_swing) {
  if (pos) memcpy(pos, o->position, sizeof(o->position));
  if (vel) memcpy(vel, o->velocity, sizeof(o->velocity));
  if (acc) memcpy(acc, o->acceleration, sizeof(o->acceleration));
  if (t0s) memcpy(t0s, o->t0s, sizeof(o->t0s));
  if (t_swing) memcpy(t_swing, o->t_swing, sizeof(o->t_swing));
}
