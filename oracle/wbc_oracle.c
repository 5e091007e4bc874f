/*
 * oracle/wbc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see qrw_oracle.h).
 *
 * CPU restatement of the whole-body-control step of the reference:
 *   InvKin::refreshAndCompute        /root/reference/src/InvKin.cpp:23-73
 *   pseudoInverse<>                  include/qrw/InvKin.hpp:60-66
 *   QPWBC (ctor, create_*, compute_matrices, update_PQ, call_solver, retrieve_result, run)
 *                                    src/QPWBC.cpp:4-30,85-211,213-297,310-343,481-537
 *   Solo12InvKin.refreshAndCompute   scripts/solo12InvKin.py:44-69
 *   wbc_controller.compute           scripts/QP_WBC.py:52-131
 * PARITY UNPINNED — see qrw_oracle.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "osqp_restate.h"
#include "qrw_oracle.h"

/* ------------------------------ InvKin ------------------------------ */

static void inv3(const double *M, int ld, double *o /* 3x3 row-major */) {
  /* 3x3 inverse by cofactors (Eigen uses PartialPivLU for this dynamic-size block,
   * InvKin.cpp:56; the two agree to rounding) */
  double a = M[0], b = M[1], c = M[2], d = M[ld], e = M[ld + 1], f = M[ld + 2], g = M[2 * ld], h = M[2 * ld + 1],
         i = M[2 * ld + 2];
  double c00 = e * i - f * h, c10 = f * g - d * i, c20 = d * h - e * g;
  double invdet = 1.0 / (a * c00 + b * c10 + c * c20);
  o[0] = c00 * invdet; o[1] = (c * h - b * i) * invdet; o[2] = (b * f - c * e) * invdet;
  o[3] = c10 * invdet; o[4] = (a * i - c * g) * invdet; o[5] = (c * d - a * f) * invdet;
  o[6] = c20 * invdet; o[7] = (b * g - a * h) * invdet; o[8] = (a * e - b * d) * invdet;
}

/* InvKin::refreshAndCompute, InvKin.cpp:23-73.  Gains InvKin.hpp:56-57. */
void invkin_oracle_refresh_and_compute(const double *contacts, const double *goals, const double *vgoals,
                                       const double *agoals, const double *posf, const double *vf, const double *wf,
                                       const double *af, const double *Jf, double *ddq, double *dq_cmd,
                                       double *q_step) {
  const double Kp = 100.0, Kd = 2.0 * sqrt(100.0);
  double acc[12], x_err[12], dx_r[12];
  for (int i = 0; i < 4; i++) {
    double pref[3], vref[3], aref[3], afeet[3], w[3], v[3], cr[3];
    for (int c = 0; c < 3; c++) { /* goals are 3x4: column i */
      pref[c] = goals[c * 4 + i];
      vref[c] = vgoals[c * 4 + i];
      aref[c] = agoals[c * 4 + i];
      w[c] = wf[i * 3 + c];
      v[c] = vf[i * 3 + c];
    }
    cr[0] = w[1] * v[2] - w[2] * v[1]; /* cross3(wf, vf), InvKin.cpp:14-20 */
    cr[1] = w[2] * v[0] - w[0] * v[2];
    cr[2] = w[0] * v[1] - w[1] * v[0];
    for (int c = 0; c < 3; c++) {
      double perr = pref[c] - posf[i * 3 + c];
      afeet[c] = +Kp * perr - Kd * (v[c] - vref[c]) + aref[c];
      if (contacts[i] != 0.0) afeet[c] *= 0.0;
      afeet[c] -= af[i * 3 + c] + cr[c];
      acc[3 * i + c] = afeet[c];
      x_err[3 * i + c] = perr;
      dx_r[3 * i + c] = vref[c];
    }
  }
  for (int i = 0; i < 4; i++) {
    double iJ[9];
    inv3(&Jf[(3 * i) * 12 + 3 * i], 12, iJ);
    for (int r = 0; r < 3; r++) {
      double a = 0, b = 0, c = 0;
      for (int k = 0; k < 3; k++) {
        a += iJ[r * 3 + k] * acc[3 * i + k];
        b += iJ[r * 3 + k] * dx_r[3 * i + k];
        c += iJ[r * 3 + k] * x_err[3 * i + k];
      }
      ddq[3 * i + r] = a;
      dq_cmd[3 * i + r] = b;
      q_step[3 * i + r] = c;
    }
  }
}

/* ------------------------------ pseudoInverse ------------------------------ */
/* InvKin.hpp:60-66 applied to the symmetric 6x6 Y (QPWBC.cpp:493): for a symmetric
 * matrix the SVD is the eigen-decomposition up to signs, so V diag(1/s_i or 0) U^H is
 * computed from a cyclic-Jacobi eigen-decomposition. tolerance = eps * 6 * s_0 where s_0 is
 * the largest singular value (JacobiSVD sorts descending). */
static void pinv_sym6(const double *Y, double *Yinv) {
  double A[36], V[36];
  memcpy(A, Y, sizeof(A));
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      A[i * 6 + j] = 0.5 * (Y[i * 6 + j] + Y[j * 6 + i]);
      V[i * 6 + j] = (i == j);
    }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 6; p++)
      for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q];
    if (off == 0.0) break;
    for (int p = 0; p < 6; p++)
      for (int q = p + 1; q < 6; q++) {
        if (A[p * 6 + q] == 0.0) continue;
        double theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.0 * A[p * 6 + q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; k++) {
          double akp = A[k * 6 + p], akq = A[k * 6 + q];
          A[k * 6 + p] = c * akp - s * akq;
          A[k * 6 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; k++) {
          double apk = A[p * 6 + k], aqk = A[q * 6 + k];
          A[p * 6 + k] = c * apk - s * aqk;
          A[q * 6 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 6; k++) {
          double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
          V[k * 6 + p] = c * vkp - s * vkq;
          V[k * 6 + q] = s * vkp + c * vkq;
        }
      }
  }
  double smax = 0;
  for (int i = 0; i < 6; i++) smax = fmax(smax, fabs(A[i * 6 + i]));
  double tol = 2.220446049250313e-16 * 6.0 * smax;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double a = 0;
      for (int k = 0; k < 6; k++) {
        double lam = A[k * 6 + k];
        double inv = (fabs(lam) > tol) ? 1.0 / lam : 0.0;
        a += V[i * 6 + k] * inv * V[j * 6 + k];
      }
      Yinv[i * 6 + j] = a;
    }
}

/* ------------------------------ QPWBC ------------------------------ */
struct qpwbc_oracle {
  int initialized;
  double mu;
  double G[240];                 /* 20x12 row-major */
  double A[72], gamma[6], H[144], g[12];
  double f_res[12], ddq_res[6];
  oq_csc ML, P;
  double NK_up[20], NK_low[20], Q[12];
  oq_work *work;
  oq_settings settings;
};

qpwbc_oracle *qpwbc_oracle_create(void) { /* QPWBC::QPWBC, QPWBC.cpp:4-30 */
  qpwbc_oracle *o = (qpwbc_oracle *)calloc(1, sizeof(*o));
  o->mu = 0.9; /* QPWBC.hpp:30 (a double here, unlike MPC.cpp:18) */
  double SC[15] = {0};
  int a[9] = {0, 1, 2, 3, 0, 1, 2, 3, 4}, b[9] = {0, 0, 1, 1, 2, 2, 2, 2, 2};
  double c[9] = {1.0, -1.0, 1.0, -1.0, -o->mu, -o->mu, -o->mu, -o->mu, -1};
  for (int i = 0; i <= 8; i++) SC[a[i] * 3 + b[i]] = -c[i];
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 5; r++)
      for (int cc = 0; cc < 3; cc++) o->G[(5 * i + r) * 12 + 3 * i + cc] = SC[r * 3 + cc];
  for (int i = 0; i < 20; i++) { o->NK_up[i] = 25.0; o->NK_low[i] = 0.0; }
  oq_set_default_settings(&o->settings);
  return o;
}

void qpwbc_oracle_destroy(qpwbc_oracle *o) {
  if (!o) return;
  free(o->ML.p); free(o->ML.i); free(o->ML.x); free(o->P.p); free(o->P.i); free(o->P.x);
  oq_cleanup(o->work);
  free(o);
}

static void qpwbc_create_matrices(qpwbc_oracle *o) {
  /* create_ML, QPWBC.cpp:85-148: all 240 entries of G stored (zeros explicit), column-major CSC */
  o->ML.m = 20; o->ML.n = 12;
  o->ML.p = (int *)malloc(13 * sizeof(int));
  o->ML.i = (int *)malloc(240 * sizeof(int));
  o->ML.x = (double *)malloc(240 * sizeof(double));
  for (int j = 0; j <= 12; j++) o->ML.p[j] = 20 * j;
  for (int j = 0; j < 12; j++)
    for (int i = 0; i < 20; i++) { o->ML.i[20 * j + i] = i; o->ML.x[20 * j + i] = o->G[i * 12 + j]; }
  /* create_weight_matrices, :151-211: full upper triangle, placeholder 1.0 */
  o->P.m = 12; o->P.n = 12;
  o->P.p = (int *)malloc(13 * sizeof(int));
  o->P.i = (int *)malloc(78 * sizeof(int));
  o->P.x = (double *)malloc(78 * sizeof(double));
  int cpt = 0;
  for (int j = 0; j < 12; j++) {
    o->P.p[j] = cpt;
    for (int i = 0; i <= j; i++) { o->P.i[cpt] = i; o->P.x[cpt] = 1.0; cpt++; }
  }
  o->P.p[12] = cpt;
  memset(o->Q, 0, sizeof(o->Q));
}

/* QPWBC::compute_matrices, QPWBC.cpp:481-498; Q1 = 0.1 I6, Q2 = 5 I12 (QPWBC.hpp:26-27) */
static void qpwbc_compute_matrices(qpwbc_oracle *o, const double *M, const double *Jc, const double *f_cmd,
                                   const double *RNEA) {
  double Y[36], X[72], Yinv[36], Xf[6], t[6], AtQ1[72];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) Y[i * 6 + j] = M[i * 18 + j];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 12; j++) X[i * 12 + j] = Jc[j * 18 + i];
  pinv_sym6(Y, Yinv);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 12; j++) {
      double a = 0;
      for (int k = 0; k < 6; k++) a += Yinv[i * 6 + k] * X[k * 12 + j];
      o->A[i * 12 + j] = a;
    }
  for (int i = 0; i < 6; i++) {
    double a = 0;
    for (int j = 0; j < 12; j++) a += X[i * 12 + j] * f_cmd[j];
    Xf[i] = a;
    t[i] = Xf[i] - RNEA[i];
  }
  for (int i = 0; i < 6; i++) {
    double a = 0;
    for (int k = 0; k < 6; k++) a += Yinv[i * 6 + k] * t[k];
    o->gamma[i] = a;
  }
  for (int i = 0; i < 12; i++)
    for (int k = 0; k < 6; k++) AtQ1[i * 6 + k] = o->A[k * 12 + i] * 0.1;
  for (int i = 0; i < 12; i++) {
    for (int j = 0; j < 12; j++) {
      double a = 0;
      for (int k = 0; k < 6; k++) a += AtQ1[i * 6 + k] * o->A[k * 12 + j];
      o->H[i * 12 + j] = a + ((i == j) ? 5.0 : 0.0);
    }
    double b = 0;
    for (int k = 0; k < 6; k++) b += AtQ1[i * 6 + k] * o->gamma[k];
    o->g[i] = b;
  }
}

/* QPWBC::run, QPWBC.cpp:310-390 */
int qpwbc_oracle_run(qpwbc_oracle *o, const double *M, const double *Jc, const double *f_cmd, const double *RNEA,
                     const double *k_contact) {
  (void)k_contact; /* accepted, unused: the force ramp is commented out (:345-362) */
  if (!o->initialized) qpwbc_create_matrices(o);
  qpwbc_compute_matrices(o, M, Jc, f_cmd, RNEA);
  int cpt = 0; /* update_PQ, :520-537 */
  for (int i = 0; i < 12; i++)
    for (int j = 0; j <= i; j++) o->P.x[cpt++] = o->H[j * 12 + i];
  for (int i = 0; i < 12; i++) o->Q[i] = o->g[i];
  const double Nz_max = 25.0; /* :337-343 */
  for (int i = 0; i < 20; i++) {
    double Gf = 0;
    for (int j = 0; j < 12; j++) Gf += o->G[i * 12 + j] * f_cmd[j];
    o->NK_low[i] = -Gf;
    o->NK_up[i] = -Gf + Nz_max;
  }
  if (!o->initialized) { /* call_solver, :213-275 */
    o->settings.eps_abs = (float)1e-5;
    o->settings.eps_rel = (float)1e-5;
    o->settings.adaptive_rho = 1;
    o->settings.adaptive_rho_interval = 200;
    o->settings.adaptive_rho_tolerance = (float)5.0;
    o->work = oq_setup(&o->P, &o->ML, o->Q, o->NK_low, o->NK_up, &o->settings, NULL);
    if (!o->work) return 1;
    o->initialized = 1;
  } else {
    oq_update_P(o->work, o->P.x);
    oq_update_lin_cost(o->work, o->Q);
    oq_update_upper_bound(o->work, o->NK_up);
    oq_update_lower_bound(o->work, o->NK_low);
  }
  oq_solve(o->work);
  const double *sol = oq_solution_x(o->work); /* retrieve_result, :277-297 */
  for (int k = 0; k < 12; k++) o->f_res[k] = sol[k];
  for (int i = 0; i < 6; i++) {
    double a = 0;
    for (int k = 0; k < 12; k++) a += o->A[i * 12 + k] * o->f_res[k];
    o->ddq_res[i] = a + o->gamma[i];
  }
  for (int k = 0; k < 12; k++) o->f_res[k] += f_cmd[k];
  return 0;
}

void qpwbc_oracle_get_f_res(const qpwbc_oracle *o, double *f) { memcpy(f, o->f_res, sizeof(o->f_res)); }
void qpwbc_oracle_get_ddq_res(const qpwbc_oracle *o, double *d) { memcpy(d, o->ddq_res, sizeof(o->ddq_res)); }
void qpwbc_oracle_get_H(const qpwbc_oracle *o, double *H) { memcpy(H, o->H, sizeof(o->H)); }
int qpwbc_oracle_iter(const qpwbc_oracle *o) { return o->work ? oq_info_iter(o->work) : -1; }
int qpwbc_oracle_status(const qpwbc_oracle *o) { return o->work ? oq_info_status(o->work) : OQ_UNSOLVED; }
double qpwbc_oracle_rho(const qpwbc_oracle *o) { return o->work ? oq_info_rho(o->work) : 0.0; }

/* ------------------------- wbc_controller ------------------------- */
struct wbc_oracle {
  double dt;
  qpwbc_oracle *box_qp;
  double k_since_contact[4];
  double feet_pos[12], feet_err[12], feet_vel[12]; /* 3x4 row-major */
  double M[324];
  int have_M;
};

wbc_oracle *wbc_oracle_create(double dt) {
  wbc_oracle *o = (wbc_oracle *)calloc(1, sizeof(*o));
  o->dt = dt;
  o->box_qp = qpwbc_oracle_create();
  return o;
}
void wbc_oracle_destroy(wbc_oracle *o) {
  if (!o) return;
  qpwbc_oracle_destroy(o->box_qp);
  free(o);
}

/* wbc_controller.compute, scripts/QP_WBC.py:52-131 (with Solo12InvKin.refreshAndCompute,
 * scripts/solo12InvKin.py:44-69 inlined where the Python calls it) */
int wbc_oracle_compute(wbc_oracle *o, const double *q, const double *dq, const double *f_cmd, const double *contacts,
                       const double *pgoals, const double *vgoals, const double *agoals, double *tau_ff, double *qdes,
                       double *vdes, double *f_with_delta, double *ddq_res) {
  for (int i = 0; i < 4; i++) { /* QP_WBC.py:65-66 */
    o->k_since_contact[i] += contacts[i];
    o->k_since_contact[i] *= contacts[i];
  }
  /* solo12InvKin.py:47-67 on the fixed-base model with q[7:], dq[6:] */
  double posf[12], vf[12], wf[12], af[12], Jf[144], ddq12[12], dq_cmd12[12], q_step[12];
  rbd_oracle_fixed_feet(q + 7, dq + 6, posf, vf, wf, af, Jf);
  invkin_oracle_refresh_and_compute(contacts, pgoals, vgoals, agoals, posf, vf, wf, af, Jf, ddq12, dq_cmd12, q_step);
  double ddq_cmd[18] = {0}, dq_cmd[18] = {0}, q_cmd[19] = {0};
  for (int i = 0; i < 12; i++) {
    ddq_cmd[6 + i] = ddq12[i];
    dq_cmd[6 + i] = dq_cmd12[i];
    q_cmd[7 + i] = q[7 + i] + q_step[i];
  }
  for (int i = 0; i < 4; i++) /* QP_WBC.py:73-80 logs */
    for (int c = 0; c < 3; c++) {
      o->feet_pos[c * 4 + i] = posf[i * 3 + c];
      o->feet_err[c * 4 + i] = pgoals[c * 4 + i] - posf[i * 3 + c];
      o->feet_vel[c * 4 + i] = vf[i * 3 + c];
    }
  /* QP_WBC.py:89-93: crba at the NEUTRAL configuration, top-left block masked to its diagonal */
  if (!o->have_M) {
    double q_tmp[19] = {0};
    q_tmp[6] = 1.0;
    rbd_oracle_crba(q_tmp, o->M);
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++)
        if (i != j) o->M[i * 18 + j] = 0.0;
    o->have_M = 1;
  }
  /* QP_WBC.py:96-101 */
  double Jall[216], Jc[216] = {0};
  rbd_oracle_feet_jacobians(q, Jall);
  for (int i = 0; i < 4; i++)
    if (contacts[i] != 0.0) memcpy(&Jc[3 * i * 18], &Jall[3 * i * 18], 3 * 18 * sizeof(double));
  double tau[18];
  rbd_oracle_rnea(q, dq, ddq_cmd, tau); /* :104 */
  qpwbc_oracle_run(o->box_qp, o->M, Jc, f_cmd, tau, o->k_since_contact); /* :107 (RNEA = tau[:6]) */
  double deltaddq[6], f[12], ddq_with_delta[18];
  qpwbc_oracle_get_ddq_res(o->box_qp, deltaddq);
  qpwbc_oracle_get_f_res(o->box_qp, f);
  memcpy(ddq_with_delta, ddq_cmd, sizeof(ddq_cmd));
  for (int i = 0; i < 6; i++) ddq_with_delta[i] += deltaddq[i];
  rbd_oracle_rnea(q, dq, ddq_with_delta, tau); /* :116 */
  if (tau_ff)
    for (int j = 0; j < 12; j++) { /* :117 */
      double a = 0;
      for (int r = 0; r < 12; r++) a += Jc[r * 18 + 6 + j] * f[r];
      tau_ff[j] = tau[6 + j] - a;
    }
  if (vdes) memcpy(vdes, dq_cmd, sizeof(dq_cmd));
  if (qdes) memcpy(qdes, q_cmd, sizeof(q_cmd));
  if (f_with_delta) memcpy(f_with_delta, f, sizeof(f));
  if (ddq_res) memcpy(ddq_res, deltaddq, sizeof(deltaddq));
  return 0;
}

int wbc_oracle_qp_iter(const wbc_oracle *o) { return qpwbc_oracle_iter(o->box_qp); }
int wbc_oracle_qp_status(const wbc_oracle *o) { return qpwbc_oracle_status(o->box_qp); }
double wbc_oracle_qp_rho(const wbc_oracle *o) { return qpwbc_oracle_rho(o->box_qp); }
void wbc_oracle_get_feet(const wbc_oracle *o, double *p, double *e, double *v) {
  if (p) memcpy(p, o->feet_pos, sizeof(o->feet_pos));
  if (e) memcpy(e, o->feet_err, sizeof(o->feet_err));
  if (v) memcpy(v, o->feet_vel, sizeof(o->feet_vel));
}
void wbc_oracle_get_k_since_contact(const wbc_oracle *o, double *k4) {
  memcpy(k4, o->k_since_contact, sizeof(o->k_since_contact));
}

int wbc_oracle_compute_batch(wbc_oracle **o, int B, const double *q, const double *dq, const double *f_cmd,
                             const double *contacts, const double *pgoals, const double *vgoals, const double *agoals,
                             double *tau_ff, double *qdes, double *vdes, double *f_with_delta, int threads) {
  int bad = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(| : bad)
  for (int b = 0; b < B; b++)
    bad |= wbc_oracle_compute(o[b], q + b * 19, dq + b * 18, f_cmd + b * 12, contacts + b * 4, pgoals + b * 12,
                              vgoals + b * 12, agoals + b * 12, tau_ff ? tau_ff + b * 12 : 0, qdes ? qdes + b * 19 : 0,
                              vdes ? vdes + b * 18 : 0, f_with_delta ? f_with_delta + b * 12 : 0, 0);
  return bad;
}
