/*
 * oracle/mpc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see qrw_oracle.h).
 *
 * Line-by-line CPU restatement of /root/reference/src/MPC.cpp (class MPC,
 * include/qrw/MPC.hpp).  Every function cites the reference lines it follows.
 * PARITY UNPINNED (no reference vectors exist; OSQP/Eigen absent) — see qrw_oracle.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "osqp_restate.h"
#include "qrw_oracle.h"

struct mpc_oracle {
  double dt, mass, mu, T_gait, h_ref;
  int n_steps, N_gait, cpt_ML, cpt_P;
  double gI[9];         /* row-major 3x3, MPC.cpp:25-26 */
  double footholds[12]; /* 3x4 row-major, MPC.cpp:24 */
  double g[12], offset_CoM[3];
  double A[144], B[144]; /* 12x12 row-major, MPC.hpp:33-34 */
  double x0[12], x_next[12];
  double *xref;        /* 12 x (N+1) row-major */
  double *x_f_applied; /* 24 x N row-major */
  int *gait;           /* N_gait x 4 */
  int *S_gait;         /* 12 N */
  oq_csc ML, P;
  int i_x_B[48], i_y_B[48], i_update_B[48];
  int *i_off;
  double *NK_up, *NK_low, *Q, *D; /* D: dense 12N x 12N as in MPC.cpp:275 */
  oq_work *work;
  oq_settings settings;
  int *perm;
};

/* ---- st_to_cc_size / st_to_cc_index / st_to_cc_values (src/st_to_cc.cpp:1622-1854):
 * sort triplets by (column, row), keep unique pairs, SUM duplicates, keep explicit zeros. */
typedef struct { int r, c; double v; } trip;
static int trip_cmp(const void *a, const void *b) {
  const trip *x = (const trip *)a, *y = (const trip *)b;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  return 0;
}
static void st_to_cc(int nst, const int *ist, const int *jst, const double *ast, int m, int n, oq_csc *out) {
  trip *t = (trip *)malloc((nst ? nst : 1) * sizeof(trip));
  for (int k = 0; k < nst; k++) { t[k].r = ist[k]; t[k].c = jst[k]; t[k].v = ast[k]; }
  qsort(t, nst, sizeof(trip), trip_cmp);
  int ncc = 0;
  for (int k = 0; k < nst; k++)
    if (k == 0 || t[k].r != t[k - 1].r || t[k].c != t[k - 1].c) ncc++;
  out->m = m; out->n = n;
  out->p = (int *)calloc(n + 1, sizeof(int));
  out->i = (int *)malloc((ncc ? ncc : 1) * sizeof(int));
  out->x = (double *)calloc(ncc ? ncc : 1, sizeof(double));
  int kcc = -1;
  for (int k = 0; k < nst; k++) {
    if (k == 0 || t[k].r != t[k - 1].r || t[k].c != t[k - 1].c) {
      kcc++;
      out->i[kcc] = t[k].r;
      out->p[t[k].c + 1]++;
    }
    out->x[kcc] += t[k].v;
  }
  for (int j = 0; j < n; j++) out->p[j + 1] += out->p[j];
  free(t);
}

/* MPC::MPC, MPC.cpp:3-32 */
mpc_oracle *mpc_oracle_create(double dt_in, int n_steps_in, double T_gait_in, int N_gait) {
  mpc_oracle *o = (mpc_oracle *)calloc(1, sizeof(mpc_oracle));
  int N = n_steps_in;
  o->dt = dt_in;
  o->n_steps = N;
  o->T_gait = T_gait_in;
  o->N_gait = N_gait;
  o->xref = (double *)calloc(12 * (1 + N), sizeof(double));
  o->S_gait = (int *)calloc(12 * N, sizeof(int));
  o->x_f_applied = (double *)calloc(24 * N, sizeof(double));
  o->gait = (int *)calloc((N_gait + 1) * 4, sizeof(int)); /* + one all-zero row: construct_S / update_ML stop there when the table is full */
  o->mass = 2.50000279f; /* float literal promoted, :17 */
  o->mu = 0.9f;          /* :18 */
  o->offset_CoM[2] = -0.03; /* :21 */
  const double fh[12] = {0.19, 0.19, -0.19, -0.19, 0.15005, -0.15005, 0.15005, -0.15005, 0.0, 0.0, 0.0, 0.0};
  memcpy(o->footholds, fh, sizeof(fh)); /* :24 (Eigen comma-init is row-major) */
  const double gi[9] = {3.09249e-2, -8.00101e-7, 1.865287e-5, -8.00101e-7, 5.106100e-2,
                        1.245813e-4, 1.865287e-5, 1.245813e-4, 6.939757e-2};
  memcpy(o->gI, gi, sizeof(gi)); /* :25-26 */
  o->h_ref = 0.2027682f;         /* :27-28 */
  o->g[8] = -9.81f * dt_in;      /* :29 */
  for (int i = 0; i < 12; i++) o->A[i * 12 + i] = 1.0; /* MPC.hpp:33 A = Identity */
  oq_set_default_settings(&o->settings);               /* :31 */
  o->NK_up = (double *)calloc(44 * N, sizeof(double));
  o->NK_low = (double *)calloc(44 * N, sizeof(double));
  o->Q = (double *)calloc(24 * N, sizeof(double));
  o->i_off = (int *)calloc(12 * N, sizeof(int));
  /* elimination order for the oracle's banded Cholesky: (f_k, X_k) interleaved per step */
  o->perm = (int *)malloc(24 * N * sizeof(int));
  for (int k = 0; k < N; k++)
    for (int i = 0; i < 12; i++) {
      o->perm[12 * k + i] = 24 * k + 12 + i;
      o->perm[12 * (N + k) + i] = 24 * k + i;
    }
  return o;
}

void mpc_oracle_destroy(mpc_oracle *o) {
  if (!o) return;
  free(o->xref); free(o->x_f_applied); free(o->gait); free(o->S_gait);
  free(o->ML.p); free(o->ML.i); free(o->ML.x); free(o->P.p); free(o->P.i); free(o->P.x);
  free(o->i_off); free(o->NK_up); free(o->NK_low); free(o->Q); free(o->D); free(o->perm);
  oq_cleanup(o->work);
  free(o);
}

#define XREF(o, r, c) ((o)->xref[(r) * ((o)->n_steps + 1) + (c)])
#define Bm(o, r, c) ((o)->B[(r)*12 + (c)])

static int row_is_zero_i(const int *row, int n) {
  for (int i = 0; i < n; i++) if (row[i] != 0) return 0;
  return 1;
}

/* MPC::construct_gait, MPC.cpp:686-701 */
static void construct_gait(mpc_oracle *o, const double *fsteps) {
  int k = 0;
  /* (k < N_gait: with a table that has no all-zero row the reference reads and writes one row past its N_gait x 4 matrix;
   * the restatement stops at the table's end, as the HIP kernel does) */
  while (k < o->N_gait) {
    int zero = 1;
    for (int i = 0; i < 12; i++) if (fsteps[k * 12 + i] != 0.0) { zero = 0; break; }
    if (zero) break;
    for (int i = 0; i < 4; i++) o->gait[k * 4 + i] = (fsteps[k * 12 + i * 3] == 0.0) ? 0 : 1;
    k++;
  }
  if (k < o->N_gait) for (int i = 0; i < 4; i++) o->gait[k * 4 + i] = 0;
}

/* MPC::construct_S, MPC.cpp:665-681 */
static void construct_S(mpc_oracle *o) {
  int i = 0;
  while (!row_is_zero_i(&o->gait[i * 4], 4)) {
    for (int b = 0; b < 4; b++)
      for (int c = 0; c < 3; c++) o->S_gait[i * 12 + 3 * b + c] = 1 - o->gait[i * 4 + b];
    i++;
  }
}

/* I_inv = (R' gI R)^-1 with R = Rz(yaw); Eigen's 3x3 inverse() is the cofactor formula.
 * MPC.cpp:215-220 and :425-430 */
static void inertia_inverse(const mpc_oracle *o, double yaw, double I_inv[9]) {
  double c = cos(yaw), s = sin(yaw);
  double R[9] = {c, -s, 0.0, s, c, 0.0, 0.0, 0.0, 1.0};
  double T[9], M[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double a = 0; /* (R' gI)(i,j) */
      for (int k = 0; k < 3; k++) a += R[k * 3 + i] * o->gI[k * 3 + j];
      T[i * 3 + j] = a;
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double a = 0;
      for (int k = 0; k < 3; k++) a += T[i * 3 + k] * R[k * 3 + j];
      M[i * 3 + j] = a;
    }
  double c00 = M[4] * M[8] - M[5] * M[7], c10 = M[5] * M[6] - M[3] * M[8], c20 = M[3] * M[7] - M[4] * M[6];
  double det = M[0] * c00 + M[1] * c10 + M[2] * c20;
  double invdet = 1.0 / det;
  I_inv[0] = c00 * invdet;
  I_inv[1] = (M[2] * M[7] - M[1] * M[8]) * invdet;
  I_inv[2] = (M[1] * M[5] - M[2] * M[4]) * invdet;
  I_inv[3] = c10 * invdet;
  I_inv[4] = (M[0] * M[8] - M[2] * M[6]) * invdet;
  I_inv[5] = (M[2] * M[3] - M[0] * M[5]) * invdet;
  I_inv[6] = c20 * invdet;
  I_inv[7] = (M[1] * M[6] - M[0] * M[7]) * invdet;
  I_inv[8] = (M[0] * M[4] - M[1] * M[3]) * invdet;
}

/* B.block(9, 3i, 3, 3) = dt * (I_inv * getSkew(l)), MPC.cpp:225,440,654-658 */
static void fill_B_block(mpc_oracle *o, int foot, const double I_inv[9], const double l[3]) {
  double S[9] = {0.0, -l[2], l[1], l[2], 0.0, -l[0], -l[1], l[0], 0.0};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double a = 0;
      for (int k = 0; k < 3; k++) a += I_inv[r * 3 + k] * S[k * 3 + c];
      Bm(o, 9 + r, 3 * foot + c) = o->dt * a;
    }
}

/* MPC::create_ML, MPC.cpp:74-256 */
static void create_ML(mpc_oracle *o) {
  int N = o->n_steps;
  const int size_nz_ML = 5000;
  int need = 12 * N + 18 * (N - 1) + 48 * N + 12 * N + 36 * N + 16;
  int cap = need > size_nz_ML ? need : size_nz_ML;
  int *r_ML = (int *)calloc(cap, sizeof(int)), *c_ML = (int *)calloc(cap, sizeof(int));
  double *v_ML = (double *)calloc(cap, sizeof(double));
  int cpt = 0;
#define ADD(i_, j_, v_) do { r_ML[cpt] = (i_); c_ML[cpt] = (j_); v_ML[cpt] = (v_); cpt++; } while (0)
  for (int k = 0; k < 12 * N; k++) ADD(k, k, -1.0); /* :84-86 */
  for (int i = 0; i < 6; i++) o->A[i * 12 + i + 6] = o->dt; /* :89 */
  for (int k = 0; k < N - 1; k++) { /* :92-99 */
    for (int i = 0; i < 12; i++) ADD((k + 1) * 12 + i, k * 12 + i, 1.0);
    for (int j = 0; j < 6; j++) ADD((k + 1) * 12 + j, k * 12 + j + 6, o->dt);
  }
  double div_tmp = o->dt / o->mass; /* :102 */
  for (int k = 0; k < N; k++) {     /* :103-114 */
    for (int i = 0; i < 4; i++) {
      ADD(12 * k + 6, 12 * (N + k) + 0 + 3 * i, div_tmp);
      ADD(12 * k + 7, 12 * (N + k) + 1 + 3 * i, div_tmp);
      ADD(12 * k + 8, 12 * (N + k) + 2 + 3 * i, div_tmp);
    }
    for (int i = 0; i < 12; i++) {
      ADD(12 * k + 9, 12 * (N + k) + i, 8.0);
      ADD(12 * k + 10, 12 * (N + k) + i, 8.0);
      ADD(12 * k + 11, 12 * (N + k) + i, 8.0);
    }
  }
  for (int i = 0; i < 4; i++) { /* :115-122 */
    Bm(o, 6, 0 + 3 * i) = div_tmp;
    Bm(o, 7, 1 + 3 * i) = div_tmp;
    Bm(o, 8, 2 + 3 * i) = div_tmp;
    Bm(o, 9, i) = 8.0;
    Bm(o, 10, i) = 8.0;
    Bm(o, 11, i) = 8.0;
  }
  for (int i = 12 * N; i < 12 * N * 2; i++) ADD(i, i, 1.0); /* :125-127 */
  int offset_L = 12 * N * 2;                               /* :130-146 */
  for (int k = 0; k < N; k++) {
    int di = offset_L + 20 * k, dj = 12 * (N + k);
    for (int i = 0; i < 4; i++) {
      int dx = 5 * i, dy = 3 * i;
      int a[9] = {0, 1, 2, 3, 0, 1, 2, 3, 4};
      int b[9] = {0, 0, 1, 1, 2, 2, 2, 2, 2};
      double c[9] = {1.0, -1.0, 1.0, -1.0, -o->mu, -o->mu, -o->mu, -o->mu, -1};
      for (int j = 0; j < 9; j++) ADD(di + dx + a[j], dj + dy + b[j], c[j]);
    }
  }
#undef ADD
  o->cpt_ML = cpt;
  st_to_cc(cpt, r_ML, c_ML, v_ML, 12 * N * 2 + 20 * N, 12 * N * 2, &o->ML); /* :148-179 */
  free(r_ML); free(c_ML); free(v_ML);

  int i_x_tmp[12] = {6, 9, 10, 11, 7, 9, 10, 11, 8, 9, 10, 11}; /* :193-199 */
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < 12; i++) {
      o->i_x_B[12 * k + i] = i_x_tmp[i];
      o->i_y_B[12 * k + i] = (12 * k + i) / 4;
    }
  int i_start = 30 * N - 18; /* :201-208 */
  int i_data[12] = {0, 1, 2, 3, 7, 8, 9, 10, 14, 15, 16, 17};
  int i_foot[4] = {0 * 24, 1 * 24, 2 * 24, 3 * 24};
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < 12; i++) o->i_update_B[12 * k + i] = i_start + i_data[i] + i_foot[k];

  /* first fill of B: DEFAULT footholds, NO offset_CoM, all N steps (:213-232) */
  for (int k = 0; k < N; k++) {
    double I_inv[9];
    inertia_inverse(o, XREF(o, 5, k), I_inv);
    for (int i = 0; i < 4; i++) {
      double l[3] = {o->footholds[0 * 4 + i] - XREF(o, 0, k), o->footholds[1 * 4 + i] - XREF(o, 1, k),
                     o->footholds[2 * 4 + i] - XREF(o, 2, k)};
      fill_B_block(o, i, I_inv, l);
    }
    int i_iter = 24 * 4 * k;
    for (int j = 0; j < 12 * 4; j++) o->ML.x[o->i_update_B[j] + i_iter] = Bm(o, o->i_x_B[j], o->i_y_B[j]);
  }
  construct_S(o); /* :235 */
  int i_tmp1[3] = {3 + 4, 3 + 4, 6 + 4}; /* :237-253 */
  o->i_off[0] = 4;
  for (int k = 1; k < 12 * N; k++) o->i_off[k] = o->i_off[k - 1] + i_tmp1[(k - 1) % 3];
  for (int k = 0; k < 12 * N; k++) o->ML.x[o->i_off[k] + i_start] = o->S_gait[k];
}

/* shared by create_NK (:266-289) and update_NK (:472-486) */
static void fill_NK_up(mpc_oracle *o) {
  int N = o->n_steps, n12 = 12 * N;
  memset(o->NK_up, 0, 44 * N * sizeof(double));
  for (int k = 0; k < N; k++) o->NK_up[12 * k + 8] = -o->g[8];
  for (int i = 0; i < 12; i++) { /* NK_up.block(0,0,12,1) += A * (-x0) */
    double a = 0;
    for (int j = 0; j < 12; j++) a += o->A[i * 12 + j] * (-o->x0[j]);
    o->NK_up[i] += a;
  }
  /* NK_up.block(0,0,12N,1) += D * vec(xref[:,1:]) — dense product as written (O(N^2)) */
  double *tmp = (double *)calloc(n12, sizeof(double));
  for (int j = 0; j < n12; j++) { /* column-major accumulation, as Eigen's gemv */
    double xj = XREF(o, j % 12, 1 + j / 12);
    const double *Dj = &o->D[(size_t)j * n12];
    for (int i = 0; i < n12; i++) tmp[i] += Dj[i] * xj;
  }
  for (int i = 0; i < n12; i++) o->NK_up[i] += tmp[i];
  free(tmp);
}

/* MPC::create_NK, MPC.cpp:261-312 */
static void create_NK(mpc_oracle *o) {
  int N = o->n_steps, n12 = 12 * N;
  o->D = (double *)calloc((size_t)n12 * n12, sizeof(double)); /* column-major: D[col*n12 + row] */
  for (int i = 0; i < n12; i++) o->D[(size_t)i * n12 + i] = 1.0;
  for (int k = 0; k < N - 1; k++) {
    for (int i = 0; i < 12; i++) o->D[(size_t)(k * 12 + i) * n12 + (k + 1) * 12 + i] = -1.0;
    for (int i = 0; i < 6; i++) o->D[(size_t)(k * 12 + i + 6) * n12 + (k + 1) * 12 + i] = -o->dt;
  }
  fill_NK_up(o);
  memset(o->NK_low, 0, 44 * N * sizeof(double));
  for (int i = 0; i < 24 * N; i++) o->NK_low[i] = o->NK_up[i];
  for (int i = 0; i < 20 * N; i++) o->NK_low[24 * N + i] = -INFINITY; /* :293-300 */
  for (int k = 0; (4 + 5 * k) < (20 * N); k++) o->NK_low[24 * N + 4 + 5 * k] = -25.0;
}

/* MPC::create_weight_matrices, MPC.cpp:317-391 */
static void create_weight_matrices(mpc_oracle *o) {
  int N = o->n_steps, nst = 24 * N, cpt = 0;
  int *r_P = (int *)calloc(nst, sizeof(int)), *c_P = (int *)calloc(nst, sizeof(int));
  double *v_P = (double *)calloc(nst, sizeof(double));
  double w[12] = {2.0f, 2.0f, 20.0f, 0.25f, 0.25f, 10.0f, 0.2f, 0.2f, 0.2f, 0.0f, 0.0f, 0.3f}; /* :330 */
  for (int k = 0; k < N; k++)
    for (int i = 0; i < 12; i++) { r_P[cpt] = c_P[cpt] = 12 * k + i; v_P[cpt] = w[i]; cpt++; }
  for (int k = N; k < 2 * N; k++)
    for (int i = 0; i < 4; i++)
      for (int c = 0; c < 3; c++) { r_P[cpt] = c_P[cpt] = 12 * k + 3 * i + c; v_P[cpt] = 5e-5f; cpt++; } /* :346-348 */
  o->cpt_P = cpt;
  st_to_cc(cpt, r_P, c_P, v_P, 24 * N, 24 * N, &o->P);
  free(r_P); free(c_P); free(v_P);
  memset(o->Q, 0, 24 * N * sizeof(double)); /* :385 */
}

/* MPC::update_ML, MPC.cpp:418-464 */
static void update_ML(mpc_oracle *o, const double *fsteps) {
  int N = o->n_steps, j = 0, k_cum = 0;
  while (!row_is_zero_i(&o->gait[j * 4], 4)) {
    for (int k = k_cum; k < (k_cum + 1); k++) {
      double I_inv[9];
      inertia_inverse(o, XREF(o, 5, k), I_inv);
      for (int i = 0; i < 4; i++) {
        double l[3];
        for (int c = 0; c < 3; c++) l[c] = fsteps[j * 12 + 3 * i + c] - (XREF(o, c, k) + o->offset_CoM[c]);
        fill_B_block(o, i, I_inv, l);
      }
      int i_iter = 24 * 4 * k;
      for (int i = 0; i < 12 * 4; i++) o->ML.x[o->i_update_B[i] + i_iter] = Bm(o, o->i_x_B[i], o->i_y_B[i]);
    }
    k_cum++;
    j++;
  }
  construct_S(o);
  int i_start = 30 * N - 18;
  for (int k = 0; k < 12 * N; k++) o->ML.x[o->i_off[k] + i_start] = o->S_gait[k];
}

/* MPC::update_NK, MPC.cpp:469-496 */
static void update_NK(mpc_oracle *o) {
  int N = o->n_steps;
  fill_NK_up(o);
  for (int i = 0; i < 24 * N; i++) o->NK_low[i] = o->NK_up[i];
}

/* MPC::call_solver, MPC.cpp:501-564 (the warmxf shuffle :503-506 feeds the commented-out
 * osqp_warm_start_x :550 and has no effect) */
static int call_solver(mpc_oracle *o, int k) {
  if (k == 0) {
    o->settings.sigma = 1e-6;
    o->settings.eps_abs = 1e-6;
    o->settings.eps_rel = 1e-6;
    o->settings.eps_prim_inf = 1e-5;
    o->settings.eps_dual_inf = 1e-4;
    o->settings.alpha = 1.6;
    o->settings.adaptive_rho = 1;
    o->settings.adaptive_rho_interval = 200;
    o->settings.adaptive_rho_tolerance = 5.0;
    oq_cleanup(o->work); /* the reference leaks the old workspace; a fresh setup is what matters */
    o->work = oq_setup(&o->P, &o->ML, o->Q, o->NK_low, o->NK_up, &o->settings, o->perm);
    if (!o->work) return 1;
  } else {
    if (!o->work) return 2; /* reference would dereference an un-setup workspace */
    oq_update_A(o->work, o->ML.x);
    oq_update_bounds(o->work, o->NK_low, o->NK_up);
  }
  oq_solve(o->work); /* status ignored, MPC.cpp:558 */
  return 0;
}

/* MPC::retrieve_result, MPC.cpp:569-599 */
static void retrieve_result(mpc_oracle *o) {
  int N = o->n_steps;
  const double *sol = oq_solution_x(o->work);
  for (int i = 0; i < N; i++)
    for (int k = 0; k < 12; k++) {
      o->x_f_applied[k * N + i] = sol[k + 12 * i] + XREF(o, k, 1 + i);
      o->x_f_applied[(k + 12) * N + i] = sol[12 * (N + i) + k];
    }
  for (int k = 0; k < 12; k++) o->x_next[k] = sol[k];
}

/* MPC::run, MPC.cpp:626-649 */
int mpc_oracle_run(mpc_oracle *o, int num_iter, const double *xref_in, const double *fsteps_in) {
  int N = o->n_steps, rc;
  construct_gait(o, fsteps_in);
  memcpy(o->xref, xref_in, 12 * (N + 1) * sizeof(double));
  for (int i = 0; i < 12; i++) o->x0[i] = XREF(o, i, 0);
  if (num_iter == 0) {
    /* create_matrices(): a second num_iter==0 call re-creates everything (the reference
     * re-allocates without freeing); cpt counters restart here because the old arrays go. */
    free(o->ML.p); free(o->ML.i); free(o->ML.x); free(o->P.p); free(o->P.i); free(o->P.x); free(o->D);
    memset(&o->ML, 0, sizeof(o->ML)); memset(&o->P, 0, sizeof(o->P)); o->D = NULL;
    create_ML(o);
    create_NK(o);
    create_weight_matrices(o);
  } else {
    if (!o->ML.x) return 2;
    update_ML(o, fsteps_in);
    update_NK(o);
  }
  rc = call_solver(o, num_iter);
  if (rc) return rc;
  retrieve_result(o);
  return 0;
}

void mpc_oracle_get_latest_result(const mpc_oracle *o, double *out) {
  memcpy(out, o->x_f_applied, 24 * o->n_steps * sizeof(double));
}
void mpc_oracle_get_gait(const mpc_oracle *o, double *out) {
  for (int i = 0; i < o->N_gait * 4; i++) out[i] = (double)o->gait[i];
}
void mpc_oracle_get_Sgait(const mpc_oracle *o, double *out) {
  for (int i = 0; i < 12 * o->n_steps; i++) out[i] = (double)o->S_gait[i];
}
int mpc_oracle_restart(mpc_oracle *o, double rho) { return o->work ? oq_restart(o->work, rho) : -1; }
int mpc_oracle_iter(const mpc_oracle *o) { return o->work ? oq_info_iter(o->work) : -1; }
int mpc_oracle_status(const mpc_oracle *o) { return o->work ? oq_info_status(o->work) : OQ_UNSOLVED; }
double mpc_oracle_rho(const mpc_oracle *o) { return o->work ? oq_info_rho(o->work) : 0.0; }
void mpc_oracle_check_ratios(const mpc_oracle *o, double *out4) {
  if (o->work) oq_info_check_ratios(o->work, out4); else for (int i = 0; i < 4; i++) out4[i] = 0.0;
}
double mpc_oracle_pri_res(const mpc_oracle *o) { return o->work ? oq_info_pri_res(o->work) : 0.0; }
double mpc_oracle_dua_res(const mpc_oracle *o) { return o->work ? oq_info_dua_res(o->work) : 0.0; }
int mpc_oracle_nnz_ML(const mpc_oracle *o) { return o->ML.p ? o->ML.p[o->ML.n] : 0; }
void mpc_oracle_get_ML(const mpc_oracle *o, int *p, int *i, double *x) {
  int nnz = o->ML.p[o->ML.n];
  memcpy(p, o->ML.p, (o->ML.n + 1) * sizeof(int));
  memcpy(i, o->ML.i, nnz * sizeof(int));
  memcpy(x, o->ML.x, nnz * sizeof(double));
}
void mpc_oracle_get_P(const mpc_oracle *o, int *p, int *i, double *x) {
  int nnz = o->P.p[o->P.n];
  memcpy(p, o->P.p, (o->P.n + 1) * sizeof(int));
  memcpy(i, o->P.i, nnz * sizeof(int));
  memcpy(x, o->P.x, nnz * sizeof(double));
}
void mpc_oracle_get_bounds(const mpc_oracle *o, double *l, double *u) {
  memcpy(l, o->NK_low, 44 * o->n_steps * sizeof(double));
  memcpy(u, o->NK_up, 44 * o->n_steps * sizeof(double));
}
void mpc_oracle_get_solution(const mpc_oracle *o, double *x) {
  memcpy(x, oq_solution_x(o->work), 24 * o->n_steps * sizeof(double));
}
void mpc_oracle_get_iterates(const mpc_oracle *o, double *x, double *z, double *y) {
  int N = o->n_steps;
  if (x) memcpy(x, oq_iter_x(o->work), 24 * N * sizeof(double));
  if (z) memcpy(z, oq_iter_z(o->work), 44 * N * sizeof(double));
  if (y) memcpy(y, oq_iter_y(o->work), 44 * N * sizeof(double));
}

int mpc_oracle_run_batch(mpc_oracle **o, int B, const int *num_iter, const double *xref, const double *fsteps,
                         double *out, int threads) {
  int N = o[0]->n_steps, Ng = o[0]->N_gait, bad = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(| : bad)
  for (int b = 0; b < B; b++) {
    bad |= mpc_oracle_run(o[b], num_iter[b], xref + (size_t)b * 12 * (N + 1), fsteps + (size_t)b * Ng * 12);
    if (out) mpc_oracle_get_latest_result(o[b], out + (size_t)b * 24 * N);
  }
  return bad;
}
