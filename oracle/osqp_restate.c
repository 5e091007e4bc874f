/*
 * oracle/osqp_restate.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see osqp_restate.h).
 *
 * Function-by-function restatement of OSQP 0.6.x (third-party, absent from
 * /root/reference; PARITY UNPINNED).  Each function names the OSQP routine it
 * restates.  Arithmetic is kept in OSQP's order wherever the order is part of the
 * published algorithm (Ruiz passes, residual definitions, termination tests); the
 * linear-system solve is a banded Cholesky of the reduced KKT matrix, which is
 * mathematically the x-tilde/z-tilde OSQP's QDLDL solve returns.
 */
#include "osqp_restate.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define c_max(a, b) (((a) > (b)) ? (a) : (b))
#define c_min(a, b) (((a) < (b)) ? (a) : (b))
#define c_absval(x) (((x) < 0) ? -(x) : (x))

struct oq_work {
  int n, m;
  oq_csc P, A; /* scaled in place, like OSQP's work->data */
  double *q, *l, *u;
  oq_settings s;
  /* scaling */
  double c, cinv;
  double *D, *Dinv, *E, *Einv;
  /* rho */
  double *rho_vec, *rho_inv_vec;
  int *constr_type;
  /* iterates and work vectors (names as in OSQPWorkspace) */
  double *x, *y, *z, *xz_tilde, *x_prev, *z_prev;
  double *Ax, *Px, *Aty, *delta_y, *Atdelta_y, *delta_x, *Pdelta_x, *Adelta_x;
  double *D_temp, *D_temp_A, *E_temp;
  /* solution + info */
  double *sol_x, *sol_y;
  int iter, status, rho_updates;
  double pri_res, dua_res, rho_estimate, obj_val;
  /* test instrumentation (not OSQP): residual / tolerance of the last two full termination checks -- how close to the
   * threshold a termination decision was.  {primal, dual} of the last check, then of the one before it. */
  double chk_ratio[4];
  /* linear system: banded Cholesky of K = P + sigma I + A' diag(rho) A */
  int *perm;  /* perm[col] = position in the elimination order */
  int bw;     /* half bandwidth under perm */
  double *Kb; /* lower band, row-major: Kb[i*(bw+1) + (i-j)] = L(i,j) */
  double *rhs_p;
  int *Ar_p, *Ar_j, *Ar_k; /* CSR view of A's pattern: row ptr, col idx, index into A.x */
};

/* ---------- small vector/matrix helpers (osqp lin_alg.c) ---------- */

static double vec_norm_inf(const double *v, int l) {
  double max = 0.0, abs_v_i;
  for (int i = 0; i < l; i++) {
    abs_v_i = c_absval(v[i]);
    if (abs_v_i > max) max = abs_v_i;
  }
  return max;
}

static double vec_scaled_norm_inf(const double *S, const double *v, int l) {
  double max = 0.0, abs_Sv_i;
  for (int i = 0; i < l; i++) {
    abs_Sv_i = c_absval(S[i] * v[i]);
    if (abs_Sv_i > max) max = abs_Sv_i;
  }
  return max;
}

static double vec_mean(const double *a, int n) {
  double mean = 0.0;
  for (int i = 0; i < n; i++) mean += a[i];
  mean /= (double)n;
  return mean;
}

static double vec_prod(const double *a, const double *b, int n) {
  double prod = 0.0;
  for (int i = 0; i < n; i++) prod += a[i] * b[i];
  return prod;
}

/* y = A x (plus_eq 0), y += A x (1), y -= A x (-1) */
static void mat_vec(const oq_csc *A, const double *x, double *y, int plus_eq) {
  if (!plus_eq)
    for (int i = 0; i < A->m; i++) y[i] = 0;
  if (A->p[A->n] == 0) return;
  if (plus_eq == -1) {
    for (int j = 0; j < A->n; j++)
      for (int i = A->p[j]; i < A->p[j + 1]; i++) y[A->i[i]] -= A->x[i] * x[j];
  } else {
    for (int j = 0; j < A->n; j++)
      for (int i = A->p[j]; i < A->p[j + 1]; i++) y[A->i[i]] += A->x[i] * x[j];
  }
}

/* y = A' x */
static void mat_tpose_vec(const oq_csc *A, const double *x, double *y, int plus_eq, int skip_diag) {
  if (!plus_eq)
    for (int i = 0; i < A->n; i++) y[i] = 0;
  if (A->p[A->n] == 0) return;
  if (plus_eq == -1) {
    for (int j = 0; j < A->n; j++)
      for (int k = A->p[j]; k < A->p[j + 1]; k++) {
        if (skip_diag && A->i[k] == j) continue;
        y[j] -= A->x[k] * x[A->i[k]];
      }
  } else {
    for (int j = 0; j < A->n; j++)
      for (int k = A->p[j]; k < A->p[j + 1]; k++) {
        if (skip_diag && A->i[k] == j) continue;
        y[j] += A->x[k] * x[A->i[k]];
      }
  }
}

static void mat_premult_diag(oq_csc *A, const double *d) {
  for (int j = 0; j < A->n; j++)
    for (int i = A->p[j]; i < A->p[j + 1]; i++) A->x[i] *= d[A->i[i]];
}

static void mat_postmult_diag(oq_csc *A, const double *d) {
  for (int j = 0; j < A->n; j++)
    for (int i = A->p[j]; i < A->p[j + 1]; i++) A->x[i] *= d[j];
}

static void mat_mult_scalar(oq_csc *A, double sc) {
  int nnzA = A->p[A->n];
  for (int i = 0; i < nnzA; i++) A->x[i] *= sc;
}

static void mat_inf_norm_cols(const oq_csc *M, double *E) {
  for (int j = 0; j < M->n; j++) E[j] = 0.;
  for (int j = 0; j < M->n; j++)
    for (int ptr = M->p[j]; ptr < M->p[j + 1]; ptr++) E[j] = c_max(c_absval(M->x[ptr]), E[j]);
}

static void mat_inf_norm_rows(const oq_csc *M, double *E) {
  for (int j = 0; j < M->m; j++) E[j] = 0.;
  for (int j = 0; j < M->n; j++)
    for (int ptr = M->p[j]; ptr < M->p[j + 1]; ptr++) {
      int i = M->i[ptr];
      E[i] = c_max(c_absval(M->x[ptr]), E[i]);
    }
}

static void mat_inf_norm_cols_sym_triu(const oq_csc *M, double *E) {
  for (int j = 0; j < M->n; j++) E[j] = 0.;
  for (int j = 0; j < M->n; j++)
    for (int ptr = M->p[j]; ptr < M->p[j + 1]; ptr++) {
      int i = M->i[ptr];
      double abs_x = c_absval(M->x[ptr]);
      E[j] = c_max(abs_x, E[j]);
      if (i != j) E[i] = c_max(abs_x, E[i]);
    }
}

/* ---------- scaling (osqp scaling.c) ---------- */

static void limit_scaling(double *D, int n) {
  for (int i = 0; i < n; i++) {
    D[i] = D[i] < OQ_MIN_SCALING ? 1.0 : D[i];
    D[i] = D[i] > OQ_MAX_SCALING ? OQ_MAX_SCALING : D[i];
  }
}

static void compute_inf_norm_cols_KKT(const oq_csc *P, const oq_csc *A, double *D, double *D_temp_A, double *E,
                                      int n) {
  mat_inf_norm_cols_sym_triu(P, D);
  mat_inf_norm_cols(A, D_temp_A);
  for (int i = 0; i < n; i++) D[i] = c_max(D[i], D_temp_A[i]);
  mat_inf_norm_rows(A, E);
}

/* osqp scale_data() */
static void scale_data(oq_work *w) {
  int n = w->n, m = w->m;
  double c_temp, inf_norm_q;

  w->c = 1.0;
  for (int i = 0; i < n; i++) w->D[i] = w->Dinv[i] = 1.;
  for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.;

  for (int it = 0; it < w->s.scaling; it++) {
    compute_inf_norm_cols_KKT(&w->P, &w->A, w->D_temp, w->D_temp_A, w->E_temp, n);
    limit_scaling(w->D_temp, n);
    limit_scaling(w->E_temp, m);
    for (int i = 0; i < n; i++) w->D_temp[i] = sqrt(w->D_temp[i]);
    for (int i = 0; i < m; i++) w->E_temp[i] = sqrt(w->E_temp[i]);
    for (int i = 0; i < n; i++) w->D_temp[i] = (double)1.0 / w->D_temp[i];
    for (int i = 0; i < m; i++) w->E_temp[i] = (double)1.0 / w->E_temp[i];

    mat_premult_diag(&w->P, w->D_temp);
    mat_postmult_diag(&w->P, w->D_temp);
    mat_premult_diag(&w->A, w->E_temp);
    mat_postmult_diag(&w->A, w->D_temp);
    for (int i = 0; i < n; i++) w->q[i] = w->D_temp[i] * w->q[i];
    for (int i = 0; i < n; i++) w->D[i] = w->D[i] * w->D_temp[i];
    for (int i = 0; i < m; i++) w->E[i] = w->E[i] * w->E_temp[i];

    /* cost normalisation step */
    mat_inf_norm_cols_sym_triu(&w->P, w->D_temp);
    c_temp = vec_mean(w->D_temp, n);
    inf_norm_q = vec_norm_inf(w->q, n);
    limit_scaling(&inf_norm_q, 1);
    c_temp = c_max(c_temp, inf_norm_q);
    limit_scaling(&c_temp, 1);
    c_temp = 1. / c_temp;
    mat_mult_scalar(&w->P, c_temp);
    for (int i = 0; i < n; i++) w->q[i] *= c_temp;
    w->c *= c_temp;
  }

  w->cinv = 1. / w->c;
  for (int i = 0; i < n; i++) w->Dinv[i] = (double)1.0 / w->D[i];
  for (int i = 0; i < m; i++) w->Einv[i] = (double)1.0 / w->E[i];
  for (int i = 0; i < m; i++) w->l[i] = w->E[i] * w->l[i];
  for (int i = 0; i < m; i++) w->u[i] = w->E[i] * w->u[i];
}

/* osqp unscale_data() */
static void unscale_data(oq_work *w) {
  int n = w->n, m = w->m;
  mat_mult_scalar(&w->P, w->cinv);
  mat_premult_diag(&w->P, w->Dinv);
  mat_postmult_diag(&w->P, w->Dinv);
  for (int i = 0; i < n; i++) w->q[i] *= w->cinv;
  for (int i = 0; i < n; i++) w->q[i] = w->Dinv[i] * w->q[i];
  mat_premult_diag(&w->A, w->Einv);
  mat_postmult_diag(&w->A, w->Dinv);
  for (int i = 0; i < m; i++) w->l[i] = w->Einv[i] * w->l[i];
  for (int i = 0; i < m; i++) w->u[i] = w->Einv[i] * w->u[i];
}

/* ---------- linear system: banded Cholesky of the reduced KKT matrix ---------- */

static void linsys_pattern(oq_work *w) {
  int n = w->n, m = w->m, nnz = w->A.p[n];
  /* CSR view of A */
  w->Ar_p = (int *)calloc(m + 1, sizeof(int));
  w->Ar_j = (int *)malloc((nnz ? nnz : 1) * sizeof(int));
  w->Ar_k = (int *)malloc((nnz ? nnz : 1) * sizeof(int));
  for (int k = 0; k < nnz; k++) w->Ar_p[w->A.i[k] + 1]++;
  for (int i = 0; i < m; i++) w->Ar_p[i + 1] += w->Ar_p[i];
  int *fill = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
  for (int j = 0; j < n; j++)
    for (int k = w->A.p[j]; k < w->A.p[j + 1]; k++) {
      int r = w->A.i[k], pos = w->Ar_p[r] + fill[r]++;
      w->Ar_j[pos] = j;
      w->Ar_k[pos] = k;
    }
  free(fill);
  /* bandwidth under perm */
  int bw = 0;
  for (int r = 0; r < m; r++)
    for (int a = w->Ar_p[r]; a < w->Ar_p[r + 1]; a++)
      for (int b = w->Ar_p[r]; b < w->Ar_p[r + 1]; b++) {
        int d = w->perm[w->Ar_j[a]] - w->perm[w->Ar_j[b]];
        if (d > bw) bw = d;
      }
  for (int j = 0; j < n; j++)
    for (int k = w->P.p[j]; k < w->P.p[j + 1]; k++) {
      int d = abs(w->perm[w->P.i[k]] - w->perm[j]);
      if (d > bw) bw = d;
    }
  w->bw = bw;
  w->Kb = (double *)malloc((size_t)n * (bw + 1) * sizeof(double));
  w->rhs_p = (double *)malloc(n * sizeof(double));
}

/* (re)factorise: K = P + sigma I + A' diag(rho_vec) A, in band storage under perm */
static int linsys_factor(oq_work *w) {
  int n = w->n, m = w->m, bw = w->bw, ld = bw + 1;
  double *K = w->Kb;
  memset(K, 0, (size_t)n * ld * sizeof(double));
  for (int j = 0; j < n; j++) {
    for (int k = w->P.p[j]; k < w->P.p[j + 1]; k++) {
      int pi = w->perm[w->P.i[k]], pj = w->perm[j];
      int hi = c_max(pi, pj), lo = c_min(pi, pj);
      K[hi * ld + (hi - lo)] += w->P.x[k];
    }
    K[w->perm[j] * ld] += w->s.sigma;
  }
  for (int r = 0; r < m; r++) {
    double rho = w->rho_vec[r];
    for (int a = w->Ar_p[r]; a < w->Ar_p[r + 1]; a++) {
      int pa = w->perm[w->Ar_j[a]];
      double va = rho * w->A.x[w->Ar_k[a]];
      for (int b = w->Ar_p[r]; b < w->Ar_p[r + 1]; b++) {
        int pb = w->perm[w->Ar_j[b]];
        if (pb > pa) continue;
        K[pa * ld + (pa - pb)] += va * w->A.x[w->Ar_k[b]];
      }
    }
  }
  /* banded Cholesky, in place: K -> L */
  for (int i = 0; i < n; i++) {
    int j0 = c_max(0, i - bw);
    for (int j = j0; j <= i; j++) {
      double sum = K[i * ld + (i - j)];
      int k0 = c_max(j0, j - bw);
      for (int k = k0; k < j; k++) sum -= K[i * ld + (i - k)] * K[j * ld + (j - k)];
      if (j == i) {
        if (!(sum > 0.0)) return 1;
        K[i * ld] = sqrt(sum);
      } else {
        K[i * ld + (i - j)] = sum / K[j * ld];
      }
    }
  }
  return 0;
}

/* solve K sol = b (b, sol length n, natural ordering) */
static void linsys_solve_reduced(oq_work *w, const double *b, double *sol) {
  int n = w->n, bw = w->bw, ld = bw + 1;
  double *K = w->Kb, *t = w->rhs_p;
  for (int j = 0; j < n; j++) t[w->perm[j]] = b[j];
  for (int i = 0; i < n; i++) {
    double sum = t[i];
    for (int k = c_max(0, i - bw); k < i; k++) sum -= K[i * ld + (i - k)] * t[k];
    t[i] = sum / K[i * ld];
  }
  for (int i = n - 1; i >= 0; i--) {
    double sum = t[i];
    int kmax = c_min(n - 1, i + bw);
    for (int k = i + 1; k <= kmax; k++) sum -= K[k * ld + (k - i)] * t[k];
    t[i] = sum / K[i * ld];
  }
  for (int j = 0; j < n; j++) sol[j] = t[w->perm[j]];
}

/*
 * osqp solve_linsys_qdldl(): on entry b = [sigma x_prev - q ; z_prev - y/rho];
 * on exit b = [x_tilde ; z_tilde].  From the KKT system
 *   [P + sigma I, A'; A, -diag(1/rho)] [x_tilde; nu] = b
 * eliminate nu = rho (A x_tilde - b_bot):  K x_tilde = b_top + A' (rho b_bot),
 * and z_tilde = b_bot + nu / rho = A x_tilde.
 */
static void linsys_solve(oq_work *w, double *b) {
  int n = w->n, m = w->m;
  double *tmp_m = w->Adelta_x; /* free work vector here */
  double *tmp_n = w->Pdelta_x;
  for (int i = 0; i < m; i++) tmp_m[i] = w->rho_vec[i] * b[n + i];
  mat_tpose_vec(&w->A, tmp_m, tmp_n, 0, 0);
  for (int i = 0; i < n; i++) tmp_n[i] = b[i] + tmp_n[i];
  linsys_solve_reduced(w, tmp_n, b);
  mat_vec(&w->A, b, b + n, 0);
}

/* ---------- rho (osqp auxil.c) ---------- */

static void set_rho_vec(oq_work *w) {
  w->s.rho = c_min(c_max(w->s.rho, OQ_RHO_MIN), OQ_RHO_MAX);
  for (int i = 0; i < w->m; i++) {
    if ((w->l[i] < -OQ_INFTY * OQ_MIN_SCALING) && (w->u[i] > OQ_INFTY * OQ_MIN_SCALING)) {
      w->constr_type[i] = -1;
      w->rho_vec[i] = OQ_RHO_MIN;
    } else if (w->u[i] - w->l[i] < OQ_RHO_TOL) {
      w->constr_type[i] = 1;
      w->rho_vec[i] = OQ_RHO_EQ_OVER_RHO_INEQ * w->s.rho;
    } else {
      w->constr_type[i] = 0;
      w->rho_vec[i] = w->s.rho;
    }
    w->rho_inv_vec[i] = 1. / w->rho_vec[i];
  }
}

static int update_rho_vec(oq_work *w) {
  int constr_type_changed = 0;
  for (int i = 0; i < w->m; i++) {
    if ((w->l[i] < -OQ_INFTY * OQ_MIN_SCALING) && (w->u[i] > OQ_INFTY * OQ_MIN_SCALING)) {
      if (w->constr_type[i] != -1) {
        w->constr_type[i] = -1;
        w->rho_vec[i] = OQ_RHO_MIN;
        w->rho_inv_vec[i] = 1. / OQ_RHO_MIN;
        constr_type_changed = 1;
      }
    } else if (w->u[i] - w->l[i] < OQ_RHO_TOL) {
      if (w->constr_type[i] != 1) {
        w->constr_type[i] = 1;
        w->rho_vec[i] = OQ_RHO_EQ_OVER_RHO_INEQ * w->s.rho;
        w->rho_inv_vec[i] = 1. / w->rho_vec[i];
        constr_type_changed = 1;
      }
    } else {
      if (w->constr_type[i] != 0) {
        w->constr_type[i] = 0;
        w->rho_vec[i] = w->s.rho;
        w->rho_inv_vec[i] = 1. / w->s.rho;
        constr_type_changed = 1;
      }
    }
  }
  if (constr_type_changed) return linsys_factor(w);
  return 0;
}

/* osqp_update_rho() */
static int update_rho(oq_work *w, double rho_new) {
  if (rho_new <= 0) return 1;
  w->s.rho = c_min(c_max(rho_new, OQ_RHO_MIN), OQ_RHO_MAX);
  for (int i = 0; i < w->m; i++) {
    if (w->constr_type[i] == 0) {
      w->rho_vec[i] = w->s.rho;
      w->rho_inv_vec[i] = 1. / w->s.rho;
    } else if (w->constr_type[i] == 1) {
      w->rho_vec[i] = OQ_RHO_EQ_OVER_RHO_INEQ * w->s.rho;
      w->rho_inv_vec[i] = 1. / w->rho_vec[i];
    }
  }
  return linsys_factor(w);
}

static double compute_rho_estimate(oq_work *w) {
  int n = w->n, m = w->m;
  double pri_res, dua_res, pri_res_norm, dua_res_norm, temp_res_norm, rho_estimate;
  pri_res = vec_norm_inf(w->z_prev, m); /* residual vectors left by compute_pri_res/dua_res */
  dua_res = vec_norm_inf(w->x_prev, n);
  pri_res_norm = vec_norm_inf(w->z, m);
  temp_res_norm = vec_norm_inf(w->Ax, m);
  pri_res_norm = c_max(pri_res_norm, temp_res_norm);
  pri_res /= (pri_res_norm + 1e-10);
  dua_res_norm = vec_norm_inf(w->q, n);
  temp_res_norm = vec_norm_inf(w->Aty, n);
  dua_res_norm = c_max(dua_res_norm, temp_res_norm);
  temp_res_norm = vec_norm_inf(w->Px, n);
  dua_res_norm = c_max(dua_res_norm, temp_res_norm);
  dua_res /= (dua_res_norm + 1e-10);
  rho_estimate = w->s.rho * sqrt(pri_res / (dua_res + 1e-10));
  rho_estimate = c_min(c_max(rho_estimate, OQ_RHO_MIN), OQ_RHO_MAX);
  return rho_estimate;
}

static int adapt_rho(oq_work *w) {
  int exitflag = 0;
  double rho_new = compute_rho_estimate(w);
  w->rho_estimate = rho_new;
  if ((rho_new > w->s.rho * w->s.adaptive_rho_tolerance) || (rho_new < w->s.rho / w->s.adaptive_rho_tolerance)) {
    exitflag = update_rho(w, rho_new);
    w->rho_updates += 1;
  }
  return exitflag;
}

/* ---------- ADMM steps (osqp auxil.c) ---------- */

static void cold_start(oq_work *w) {
  memset(w->x, 0, w->n * sizeof(double));
  memset(w->z, 0, w->m * sizeof(double));
  memset(w->y, 0, w->m * sizeof(double));
}

static void update_xz_tilde(oq_work *w) {
  int n = w->n, m = w->m;
  for (int i = 0; i < n; i++) w->xz_tilde[i] = w->s.sigma * w->x_prev[i] - w->q[i];
  for (int i = 0; i < m; i++) w->xz_tilde[i + n] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
  linsys_solve(w, w->xz_tilde);
}

static void update_x(oq_work *w) {
  for (int i = 0; i < w->n; i++)
    w->x[i] = w->s.alpha * w->xz_tilde[i] + ((double)1.0 - w->s.alpha) * w->x_prev[i];
  for (int i = 0; i < w->n; i++) w->delta_x[i] = w->x[i] - w->x_prev[i];
}

static void update_z(oq_work *w) {
  int n = w->n;
  for (int i = 0; i < w->m; i++)
    w->z[i] = w->s.alpha * w->xz_tilde[i + n] + ((double)1.0 - w->s.alpha) * w->z_prev[i] +
              w->rho_inv_vec[i] * w->y[i];
  for (int i = 0; i < w->m; i++) w->z[i] = c_min(c_max(w->z[i], w->l[i]), w->u[i]); /* project() */
}

static void update_y(oq_work *w) {
  int n = w->n;
  for (int i = 0; i < w->m; i++) {
    w->delta_y[i] =
        w->rho_vec[i] * (w->s.alpha * w->xz_tilde[i + n] + ((double)1.0 - w->s.alpha) * w->z_prev[i] - w->z[i]);
    w->y[i] += w->delta_y[i];
  }
}

static double compute_pri_res(oq_work *w, const double *x, const double *z) {
  mat_vec(&w->A, x, w->Ax, 0);
  for (int i = 0; i < w->m; i++) w->z_prev[i] = w->Ax[i] - z[i]; /* z_prev is the work vector */
  if (w->s.scaling && !w->s.scaled_termination) return vec_scaled_norm_inf(w->Einv, w->z_prev, w->m);
  return vec_norm_inf(w->z_prev, w->m);
}

static double compute_pri_tol(oq_work *w, double eps_abs, double eps_rel) {
  double max_rel_eps, temp_rel_eps;
  if (w->s.scaling && !w->s.scaled_termination) {
    max_rel_eps = vec_scaled_norm_inf(w->Einv, w->z, w->m);
    temp_rel_eps = vec_scaled_norm_inf(w->Einv, w->Ax, w->m);
    max_rel_eps = c_max(max_rel_eps, temp_rel_eps);
  } else {
    max_rel_eps = vec_norm_inf(w->z, w->m);
    temp_rel_eps = vec_norm_inf(w->Ax, w->m);
    max_rel_eps = c_max(max_rel_eps, temp_rel_eps);
  }
  return eps_abs + eps_rel * max_rel_eps;
}

static double compute_dua_res(oq_work *w, const double *x, const double *y) {
  int n = w->n;
  memcpy(w->x_prev, w->q, n * sizeof(double)); /* x_prev is the work vector */
  mat_vec(&w->P, x, w->Px, 0);
  mat_tpose_vec(&w->P, x, w->Px, 1, 1);
  for (int i = 0; i < n; i++) w->x_prev[i] = w->x_prev[i] + w->Px[i];
  if (w->m > 0) {
    mat_tpose_vec(&w->A, y, w->Aty, 0, 0);
    for (int i = 0; i < n; i++) w->x_prev[i] = w->x_prev[i] + w->Aty[i];
  }
  if (w->s.scaling && !w->s.scaled_termination) return w->cinv * vec_scaled_norm_inf(w->Dinv, w->x_prev, n);
  return vec_norm_inf(w->x_prev, n);
}

static double compute_dua_tol(oq_work *w, double eps_abs, double eps_rel) {
  double max_rel_eps, temp_rel_eps;
  int n = w->n;
  if (w->s.scaling && !w->s.scaled_termination) {
    max_rel_eps = vec_scaled_norm_inf(w->Dinv, w->q, n);
    temp_rel_eps = vec_scaled_norm_inf(w->Dinv, w->Aty, n);
    max_rel_eps = c_max(max_rel_eps, temp_rel_eps);
    temp_rel_eps = vec_scaled_norm_inf(w->Dinv, w->Px, n);
    max_rel_eps = c_max(max_rel_eps, temp_rel_eps);
    max_rel_eps *= w->cinv;
  } else {
    max_rel_eps = vec_norm_inf(w->q, n);
    temp_rel_eps = vec_norm_inf(w->Aty, n);
    max_rel_eps = c_max(max_rel_eps, temp_rel_eps);
    temp_rel_eps = vec_norm_inf(w->Px, n);
    max_rel_eps = c_max(max_rel_eps, temp_rel_eps);
  }
  return eps_abs + eps_rel * max_rel_eps;
}

static int is_primal_infeasible(oq_work *w, double eps_prim_inf) {
  int m = w->m, n = w->n;
  double norm_delta_y, ineq_lhs = 0.0;
  for (int i = 0; i < m; i++) {
    if (w->u[i] > OQ_INFTY * OQ_MIN_SCALING) {
      if (w->l[i] < -OQ_INFTY * OQ_MIN_SCALING) {
        w->delta_y[i] = 0.0;
      } else {
        w->delta_y[i] = c_min(w->delta_y[i], 0.0);
      }
    } else if (w->l[i] < -OQ_INFTY * OQ_MIN_SCALING) {
      w->delta_y[i] = c_max(w->delta_y[i], 0.0);
    }
  }
  if (w->s.scaling && !w->s.scaled_termination) {
    for (int i = 0; i < m; i++) w->Adelta_x[i] = w->E[i] * w->delta_y[i];
    norm_delta_y = vec_norm_inf(w->Adelta_x, m);
  } else {
    norm_delta_y = vec_norm_inf(w->delta_y, m);
  }
  if (norm_delta_y > eps_prim_inf) {
    for (int i = 0; i < m; i++)
      ineq_lhs += w->u[i] * c_max(w->delta_y[i], 0) + w->l[i] * c_min(w->delta_y[i], 0);
    if (ineq_lhs < -eps_prim_inf * norm_delta_y) {
      mat_tpose_vec(&w->A, w->delta_y, w->Atdelta_y, 0, 0);
      if (w->s.scaling && !w->s.scaled_termination)
        for (int i = 0; i < n; i++) w->Atdelta_y[i] = w->Dinv[i] * w->Atdelta_y[i];
      return vec_norm_inf(w->Atdelta_y, n) < eps_prim_inf * norm_delta_y;
    }
  }
  return 0;
}

static int is_dual_infeasible(oq_work *w, double eps_dual_inf) {
  int n = w->n, m = w->m;
  double norm_delta_x, cost_scaling;
  if (w->s.scaling && !w->s.scaled_termination) {
    norm_delta_x = vec_scaled_norm_inf(w->D, w->delta_x, n);
    cost_scaling = w->c;
  } else {
    norm_delta_x = vec_norm_inf(w->delta_x, n);
    cost_scaling = 1.0;
  }
  if (norm_delta_x > eps_dual_inf) {
    if (vec_prod(w->q, w->delta_x, n) < -cost_scaling * eps_dual_inf * norm_delta_x) {
      mat_vec(&w->P, w->delta_x, w->Pdelta_x, 0);
      mat_tpose_vec(&w->P, w->delta_x, w->Pdelta_x, 1, 1);
      if (w->s.scaling && !w->s.scaled_termination)
        for (int i = 0; i < n; i++) w->Pdelta_x[i] = w->Dinv[i] * w->Pdelta_x[i];
      if (vec_norm_inf(w->Pdelta_x, n) < cost_scaling * eps_dual_inf * norm_delta_x) {
        mat_vec(&w->A, w->delta_x, w->Adelta_x, 0);
        if (w->s.scaling && !w->s.scaled_termination)
          for (int i = 0; i < m; i++) w->Adelta_x[i] = w->Einv[i] * w->Adelta_x[i];
        for (int i = 0; i < m; i++) {
          if (((w->u[i] < OQ_INFTY * OQ_MIN_SCALING) && (w->Adelta_x[i] > eps_dual_inf * norm_delta_x)) ||
              ((w->l[i] > -OQ_INFTY * OQ_MIN_SCALING) && (w->Adelta_x[i] < -eps_dual_inf * norm_delta_x))) {
            return 0;
          }
        }
        return 1;
      }
    }
  }
  return 0;
}

static void update_info(oq_work *w, int iter) {
  w->iter = iter;
  if (w->m == 0)
    w->pri_res = 0.;
  else
    w->pri_res = compute_pri_res(w, w->x, w->z);
  w->dua_res = compute_dua_res(w, w->x, w->y);
}

static int check_termination(oq_work *w, int approximate) {
  double eps_prim = 1.0, eps_dual, eps_prim_inf, eps_dual_inf, eps_abs, eps_rel;
  int exitflag = 0, prim_res_check = 0, dual_res_check = 0, prim_inf_check = 0, dual_inf_check = 0;

  eps_abs = w->s.eps_abs;
  eps_rel = w->s.eps_rel;
  eps_prim_inf = w->s.eps_prim_inf;
  eps_dual_inf = w->s.eps_dual_inf;

  if ((w->pri_res > OQ_INFTY) || (w->dua_res > OQ_INFTY)) {
    w->status = OQ_NON_CVX;
    w->obj_val = NAN;
    return 1;
  }
  if (approximate) {
    eps_abs *= 10;
    eps_rel *= 10;
    eps_prim_inf *= 10;
    eps_dual_inf *= 10;
  }
  if (w->m == 0) {
    prim_res_check = 1;
  } else {
    eps_prim = compute_pri_tol(w, eps_abs, eps_rel);
    if (w->pri_res < eps_prim) {
      prim_res_check = 1;
    } else {
      prim_inf_check = is_primal_infeasible(w, eps_prim_inf);
    }
  }
  eps_dual = compute_dua_tol(w, eps_abs, eps_rel);
  if (!approximate) {
    w->chk_ratio[2] = w->chk_ratio[0];
    w->chk_ratio[3] = w->chk_ratio[1];
    w->chk_ratio[0] = (w->m == 0) ? 0.0 : w->pri_res / eps_prim;
    w->chk_ratio[1] = w->dua_res / eps_dual;
  }
  if (w->dua_res < eps_dual) {
    dual_res_check = 1;
  } else {
    dual_inf_check = is_dual_infeasible(w, eps_dual_inf);
  }
  if (prim_res_check && dual_res_check) {
    w->status = approximate ? OQ_SOLVED_INACCURATE : OQ_SOLVED;
    exitflag = 1;
  } else if (prim_inf_check) {
    w->status = approximate ? OQ_PRIMAL_INFEASIBLE_INACCURATE : OQ_PRIMAL_INFEASIBLE;
    if (w->s.scaling && !w->s.scaled_termination)
      for (int i = 0; i < w->m; i++) w->delta_y[i] = w->E[i] * w->delta_y[i];
    w->obj_val = OQ_INFTY;
    exitflag = 1;
  } else if (dual_inf_check) {
    w->status = approximate ? OQ_DUAL_INFEASIBLE_INACCURATE : OQ_DUAL_INFEASIBLE;
    if (w->s.scaling && !w->s.scaled_termination)
      for (int i = 0; i < w->n; i++) w->delta_x[i] = w->D[i] * w->delta_x[i];
    w->obj_val = -OQ_INFTY;
    exitflag = 1;
  }
  return exitflag;
}

static int has_solution(const oq_work *w) {
  return ((w->status != OQ_PRIMAL_INFEASIBLE) && (w->status != OQ_PRIMAL_INFEASIBLE_INACCURATE) &&
          (w->status != OQ_DUAL_INFEASIBLE) && (w->status != OQ_DUAL_INFEASIBLE_INACCURATE) &&
          (w->status != OQ_NON_CVX));
}

static void store_solution(oq_work *w) {
  if (has_solution(w)) {
    memcpy(w->sol_x, w->x, w->n * sizeof(double));
    memcpy(w->sol_y, w->y, w->m * sizeof(double));
    if (w->s.scaling) { /* unscale_solution() */
      for (int i = 0; i < w->n; i++) w->sol_x[i] = w->D[i] * w->sol_x[i];
      for (int i = 0; i < w->m; i++) w->sol_y[i] = w->E[i] * w->sol_y[i];
      for (int i = 0; i < w->m; i++) w->sol_y[i] *= w->cinv;
    }
  } else {
    for (int i = 0; i < w->n; i++) w->sol_x[i] = NAN;
    for (int i = 0; i < w->m; i++) w->sol_y[i] = NAN;
    cold_start(w);
  }
}

static void reset_info(oq_work *w) {
  w->status = OQ_UNSOLVED;
  w->rho_updates = 0;
}

/* ---------- public API ---------- */

void oq_set_default_settings(oq_settings *s) {
  s->rho = 0.1;
  s->sigma = 1E-06;
  s->scaling = 10;
  s->adaptive_rho = 1;
  s->adaptive_rho_interval = 0;
  s->adaptive_rho_tolerance = 5;
  s->max_iter = 4000;
  s->eps_abs = 1E-3;
  s->eps_rel = 1E-3;
  s->eps_prim_inf = 1E-4;
  s->eps_dual_inf = 1E-4;
  s->alpha = 1.6;
  s->scaled_termination = 0;
  s->check_termination = 25;
  s->warm_start = 1;
}

static void csc_copy(oq_csc *dst, const oq_csc *src) {
  int nnz = src->p[src->n];
  dst->m = src->m;
  dst->n = src->n;
  dst->p = (int *)malloc((src->n + 1) * sizeof(int));
  dst->i = (int *)malloc((nnz ? nnz : 1) * sizeof(int));
  dst->x = (double *)malloc((nnz ? nnz : 1) * sizeof(double));
  memcpy(dst->p, src->p, (src->n + 1) * sizeof(int));
  memcpy(dst->i, src->i, nnz * sizeof(int));
  memcpy(dst->x, src->x, nnz * sizeof(double));
}

static double *vec_copy(const double *a, int n) {
  double *b = (double *)malloc((n ? n : 1) * sizeof(double));
  memcpy(b, a, n * sizeof(double));
  return b;
}

static double *vec_zero(int n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

oq_work *oq_setup(const oq_csc *P, const oq_csc *A, const double *q, const double *l, const double *u,
                  const oq_settings *settings, const int *perm) {
  int n = A->n, m = A->m;
  oq_work *w = (oq_work *)calloc(1, sizeof(oq_work));
  w->n = n;
  w->m = m;
  csc_copy(&w->P, P);
  csc_copy(&w->A, A);
  w->q = vec_copy(q, n);
  w->l = vec_copy(l, m);
  w->u = vec_copy(u, m);
  w->s = *settings;
  w->D = vec_zero(n);
  w->Dinv = vec_zero(n);
  w->E = vec_zero(m);
  w->Einv = vec_zero(m);
  w->rho_vec = vec_zero(m);
  w->rho_inv_vec = vec_zero(m);
  w->constr_type = (int *)calloc(m ? m : 1, sizeof(int));
  w->x = vec_zero(n);
  w->z = vec_zero(m);
  w->y = vec_zero(m);
  w->xz_tilde = vec_zero(n + m);
  w->x_prev = vec_zero(n);
  w->z_prev = vec_zero(m);
  w->Ax = vec_zero(m);
  w->Px = vec_zero(n);
  w->Aty = vec_zero(n);
  w->delta_y = vec_zero(m);
  w->Atdelta_y = vec_zero(n);
  w->delta_x = vec_zero(n);
  w->Pdelta_x = vec_zero(n);
  w->Adelta_x = vec_zero(m);
  w->D_temp = vec_zero(n);
  w->D_temp_A = vec_zero(n);
  w->E_temp = vec_zero(m);
  w->sol_x = vec_zero(n);
  w->sol_y = vec_zero(m);
  w->perm = (int *)malloc(n * sizeof(int));
  for (int j = 0; j < n; j++) w->perm[j] = perm ? perm[j] : j;

  if (w->s.scaling) {
    scale_data(w);
  } else {
    w->c = w->cinv = 1.0;
    for (int i = 0; i < n; i++) w->D[i] = w->Dinv[i] = 1.;
    for (int i = 0; i < m; i++) w->E[i] = w->Einv[i] = 1.;
  }
  set_rho_vec(w);
  linsys_pattern(w);
  if (linsys_factor(w)) {
    oq_cleanup(w);
    return NULL;
  }
  w->status = OQ_UNSOLVED;
  w->rho_updates = 0;
  w->rho_estimate = w->s.rho;
  return w;
}

void oq_cleanup(oq_work *w) {
  if (!w) return;
  free(w->P.p); free(w->P.i); free(w->P.x);
  free(w->A.p); free(w->A.i); free(w->A.x);
  free(w->q); free(w->l); free(w->u);
  free(w->D); free(w->Dinv); free(w->E); free(w->Einv);
  free(w->rho_vec); free(w->rho_inv_vec); free(w->constr_type);
  free(w->x); free(w->y); free(w->z); free(w->xz_tilde); free(w->x_prev); free(w->z_prev);
  free(w->Ax); free(w->Px); free(w->Aty); free(w->delta_y); free(w->Atdelta_y);
  free(w->delta_x); free(w->Pdelta_x); free(w->Adelta_x);
  free(w->D_temp); free(w->D_temp_A); free(w->E_temp);
  free(w->sol_x); free(w->sol_y);
  free(w->perm); free(w->Kb); free(w->rhs_p); free(w->Ar_p); free(w->Ar_j); free(w->Ar_k);
  free(w);
}

int oq_update_A(oq_work *w, const double *Ax_new) {
  int nnzA = w->A.p[w->A.n];
  if (w->s.scaling) unscale_data(w);
  for (int i = 0; i < nnzA; i++) w->A.x[i] = Ax_new[i];
  if (w->s.scaling) scale_data(w);
  int exitflag = linsys_factor(w);
  reset_info(w);
  return exitflag;
}

int oq_update_P(oq_work *w, const double *Px_new) {
  int nnzP = w->P.p[w->P.n];
  if (w->s.scaling) unscale_data(w);
  for (int i = 0; i < nnzP; i++) w->P.x[i] = Px_new[i];
  if (w->s.scaling) scale_data(w);
  int exitflag = linsys_factor(w);
  reset_info(w);
  return exitflag;
}

int oq_update_lin_cost(oq_work *w, const double *q_new) {
  memcpy(w->q, q_new, w->n * sizeof(double));
  if (w->s.scaling) {
    for (int i = 0; i < w->n; i++) w->q[i] = w->D[i] * w->q[i];
    for (int i = 0; i < w->n; i++) w->q[i] *= w->c;
  }
  reset_info(w);
  return 0;
}

int oq_update_bounds(oq_work *w, const double *l_new, const double *u_new) {
  for (int i = 0; i < w->m; i++)
    if (l_new[i] > u_new[i]) return 1;
  memcpy(w->l, l_new, w->m * sizeof(double));
  memcpy(w->u, u_new, w->m * sizeof(double));
  if (w->s.scaling) {
    for (int i = 0; i < w->m; i++) w->l[i] = w->E[i] * w->l[i];
    for (int i = 0; i < w->m; i++) w->u[i] = w->E[i] * w->u[i];
  }
  reset_info(w);
  return update_rho_vec(w);
}

int oq_update_lower_bound(oq_work *w, const double *l_new) {
  memcpy(w->l, l_new, w->m * sizeof(double));
  if (w->s.scaling)
    for (int i = 0; i < w->m; i++) w->l[i] = w->E[i] * w->l[i];
  for (int i = 0; i < w->m; i++)
    if (w->l[i] > w->u[i]) return 1;
  reset_info(w);
  return update_rho_vec(w);
}

int oq_update_upper_bound(oq_work *w, const double *u_new) {
  memcpy(w->u, u_new, w->m * sizeof(double));
  if (w->s.scaling)
    for (int i = 0; i < w->m; i++) w->u[i] = w->E[i] * w->u[i];
  for (int i = 0; i < w->m; i++)
    if (w->u[i] < w->l[i]) return 1;
  reset_info(w);
  return update_rho_vec(w);
}

/* osqp_solve() — built with PRINTING/PROFILING off as the reference README asks */
int oq_solve(oq_work *w) {
  int iter, can_check_termination = 0;
  double *tmp;

  if (!w->s.warm_start) cold_start(w);
  for (int i = 0; i < 4; i++) w->chk_ratio[i] = 0.0; /* (test instrumentation) */

  for (iter = 1; iter <= w->s.max_iter; iter++) {
    tmp = w->x; w->x = w->x_prev; w->x_prev = tmp; /* swap_vectors */
    tmp = w->z; w->z = w->z_prev; w->z_prev = tmp;

    update_xz_tilde(w);
    update_x(w);
    update_z(w);
    update_y(w);

    can_check_termination = w->s.check_termination && (iter % w->s.check_termination == 0);
    if (can_check_termination) {
      update_info(w, iter);
      if (check_termination(w, 0)) break;
    }
    if (w->s.adaptive_rho && w->s.adaptive_rho_interval && (iter % w->s.adaptive_rho_interval == 0)) {
      if (!can_check_termination) update_info(w, iter);
      if (adapt_rho(w)) return 1;
    }
  }

  if (!can_check_termination) {
    update_info(w, iter - 1);
    check_termination(w, 0);
  }
  if (w->status == OQ_UNSOLVED) {
    if (!check_termination(w, 1)) w->status = OQ_MAX_ITER_REACHED;
  }
  w->rho_estimate = compute_rho_estimate(w);
  store_solution(w);
  return 0;
}

const double *oq_solution_x(const oq_work *w) { return w->sol_x; }
const double *oq_solution_y(const oq_work *w) { return w->sol_y; }
/* test hook: what an integrator does to restart a workspace whose last solve was abandoned half-way -- osqp_update_rho(work,
 * rho) followed by a cold start (x = z = y = 0, auxil.c cold_start) */
int oq_restart(oq_work *w, double rho) {
  int rc = update_rho(w, rho);
  cold_start(w);
  return rc;
}

int oq_info_iter(const oq_work *w) { return w->iter; }
int oq_info_status(const oq_work *w) { return w->status; }
void oq_info_check_ratios(const oq_work *w, double out4[4]) {
  for (int i = 0; i < 4; i++) out4[i] = w->chk_ratio[i];
}
double oq_info_pri_res(const oq_work *w) { return w->pri_res; }
double oq_info_dua_res(const oq_work *w) { return w->dua_res; }
double oq_info_rho(const oq_work *w) { return w->s.rho; }
int oq_info_rho_updates(const oq_work *w) { return w->rho_updates; }
const double *oq_iter_x(const oq_work *w) { return w->x; }
const double *oq_iter_y(const oq_work *w) { return w->y; }
const double *oq_iter_z(const oq_work *w) { return w->z; }
const double *oq_scaling_D(const oq_work *w) { return w->D; }
const double *oq_scaling_E(const oq_work *w) { return w->E; }
double oq_scaling_c(const oq_work *w) { return w->c; }
