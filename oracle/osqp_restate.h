/*
 * oracle/osqp_restate.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the OSQP 0.6.x ADMM QP solver, which the reference links as a
 * third-party dependency that is ABSENT from /root/reference and from this image
 * (reference README.md:21-31 "clone osqp ... master", un-pinned; the C API used by
 * the reference — osqp_setup(&work,data,settings), osqp_update_A(work,x,NULL,0) —
 * is the 0.6.x API).  Reference call sites this stands in for:
 *   src/MPC.cpp:31,527-541,548-549,558,573-578
 *   src/QPWBC.cpp:28,239-252,258-265,270,287
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and OSQP
 * itself cannot be built or imported here, so this restatement (written from the
 * published algorithm: Stellato et al., "OSQP: an operator splitting solver for
 * quadratic programs", and the 0.6.x sources' function structure) is checked only
 * against analytic known-answers and an independent high-accuracy QP solve.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef OSQP_RESTATE_H_
#define OSQP_RESTATE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define OQ_INFTY 1e30
#define OQ_RHO_MIN 1e-06
#define OQ_RHO_MAX 1e06
#define OQ_RHO_EQ_OVER_RHO_INEQ 1e03
#define OQ_RHO_TOL 1e-04
#define OQ_MIN_SCALING 1e-04
#define OQ_MAX_SCALING 1e+04

/* status values (osqp constants.h) */
#define OQ_DUAL_INFEASIBLE_INACCURATE 4
#define OQ_PRIMAL_INFEASIBLE_INACCURATE 3
#define OQ_SOLVED_INACCURATE 2
#define OQ_SOLVED 1
#define OQ_MAX_ITER_REACHED (-2)
#define OQ_PRIMAL_INFEASIBLE (-3)
#define OQ_DUAL_INFEASIBLE (-4)
#define OQ_NON_CVX (-7)
#define OQ_UNSOLVED (-10)

typedef struct {
  int m, n; /* rows, cols */
  int *p;   /* column pointers (n+1) */
  int *i;   /* row indices */
  double *x;
} oq_csc;

typedef struct {
  double rho, sigma, alpha;
  double eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  double adaptive_rho_tolerance;
  int max_iter, scaling, adaptive_rho, adaptive_rho_interval;
  int check_termination, warm_start, scaled_termination;
} oq_settings;

typedef struct oq_work oq_work;

void oq_set_default_settings(oq_settings *s);

/* P: upper-triangular CSC (n x n); A: CSC (m x n). Data are copied.
 * perm (length n, may be NULL): elimination ordering used for the banded Cholesky of
 * the reduced KKT matrix P + sigma I + A' diag(rho) A. Any exact linear solve yields
 * the same ADMM iterates up to rounding (OSQP itself uses QDLDL on the KKT form). */
oq_work *oq_setup(const oq_csc *P, const oq_csc *A, const double *q, const double *l, const double *u,
                  const oq_settings *settings, const int *perm);
void oq_cleanup(oq_work *w);

int oq_update_A(oq_work *w, const double *Ax_new);          /* osqp_update_A(work, x, NULL, 0) */
int oq_update_P(oq_work *w, const double *Px_new);          /* osqp_update_P(work, x, NULL, 0) */
int oq_update_lin_cost(oq_work *w, const double *q_new);    /* osqp_update_lin_cost */
int oq_update_bounds(oq_work *w, const double *l_new, const double *u_new);
int oq_update_lower_bound(oq_work *w, const double *l_new);
int oq_update_upper_bound(oq_work *w, const double *u_new);
int oq_solve(oq_work *w);
int oq_restart(oq_work *w, double rho);                     /* osqp_update_rho + cold start (tests of the aborted-launch path) */

/* results / introspection */
const double *oq_solution_x(const oq_work *w);
const double *oq_solution_y(const oq_work *w);
int oq_info_iter(const oq_work *w);
int oq_info_status(const oq_work *w);
double oq_info_pri_res(const oq_work *w);
void oq_info_check_ratios(const oq_work *w, double out4[4]); /* test instrumentation: residual / tolerance of the last two checks */
double oq_info_dua_res(const oq_work *w);
double oq_info_rho(const oq_work *w);
int oq_info_rho_updates(const oq_work *w);
/* scaled iterates (warm-start state) and scaling, for parity checks of persisted state */
const double *oq_iter_x(const oq_work *w);
const double *oq_iter_y(const oq_work *w);
const double *oq_iter_z(const oq_work *w);
const double *oq_scaling_D(const oq_work *w);
const double *oq_scaling_E(const oq_work *w);
double oq_scaling_c(const oq_work *w);

#ifdef __cplusplus
}
#endif
#endif
