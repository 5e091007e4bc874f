/*
 * oracle/qrw_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's control-loop hot path:
 *   - MPC QP build + solve            /root/reference/src/MPC.cpp (all), include/qrw/MPC.hpp
 *   - WBC box-QP                      src/QPWBC.cpp:4-30,85-343,481-537, include/qrw/QPWBC.hpp:26-65
 *   - InvKin                          src/InvKin.cpp:3-73, include/qrw/InvKin.hpp:56-66
 *   - WBC / InvKin Python drivers     scripts/QP_WBC.py:52-131, scripts/solo12InvKin.py:44-69
 *   - st_to_cc triplet->CSC semantics src/st_to_cc.cpp:1622-1854
 * plus restatements of the third-party slices those files call and that are ABSENT
 * from /root/reference and from this image: OSQP 0.6.x (osqp_restate.c) and the
 * Pinocchio rigid-body algorithms on the Solo12 model (rbd_oracle.c, constants in
 * include/qrw_solo12_model.h).
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or
 * fixtures for this path (its tests are add(1,2)==3 leftovers and one stale
 * property test, scripts/test_mpc.py) and none of it can be compiled or imported
 * here (Eigen, OSQP, Pinocchio, eigenpy, Boost.Python, the Solo12 URDF: all absent;
 * no network).  The oracle is checked against the analytic properties the stale
 * test states (equal forces / sum f_z = m g in four-stance), structural invariants
 * of the QP, and an independent high-accuracy solve of the same QP.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import,
 * link or call this library; the product path never does.
 *
 * Array conventions at this boundary: numpy C-order (row-major) doubles, with the
 * reference's own shapes: xref 12 x (N+1), fsteps N_gait x 12, result 24 x N.
 */
#ifndef QRW_ORACLE_H_
#define QRW_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------- MPC (src/MPC.cpp) ------------------------- */
typedef struct mpc_oracle mpc_oracle;

mpc_oracle *mpc_oracle_create(double dt, int n_steps, double T_gait, int N_gait); /* MPC::MPC  MPC.cpp:3-32 */
void mpc_oracle_destroy(mpc_oracle *o);
int mpc_oracle_run(mpc_oracle *o, int num_iter, const double *xref, const double *fsteps); /* MPC::run :626 */
void mpc_oracle_get_latest_result(const mpc_oracle *o, double *out_24xN);                 /* :604 */
void mpc_oracle_get_gait(const mpc_oracle *o, double *out_Ngaitx4);                       /* :770 */
void mpc_oracle_get_Sgait(const mpc_oracle *o, double *out_12N);                          /* :776 */
/* introspection for tests */
int mpc_oracle_restart(mpc_oracle *o, double rho); /* osqp_update_rho + cold start of the MPC's workspace (test hook) */
int mpc_oracle_iter(const mpc_oracle *o);
int mpc_oracle_status(const mpc_oracle *o);
double mpc_oracle_rho(const mpc_oracle *o);
double mpc_oracle_pri_res(const mpc_oracle *o);
void mpc_oracle_check_ratios(const mpc_oracle *o, double *out4); /* tests: residual / tolerance of the last two termination checks */
double mpc_oracle_dua_res(const mpc_oracle *o);
int mpc_oracle_nnz_ML(const mpc_oracle *o);
/* copies of the assembled (unscaled) QP, as handed to OSQP: CSC of ML, P diag, bounds */
void mpc_oracle_get_ML(const mpc_oracle *o, int *p, int *i, double *x);
void mpc_oracle_get_P(const mpc_oracle *o, int *p, int *i, double *x);
void mpc_oracle_get_bounds(const mpc_oracle *o, double *l, double *u);
void mpc_oracle_get_solution(const mpc_oracle *o, double *x_24N);
void mpc_oracle_get_iterates(const mpc_oracle *o, double *x, double *z, double *y);

/* ---------------- rigid-body slices (Pinocchio semantics) ---------------- */
/* Fixed-base 12-DoF Solo12 (scripts/solo12InvKin.py:47-59): per foot i (FL,FR,HL,HR)
 * posf 4x3, vf 4x3 (LOCAL_WORLD_ALIGNED linear), wf 4x3 (angular), af 4x3 (LWA *spatial*
 * linear acceleration with zero joint acceleration), Jf 12x12 (rows 3i..3i+2 = LWA linear
 * Jacobian of foot i). */
void rbd_oracle_fixed_feet(const double *q12, const double *dq12, double *posf, double *vf, double *wf, double *af,
                           double *Jf);
/* Free-flyer 18-DoF Solo12: q = (p, quat xyzw, joints), v = (v lin, w) in base frame + joints. */
void rbd_oracle_rnea(const double *q19, const double *v18, const double *a18, double *tau18); /* pin.rnea */
void rbd_oracle_crba_base_block(const double *q19, double *M6x6);                             /* pin.crba [:6,:6] */
void rbd_oracle_crba(const double *q19, double *M18x18);                                       /* pin.crba (full, symmetric) */
void rbd_oracle_feet_jacobians(const double *q19, double *J12x18); /* getFrameJacobian(LWA)[:3] per foot */

/* ------------------------- InvKin (src/InvKin.cpp) ------------------------- */
/* contacts 4, goals/vgoals/agoals 3x4, posf/vf/wf/af 4x3, Jf 12x12 -> ddq 12, dq_cmd 12, q_step 12 */
void invkin_oracle_refresh_and_compute(const double *contacts, const double *goals, const double *vgoals,
                                       const double *agoals, const double *posf, const double *vf, const double *wf,
                                       const double *af, const double *Jf, double *ddq, double *dq_cmd,
                                       double *q_step);

/* ------------------------- QPWBC (src/QPWBC.cpp) ------------------------- */
typedef struct qpwbc_oracle qpwbc_oracle;
qpwbc_oracle *qpwbc_oracle_create(void);
void qpwbc_oracle_destroy(qpwbc_oracle *o);
/* M 18x18, Jc 12x18, f_cmd 12, RNEA 6 (k_contact is accepted and unused, QPWBC.cpp:345-362) */
int qpwbc_oracle_run(qpwbc_oracle *o, const double *M, const double *Jc, const double *f_cmd, const double *RNEA,
                     const double *k_contact);
void qpwbc_oracle_get_f_res(const qpwbc_oracle *o, double *f12);
void qpwbc_oracle_get_ddq_res(const qpwbc_oracle *o, double *ddq6);
void qpwbc_oracle_get_H(const qpwbc_oracle *o, double *H12x12);
int qpwbc_oracle_iter(const qpwbc_oracle *o);
int qpwbc_oracle_status(const qpwbc_oracle *o);
double qpwbc_oracle_rho(const qpwbc_oracle *o);

/* --------- wbc_controller.compute (scripts/QP_WBC.py:52-131) --------- */
typedef struct wbc_oracle wbc_oracle;
wbc_oracle *wbc_oracle_create(double dt);
void wbc_oracle_destroy(wbc_oracle *o);
/* q 19, dq 18, f_cmd 12, contacts 4, pgoals/vgoals/agoals 3x4 (row-major) ->
 * tau_ff 12, qdes 19, vdes 18, f_with_delta 12, ddq_res 6 (each may be NULL) */
int wbc_oracle_compute(wbc_oracle *o, const double *q, const double *dq, const double *f_cmd, const double *contacts,
                       const double *pgoals, const double *vgoals, const double *agoals, double *tau_ff, double *qdes,
                       double *vdes, double *f_with_delta, double *ddq_res);
int wbc_oracle_qp_iter(const wbc_oracle *o);
int wbc_oracle_qp_status(const wbc_oracle *o); /* tests only */
double wbc_oracle_qp_rho(const wbc_oracle *o);  /* tests only */
void wbc_oracle_get_feet(const wbc_oracle *o, double *feet_pos3x4, double *feet_err3x4, double *feet_vel3x4);
void wbc_oracle_get_k_since_contact(const wbc_oracle *o, double *k4);

/* --------- planners feeding the hot path (oracle/planner_oracle.c) --------- */
typedef struct planner_oracle planner_oracle;
planner_oracle *planner_oracle_create(double dt_mpc, double dt_wbc, double T_gait, double T_mpc, int N_gait, int k_mpc,
                                      double h_ref, const double *shoulders3x4, double max_height, double lock_time,
                                      const double *init_target3x4, const double *init_foot_pos3x4);
void planner_oracle_destroy(planner_oracle *o);
void planner_oracle_gait_update(planner_oracle *o, int k, const double *q7, int code);          /* Gait::updateGait */
void planner_oracle_footsteps_update(planner_oracle *o, int refresh, int k, const double *q7, const double *b_v,
                                     const double *b_vref, double *out_target3x4);              /* FootstepPlanner::updateFootsteps */
void planner_oracle_traj_update(planner_oracle *o, int k, const double *target3x4);             /* FootTrajectoryGenerator::update */
void planner_oracle_state_compute(planner_oracle *o, const double *q7, const double *v6, const double *vref6,
                                  double z_average);                                            /* StatePlanner::computeReferenceStates */
void planner_oracle_step(planner_oracle *o, int k, const double *q7, const double *h_v6, const double *vref6, int code);
double planner_oracle_phase_duration(planner_oracle *o, int i, int j, double value);            /* Gait::getPhaseDuration */
void planner_oracle_get_gaits(const planner_oracle *o, double *past, double *cur, double *des);
void planner_oracle_get_flags(const planner_oracle *o, double *out4); /* newPhase, is_static, remainingTime, #swing feet */
void planner_oracle_get_xref(const planner_oracle *o, double *xref);
void planner_oracle_get_footsteps(const planner_oracle *o, double *fsteps_Ngx12, double *target3x4, double *o_target3x4);
void planner_oracle_get_Rz(const planner_oracle *o, double *out3x3); /* FootstepPlanner::getRz, FootstepPlanner.cpp:236 */
void planner_oracle_get_feet(const planner_oracle *o, double *pos, double *vel, double *acc, double *t0s, double *t_swing);

/* batched helpers for the bench's cpu_baseline leg: `threads` OpenMP threads over instances */
int mpc_oracle_run_batch(mpc_oracle **o, int B, const int *num_iter, const double *xref, const double *fsteps,
                         double *out, int threads);
int wbc_oracle_compute_batch(wbc_oracle **o, int B, const double *q, const double *dq, const double *f_cmd,
                             const double *contacts, const double *pgoals, const double *vgoals, const double *agoals,
                             double *tau_ff, double *qdes, double *vdes, double *f_with_delta, int threads);

#ifdef __cplusplus
}
#endif
#endif
