/*
 * include/qrw_hip.h — C ABI of libqrw_hip.so, the MI355X (gfx950) implementation of the
 * quadruped-reactive-walking control-loop hot path, batched over B independent Solo12
 * instances.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Each entry point names the reference interface it replaces (paths under
 * /root/reference).  In the reference these are Boost.Python/eigenpy bound C++ methods
 * (python/gepadd.cpp) called from scripts/MPC_Wrapper.py and scripts/QP_WBC.py; here the
 * same calls arrive through ctypes (see INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions
 *   - all arrays are C-order (row-major) doubles with the reference's own shapes and a
 *     leading batch dimension B; `d_` pointers are device (HBM) pointers, `h_` host pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); device-pointer
 *     entry points only enqueue work and never synchronise or allocate;
 *   - every function returns 0 on success, <0 on error (qrw_last_error() has the text);
 *     like the reference (src/MPC.cpp:558,648; src/QPWBC.cpp:270,389) a solver that stops
 *     at max-iter or detects infeasibility is NOT an error: its status is reported per
 *     instance through qrw_*_get_stats and the (possibly NaN) result is passed through;
 *   - a handle owns the per-instance persistent solver state (warm start, rho, stale
 *     B/S entries — SURVEY.md §0.4) of ONE GPU's shard; single caller, not thread-safe;
 *   - synchronisation is per handle, never device-wide: a `_host` entry point or a getter waits for the last launch of the
 *     state it reads (on whatever stream the caller launched it) and for its own copies on a private non-blocking stream of
 *     the handle -- another handle's (or anybody's) work in flight on other streams is neither waited for nor stalled;
 *   - entry points that exist for the test suite only are declared in include/qrw_hip_test.h.
 */
#ifndef QRW_HIP_H_
#define QRW_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qrw_handle_s *qrw_handle;

typedef struct {
  int32_t batch;   /* B: instances owned by this handle                                  */
  int32_t n_steps; /* N: MPC horizon steps, MPC::MPC n_steps_in (src/MPC.cpp:3); 1..32   */
  int32_t N_gait;  /* rows of fsteps / gait matrices, MPC::MPC N_gait; >= n_steps        */
  int32_t device;  /* HIP device ordinal                                                  */
  double dt_mpc;   /* MPC::MPC dt_in (src/config_solo12.yaml:11 -> 0.02)                  */
  double T_gait;   /* MPC::MPC T_gait_in (unused by the maths, kept for the call surface) */
  double dt_wbc;   /* wbc_controller dt / InvKin dt (scripts/QP_WBC.py:18 -> 0.002)       */
} qrw_config;

/* solver status values (OSQP constants, as ignored by src/MPC.cpp:558) */
#define QRW_STATUS_SOLVED 1
#define QRW_STATUS_SOLVED_INACCURATE 2
#define QRW_STATUS_MAX_ITER_REACHED (-2)
#define QRW_STATUS_PRIMAL_INFEASIBLE (-3)
#define QRW_STATUS_DUAL_INFEASIBLE (-4)
#define QRW_STATUS_NON_CVX (-7)
#define QRW_STATUS_UNSOLVED (-10)
#define QRW_STATUS_NOT_SETUP (-100) /* run with num_iter != 0 before any num_iter == 0 call */

/* Replaces the constructors MPC::MPC (src/MPC.cpp:3-32; binding python/gepadd.cpp:22-24),
 * QPWBC::QPWBC (src/QPWBC.cpp:4-30; python/gepadd.cpp:217) and InvKin::InvKin
 * (src/InvKin.cpp:3-10; python/gepadd.cpp:186) for a batch of B instances. */
int qrw_create(const qrw_config *cfg, qrw_handle *out);
int qrw_destroy(qrw_handle h);
const char *qrw_last_error(void);

/* Replaces MPC::run(num_iter, xref_in, fsteps_in) + MPC::get_latest_result()
 * (src/MPC.cpp:626-649,604; python/gepadd.cpp:27-29), i.e. what
 * MPC_Wrapper.run_MPC_synchronous does (scripts/MPC_Wrapper.py:142,148), per instance.
 *   d_xref     [B][12][N+1]   column 0 = current state
 *   d_fsteps   [B][N_gait][12] row i = horizon step i, zero = swing / beyond horizon
 *   d_num_iter [B] int32 or NULL; if NULL num_iter_scalar applies to every instance.
 *              num_iter == 0 (re)creates the QP and cold-starts the solver (MPC.cpp:636-637)
 *   d_out      [B][24][N]     x_f_applied: rows 0-11 predicted states, 12-23 forces
 * Scheduling inside the call (no reference counterpart; never visible in the results): instances are started longest first by
 * a moving average of their previous iteration counts; at N > 16 with more instances than the device holds at a time, a
 * workgroup runs at most 600 ADMM iterations of a solve, then parks it (bit-exact resume) for one of the launch's taker
 * workgroups if one is still to come -- it goes on itself otherwise -- and the parked solves are taken
 * longest-predicted-remainder first, the prediction read off the decay of the residuals.  No workgroup waits for work that a
 * workgroup not yet running would have to produce (since round 5; a taker that finds nothing parked leaves).  d_out is
 * pre-filled with NaN then: should a parked solve not reach its taker (a queue overrun, or the few instructions between a
 * parker's reservation and its store taking 2 s -- never expected), the unfinished instances keep NaN, the kernel
 * leaves a code in a host-mapped word, and THE NEXT qrw_mpc_solve of the handle returns -12 once, without launching (no device
 * sync involved; qrw_mpc_get_stats reports it as well).  The call after that runs normally and starts the unfinished instances
 * cold (zero x, z, y, rho 0.1: what OSQP's store_solution leaves after a failed solve).  NaN forces stop the robot through the
 * controller's fourth error code (qrw_controller_result).  Environment knobs: INTEGRATION.md. */
int qrw_mpc_solve(qrw_handle h, const double *d_xref, const double *d_fsteps, const int32_t *d_num_iter,
                  int32_t num_iter_scalar, double *d_out, void *stream);
/* same with host buffers (H2D, solve, D2H, synchronised) — used by the single-robot drop-in */
int qrw_mpc_solve_host(qrw_handle h, const double *h_xref, const double *h_fsteps, const int32_t *h_num_iter,
                       int32_t num_iter_scalar, double *h_out);

/* MPC::get_gait / MPC::get_Sgait (src/MPC.cpp:770-780; python/gepadd.cpp:30-31) of instance b:
 * h_gait [N_gait][4], h_Sgait [12 N] */
int qrw_mpc_get_gait(qrw_handle h, int32_t b, double *h_gait, double *h_Sgait);

/* K consecutive MPC::run calls of every instance (call indices first_num_iter .. first_num_iter + K - 1) in ONE launch:
 * d_xref [K][B][12][N+1], d_fsteps [K][B][N_gait][12] -> d_out [K][B][24][N], d_iters [K][B] int32 (may be NULL).  Results
 * and the persistent solver state afterwards are those of K calls of qrw_mpc_solve, bit for bit; what differs is the
 * ordering: there is no device-wide barrier between the calls, only each instance's own order (a task queue hands call s+1
 * of an instance out when its call s has finished), so one instance's long solve delays nobody else's next call.  For
 * inputs that are all known beforehand — replaying the planner outputs a log holds (`planner_xref`, `planner_fsteps`,
 * scripts/LoggerControl.py:61-65,142-143) to recompute `mpc_x_f` (:76,:152), or open-loop evaluation sweeps — not for a
 * closed control loop, whose next inputs depend on this call's result.  A diagnostic for an out-of-scope consumer (log
 * replay), not part of the control path.  d_out is pre-filled with NaN and d_iters with -1, so a call that never ran cannot
 * be mistaken for a result.  qrw_mpc_sequence_error waits for the handle's sequence launch and reports whether any workgroup gave up
 * waiting for a task (2 s without ANY task of the sequence finishing -- the clock restarts on observed progress, so the
 * length of K or of one instance's chain does not matter; never expected).  Forward progress assumes that workgroups of one
 * launch start in index order (true of the hardware dispatcher; also on CU-masked streams, where fewer are resident). */
int qrw_mpc_solve_sequence(qrw_handle h, int32_t K, const double *d_xref, const double *d_fsteps, int32_t first_num_iter,
                           double *d_out, int32_t *d_iters, void *stream);
int qrw_mpc_sequence_error(qrw_handle h, int32_t *timed_out);

/* Device-to-device copy of the last solve's ADMM iteration counts (int32 [B]) on `stream`: lets a caller keep per-launch
 * work statistics in HBM without a host round trip inside a timed region (bench.py's roofline accounting).  No reference
 * counterpart: OSQP's info->iter is never read by the reference (src/MPC.cpp:558). */
int qrw_mpc_copy_iters(qrw_handle h, int32_t *d_iters, void *stream);

/* Per-instance solver statistics of the last qrw_mpc_solve (host arrays of B, any may be NULL).
 * The reference never inspects them (OSQP status ignored); exposed so the silent
 * pass-through can be observed (SURVEY.md §8(b) error convention). */
int qrw_mpc_get_stats(qrw_handle h, int32_t *h_iters, int32_t *h_status, double *h_rho, double *h_pri_res,
                      double *h_dua_res);

/* Diagnostic: OSQP-ordered copies of instance b's persisted scaled iterates x[24N], z[44N],
 * y[44N] and of the last solve's scaling D[24N], E[44N], c (any may be NULL). Tests only. */
int qrw_mpc_get_state(qrw_handle h, int32_t b, double *h_x, double *h_z, double *h_y, double *h_D, double *h_E,
                      double *h_c);

/* Diagnostic: the block order the NEXT qrw_mpc_solve will use (h_order[i] = instance solved by workgroup i: a
 * permutation of 0..B-1, longest predicted solve first) and the moving average of iteration counts it was sorted by
 * (either may be NULL).  *has_order = 0 while no order exists (batch <= 1024, or before the first solve).  Synchronises
 * the device.  No reference counterpart (scheduling only; results do not depend on it).  Tests only. */
int qrw_mpc_get_order(qrw_handle h, int32_t *h_order, float *h_ema, int32_t *has_order);

/* Diagnostic: bookkeeping of the last time-sliced qrw_mpc_solve (N > 16 with more instances than resident slots; all zero
 * otherwise): priority levels and slice length in use, solves parked into each level (level 0 = first parks), taker
 * workgroups that took a parked solve, instances counted as finished.  No reference counterpart. */
int qrw_mpc_get_slice_stats(qrw_handle h, int32_t *levels, int32_t *chunk, uint32_t *h_parks_per_level /* [9] */,
                            uint32_t *h_takers, uint32_t *h_finished);

/* Replaces wbc_controller.compute (scripts/QP_WBC.py:52-131) with everything it calls:
 * Solo12InvKin.refreshAndCompute (scripts/solo12InvKin.py:44-69), InvKin::refreshAndCompute
 * (src/InvKin.cpp:23-73), the Pinocchio crba/Jacobian/rnea calls (QP_WBC.py:89-116) and
 * QPWBC::run + getters (src/QPWBC.cpp:310-390,302-308; python/gepadd.cpp:219-223).
 *   in : d_q [B][19], d_dq [B][18], d_f_cmd [B][12], d_contacts [B][4],
 *        d_pgoals / d_vgoals / d_agoals [B][3][4]
 *   out: d_tau_ff [B][12], d_qdes [B][19], d_vdes [B][18], d_f_with_delta [B][12],
 *        d_ddq_res [B][6] (QPWBC::get_ddq_res), d_feet [B][3][3][4] = feet_pos, feet_err,
 *        feet_vel (QP_WBC.py:73-80); any output may be NULL */
int qrw_wbc_compute(qrw_handle h, const double *d_q, const double *d_dq, const double *d_f_cmd,
                    const double *d_contacts, const double *d_pgoals, const double *d_vgoals, const double *d_agoals,
                    double *d_tau_ff, double *d_qdes, double *d_vdes, double *d_f_with_delta, double *d_ddq_res,
                    double *d_feet, void *stream);
int qrw_wbc_compute_host(qrw_handle h, const double *h_q, const double *h_dq, const double *h_f_cmd,
                         const double *h_contacts, const double *h_pgoals, const double *h_vgoals,
                         const double *h_agoals, double *h_tau_ff, double *h_qdes, double *h_vdes,
                         double *h_f_with_delta, double *h_ddq_res, double *h_feet);
int qrw_wbc_get_stats(qrw_handle h, int32_t *h_iters, int32_t *h_status, double *h_rho, double *h_k_since_contact);
/* Scheduling only (no reference counterpart; results agree to rounding, ADMM iteration counts are identical): lanes per robot
 * instance of qrw_wbc_compute / qrw_wbc_compute_result.  16 (default): one 16-lane DPP row per instance, batch / 4 wavefronts --
 * 27.9 us per 4096 robots on the whole chip; 4: one quad per instance, batch / 16 wavefronts of 1.55 x the length -- 43.2 us on
 * the whole chip, but the faster one on a stream restricted to few compute units (the asynchronous control loop's 32).
 * QRW_WBC16=0 makes 4 the default of new handles. */
int qrw_wbc_set_lanes(qrw_handle h, int32_t lanes);

/* Stand-alone pieces of the WBC step with the reference's own signatures, for callers that
 * use the bound classes directly (scripts/solo12InvKin.py:62-67, scripts/QP_WBC.py:107-111):
 * InvKin::refreshAndCompute (+ get_q_step, get_dq_cmd): host arrays, batch of B.
 *   h_contacts [B][4], h_goals/vgoals/agoals [B][3][4], h_posf/vf/wf/af [B][4][3], h_Jf [B][12][12]
 *   -> h_ddq, h_dq_cmd, h_q_step [B][12] */
int qrw_invkin_host(qrw_handle h, const double *h_contacts, const double *h_goals, const double *h_vgoals,
                    const double *h_agoals, const double *h_posf, const double *h_vf, const double *h_wf,
                    const double *h_af, const double *h_Jf, double *h_ddq, double *h_dq_cmd, double *h_q_step);
/* QPWBC::run(M, Jc, f_cmd, RNEA, k_contacts) + get_f_res / get_ddq_res / get_H:
 *   h_M [B][18][18] (only the diagonal of its top-left 6x6 is read: the caller masks it,
 *   scripts/QP_WBC.py:93), h_Jc [B][12][18], h_f_cmd [B][12], h_RNEA [B][6]
 *   -> h_f_res [B][12], h_ddq_res [B][6], h_H [B][12][12] (may be NULL) */
int qrw_qpwbc_host(qrw_handle h, const double *h_M, const double *h_Jc, const double *h_f_cmd, const double *h_RNEA,
                   double *h_f_res, double *h_ddq_res, double *h_H);

/* Rigid-body slice used by Solo12InvKin (scripts/solo12InvKin.py:47-59): fixed-base feet
 * kinematics. h_q12/h_dq12 [B][12] -> h_posf/vf/wf/af [B][4][3], h_Jf [B][12][12] */
int qrw_fixed_feet_host(qrw_handle h, const double *h_q12, const double *h_dq12, double *h_posf, double *h_vf,
                        double *h_wf, double *h_af, double *h_Jf);

/* Diagonal of the neutral-configuration CRBA base block (scripts/QP_WBC.py:89-93), constant. */
int qrw_get_base_inertia_diag(qrw_handle h, double *h_Y6);

/* ---------------- planners feeding the hot path ----------------
 * Batched Gait + FootstepPlanner + FootTrajectoryGenerator + StatePlanner with persistent per-instance state,
 * wired as scripts/Controller.py:119-137 wires them. The timing parameters (dt_mpc, dt_wbc, T_gait, N_gait,
 * n_steps; T_mpc = n_steps * dt_mpc) come from the handle's qrw_config. */
typedef struct {
  int32_t k_mpc;            /* WBC iterations per MPC iteration (scripts/main_solo12_control.py:123)           */
  double h_ref;             /* StatePlanner::initialize h_ref_in, FootstepPlanner::initialize h_ref_in          */
  double shoulders[12];     /* 3x4 row-major, FootstepPlanner::initialize shouldersIn (Controller.py:131-135)    */
  double max_height;        /* FootTrajectoryGenerator::initialize maxHeightIn (Controller.py:137 -> 0.05)       */
  double lock_time;         /* ... lockTimeIn (0.07)                                                             */
  double init_target[12];   /* ... targetFootstepIn, 3x4                                                         */
  double init_foot_pos[12]; /* ... initialFootPosition, 3x4                                                      */
} qrw_planner_config;

/* Replaces Gait::initialize (src/Gait.cpp:19-36), StatePlanner::initialize (src/StatePlanner.cpp:12-19),
 * FootstepPlanner::initialize (src/FootstepPlanner.cpp:22-49), FootTrajectoryGenerator::initialize
 * (src/FootTrajectoryGenerator.cpp:22-38). Returns -3 where Gait::initialize throws (N_gait too small). */
int qrw_planner_init(qrw_handle h, const qrw_planner_config *pc, void *stream);

/* One control iteration of the planners, in the order of scripts/Controller.py:222-236:
 * Gait::updateGait(k, k_mpc, q, code) (src/Gait.cpp:184-192), FootstepPlanner::updateFootsteps(k % k_mpc == 0 && k != 0,
 * k_mpc - k % k_mpc, q, b_v, b_vref) (src/FootstepPlanner.cpp:51), FootTrajectoryGenerator::update(k, o_target)
 * (src/FootTrajectoryGenerator.cpp:108), StatePlanner::computeReferenceStates(q, v, vref, 0) (src/StatePlanner.cpp:21).
 *   in : d_q7 [B][q_ld] of which the first 7 entries of a row are read (position, quaternion xyzw; q_ld = 7, or 19 to
 *        pass the q [B][19] of qrw_controller_update_state directly), d_hv [B][6], d_vref [B][6], d_code [B] int32
 *        joystick codes or NULL (code_scalar then applies; 1 pacing, 2 bounding, 3 trot, 4 static as
 *        src/Gait.cpp:194-219; 5 walk)
 *   out: d_xref [B][12][N+1], d_fsteps [B][N_gait][12], d_gait [B][N_gait][4] (current gait), d_target [B][3][4]
 *        (o_targetFootstep), d_feet_pva [B][3][3][4] (foot position, velocity, acceleration goals), d_contacts [B][4]
 *        (row 0 of the current gait = the `contacts` operand of qrw_wbc_compute); any may be NULL */
int qrw_planner_step(qrw_handle h, int32_t k, const double *d_q7, int32_t q_ld, const double *d_hv, const double *d_vref,
                     const int32_t *d_code, int32_t code_scalar, double *d_xref, double *d_fsteps, double *d_gait,
                     double *d_target, double *d_feet_pva, double *d_contacts, void *stream);

/* One planner method at a time with host buffers (backs the bound-class drop-ins Gait / FootstepPlanner /
 * FootTrajectoryGenerator / StatePlanner, python/gepadd.cpp:44-181). mode is a bit set: 2 Gait::updateGait,
 * 4 FootstepPlanner::updateFootsteps (k_footsteps, refresh), 8 FootTrajectoryGenerator::update (h_target_in or the stored
 * target), 16 StatePlanner::computeReferenceStates (z_average), 32 only copy the requested outputs. */
int qrw_planner_call_host(qrw_handle h, int32_t mode, int32_t k, int32_t k_footsteps, int32_t refresh, const double *h_q7,
                          const double *h_v6, const double *h_vref6, int32_t code, const double *h_target_in,
                          double z_average, double *h_xref, double *h_fsteps, double *h_gait, double *h_target,
                          double *h_feet_pva);

/* Getter of planner state items of instance b (count doubles): which = 0 past gait, 1 current gait, 2 desired gait
 * (N_gait*4 each), 3 newPhase, 4 is_static, 5 remainingTime, 6 #swing feet, 7 targetFootstep (12), 8 o_targetFootstep (12),
 * 9/10/11 foot position/velocity/acceleration (12), 12 t0s (4), 13 t_swing (4), 14 footsteps (N_gait*12, [row][xyz][foot]),
 * 15 currentFootstep (12), 16 q_static (7), 17 trajectory target (12). */
int qrw_planner_get_host(qrw_handle h, int32_t which, int32_t b, int32_t count, double *h_out);

/* ---------------- controller glue around the hot path ----------------
 * The element-wise parts of Controller.compute (scripts/Controller.py) between planners, MPC and WBC, batched, with
 * their per-instance state (perfect x/y/yaw integration, previous foot commands, previous q_des / v_des, error flag). */
/* Controller.__init__ state (scripts/Controller.py:119-123,154): d_q_init12 [B][12] (or NULL = zeros) -> qdes[7:]. */
int qrw_controller_init(qrw_handle h, const double *d_q_init12, double h_ref, void *stream);
/* Controller.updateState (scripts/Controller.py:381-426, non-static branch): d_joy_vref [B][6], d_q_filt [B][19],
 * d_v_filt [B][18], d_rpy [B][3] -> d_q [B][19], d_v [B][18], d_hv [B][6], d_vref [B][6] (may be NULL),
 * d_oRh_oTh [B][12] = oRh row-major | oTh (may be NULL). */
int qrw_controller_update_state(qrw_handle h, const double *d_joy_vref, const double *d_q_filt, const double *d_v_filt,
                                const double *d_rpy, double *d_q, double *d_v, double *d_hv, double *d_vref,
                                double *d_oRh_oTh, void *stream);
/* WBC target assembly (scripts/Controller.py:258-296): d_x_f_mpc [B][24][N], d_xref [B][12][N+1], d_feet_pva [B][3][3][4]
 * (FootTrajectoryGenerator position / velocity / acceleration), d_v [B][18] -> d_x_f_wbc [B][24] (may be NULL),
 * d_q_wbc [B][19], d_b_v [B][18], d_f_cmd [B][12] (= x_f_wbc[12:], may be NULL), d_feet_cmd [3][B][3][4] (planes feet_p_cmd, feet_v_cmd, feet_a_cmd,
 * each directly usable as the pgoals / vgoals / agoals operand of qrw_wbc_compute). */
int qrw_controller_wbc_inputs(qrw_handle h, const double *d_x_f_mpc, const double *d_xref, const double *d_feet_pva,
                              const double *d_v, double *d_x_f_wbc, double *d_q_wbc, double *d_b_v, double *d_f_cmd,
                              double *d_feet_cmd, void *stream);
/* Result + security_check (scripts/Controller.py:306-310,341-365): d_tau_ff [B][12], d_qdes [B][19], d_vdes [B][18],
 * d_q_filt [B][19], d_v_secu [B][12] -> d_result [B][5][12] = P, D, q_des, v_des, tau_ff (0.8 x), d_error_flag [B]
 * int32 (0 ok, 1 joint position, 2 joint velocity, 3 torque: the reference's three codes; 4: a non-finite tau_ff / q_des /
 * v_des -- NOT in the reference, whose `> limit` comparisons are blind to NaN and would pass it to the motors; sticky, may be
 * NULL).  Any code sets the reference's security output (P = 0, D = 0.1, zero targets and torques). */
int qrw_controller_result(qrw_handle h, const double *d_tau_ff, const double *d_qdes, const double *d_vdes,
                          const double *d_q_filt, const double *d_v_secu, double *d_result, int32_t *d_error_flag,
                          void *stream);

/* Bookkeeping of MPC_Wrapper.solve on the result the loop currently reads (scripts/MPC_Wrapper.py:89-102; observable in
 * the reference's asynchronous mode only, where get_latest_result keeps returning it until the child process delivers):
 * rows 12..23 of d_x_f_mpc [B][24][N] are rolled one column to the left (np.roll, the first column wraps to the end) and,
 * when the last non-zero row of d_gait [B][N_gait][4] differs from row 0, the last column becomes m g / n_contacts
 * (mass 2.5) on that row's stance feet.  The caller applies it when k > 2 (:89), as the reference does. */
int qrw_mpc_result_shift(qrw_handle h, const double *d_gait, double *d_x_f_mpc, void *stream);

/* Fused forms of the above, one launch each, for the device-resident control loop (same arithmetic):
 * qrw_control_pre = qrw_controller_update_state + qrw_planner_step (fed with its q, h_v, v_ref) and, when d_x_f_mpc is
 * not NULL (the MPC result to use is already known: every iteration that does not solve), + qrw_controller_wbc_inputs;
 * qrw_wbc_compute_result = qrw_wbc_compute + qrw_controller_result (scripts/Controller.py:200-326 end to end in two
 * launches plus the MPC solve every k_mpc-th iteration).  Operands as in the separate entry points.  d_fsteps and d_gait may
 * be NULL; with d_fsteps NULL and d_x_f_mpc given (an iteration that does not solve: nobody reads the MPC's inputs) only
 * column 0 and horizon step 1 of d_xref are written, which is all the WBC target assembly reads of it, and of the planner's
 * own copy of the footstep table (qrw_planner_get item "fsteps") only rows 0 and 1 are refreshed -- row 1 is what
 * updateNewContact may take at the next gait change (src/FootstepPlanner.cpp:225-230), the other rows are rewritten by the
 * next call with d_fsteps before anything reads them. */
int qrw_control_pre(qrw_handle h, int32_t k, const double *d_joy_vref, const double *d_q_filt, const double *d_v_filt,
                    const double *d_rpy, const int32_t *d_code, int32_t code_scalar, const double *d_x_f_mpc, double *d_q,
                    double *d_v, double *d_hv, double *d_vref, double *d_oRh_oTh, double *d_xref, double *d_fsteps,
                    double *d_gait, double *d_target, double *d_feet_pva, double *d_contacts, double *d_x_f_wbc,
                    double *d_q_wbc, double *d_b_v, double *d_f_cmd, double *d_feet_cmd, void *stream);
int qrw_wbc_compute_result(qrw_handle h, const double *d_q, const double *d_dq, const double *d_f_cmd,
                           const double *d_contacts, const double *d_pgoals, const double *d_vgoals,
                           const double *d_agoals, double *d_tau_ff, double *d_qdes, double *d_vdes,
                           double *d_f_with_delta, double *d_ddq_res, double *d_feet, const double *d_q_filt,
                           const double *d_v_secu, double *d_result, int32_t *d_error_flag, void *stream);

/* A control iteration that does not solve (qrw_control_pre given d_x_f_mpc, without d_fsteps / d_gait, + qrw_wbc_compute_result
 * on its outputs) with every buffer BOUND ONCE: a control loop passes the same buffers on every tick -- the sensor values are
 * written into fixed input arrays, the outputs read from fixed ones -- so the ~50 pointer arguments of the two calls need not be
 * marshalled again every 2 ms (scripts/Controller.py:200-326 runs 9 of 10 iterations this way, :246).  qrw_iteration_bind stores
 * the pointers in the handle (no launch, no allocation; binding again replaces them); qrw_iteration_step enqueues the two launches
 * for iteration k on `stream` using the MPC result d_x_f_mpc [B][24][N] (whichever buffer the loop has adopted).  Same kernels,
 * same arithmetic as the two separate calls.  Why not a HIP graph: on this runtime hipGraphLaunch of the captured pair costs
 * 19 us of host time against 8.6-9.3 us for the two launches (scripts/ubench/graph_launch.hip, profiles/r5_graph_launch.txt). */
typedef struct {
  /* inputs (read every step) */
  const double *d_joy_vref, *d_q_filt, *d_v_filt, *d_rpy, *d_v_secu;
  const int32_t *d_code; /* per-robot joystick codes or NULL */
  int32_t code_scalar;   /* used when d_code is NULL         */
  /* outputs of qrw_control_pre (d_fsteps / d_gait are not produced on an iteration that does not solve) */
  double *d_q, *d_v, *d_hv, *d_vref, *d_oRh_oTh, *d_xref, *d_target, *d_feet_pva, *d_contacts, *d_x_f_wbc, *d_q_wbc, *d_b_v,
      *d_f_cmd, *d_feet_cmd;
  /* outputs of qrw_wbc_compute_result */
  double *d_tau_ff, *d_qdes, *d_vdes, *d_f_with_delta, *d_ddq_res, *d_feet, *d_result;
  int32_t *d_error_flag;
} qrw_iteration_buffers;
int qrw_iteration_bind(qrw_handle h, const qrw_iteration_buffers *buffers);
int qrw_iteration_step(qrw_handle h, int32_t k, const double *d_x_f_mpc, void *stream);

/* ---------------- asynchronous MPC ----------------
 * The reference runs the MPC in a child process on its own CPU core and polls a shared flag
 * (scripts/MPC_Wrapper.py:150-298).  Here the MPC gets its own HIP stream restricted to a subset of the compute
 * units, so that the control loop's kernels (planners, WBC, glue) on a second stream restricted to the remaining
 * units never queue behind a running solve.  Creates a stream whose kernels only run on compute units
 * [first_cu, first_cu + n_cus) of `device` (hipExtStreamCreateWithCUMask); n_cus <= 0 = no restriction. */
int qrw_stream_create(int32_t device, int32_t first_cu, int32_t n_cus, void **stream);
int qrw_stream_destroy(void *stream);
/* number of compute units of the device (256 on MI355X) */
int qrw_device_cu_count(int32_t device, int32_t *n_cus);
/* Stream hand-over without host synchronisation: everything enqueued on `waiter` after this call runs after what is on `signaller`
 * now (hipEventRecord on `signaller` + hipStreamWaitEvent on `waiter`, with an event the handle keeps: two runtime calls, where
 * torch's Stream.wait_stream creates and destroys an event object each time -- ~12 us, and a joined stream-group iteration needs
 * several).  Plumbing of the stream-group / asynchronous modes; no reference counterpart (a child process and flags there). */
int qrw_stream_wait_stream(qrw_handle h, void *waiter, void *signaller);

/* Diagnostic: checks on the device what the MPC solver's linear algebra relies on — the row_newbcast form of
 * v_fmac_f64, the twisted block sweeps in the production LDS layout and the in-register Gauss-Jordan inverse
 * (csrc/chain_sweep.h) — against a host evaluation — and then one whole known-answer MPC solve (the reference's
 * four-stance immobile scenario, scripts/test_mpc.py:54-62: 350 ADMM iterations, equal vertical forces, see
 * csrc/qrw_api.hip).  0 = ok, 1 = sweep mismatch, 2 = the known-answer solve is wrong, <0 = HIP error. *max_err (of the
 * sweep check) may be NULL.  qrw_create runs the known-answer solve once per process and device and fails with -20 if
 * this build of the library computes wrong results (docs/HISTORY.md 6b). */
int qrw_selftest_sweeps(double *max_err);

/* workspace sizes, for callers that budget HBM */
int64_t qrw_state_bytes(qrw_handle h);

#ifdef __cplusplus
}
#endif
#endif /* QRW_HIP_H_ */
