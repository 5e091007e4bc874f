// Host <-> kernel argument blocks and per-instance state layouts (gfx950 build only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qrw {

constexpr int kMpcMaxN = 32;  // horizon steps: 16 per wavefront (lane = 4*step + foot), one or two wavefronts per instance
inline int mpc_threads(int n_steps) { return n_steps <= 16 ? 64 : 128; }

constexpr int kStatusSolved = 1;
constexpr int kStatusSolvedInaccurate = 2;
constexpr int kStatusMaxIter = -2;
constexpr int kStatusPrimalInf = -3;
constexpr int kStatusDualInf = -4;
constexpr int kStatusNonCvx = -7;
constexpr int kStatusUnsolved = -10;
constexpr int kStatusNotSetup = -100;

// MPC persistent state: st[instance][item][thread], thread = 4*step + foot (coalesced rows of mpc_threads(N) doubles)
enum MpcStateItem {
  kStXX = 0,    // x: state entries X_k[3j+t]            (OSQP scaled iterate)
  kStXF = 3,    // x: force entries f_k[3j+t]
  kStZD = 6,    // z: dynamics rows
  kStZC = 9,    // z: cone rows (5)
  kStYD = 14,   // y: dynamics rows
  kStYS = 17,   // y: force-enable rows (their z is identically 0)
  kStYC = 20,   // y: cone rows (5)
  kStB = 25,    // B[9+r][3j+t] (9), stale beyond the gait length like ML->x (MPC.cpp:422)
  kStS = 34,    // S_gait entries (3)
  kStRho = 37,  // rho (wave-uniform)
  kStDX = 38,   // last solve's scaling, diagnostics only
  kStDF = 41,
  kStED = 44,
  kStES = 47,
  kStEC = 50,
  kStC = 55,
  kMpcStItems = 56
};

// diagnostic builds: doubles per instance in MpcArgs::prof (-DQRW_TRACE_RES: 20 x (primal ratio, dual ratio, rho) at the
// adaptive-rho tests, scripts/gpu_res_trace.py)
#ifdef QRW_TRACE_RES
constexpr int kMpcProfItems = 64 + 160;  // + the primal ratio at every termination check (every 25 iterations)
#else
constexpr int kMpcProfItems = 10;
#endif
struct MpcArgs {
  int B, N, N_gait;
  double dt;
  const double* xref;    // [B][12][N+1]
  const double* fsteps;  // [B][N_gait][12]
  const int32_t* num_iter;  // [B] or null
  int num_iter_scalar;
  double* out;  // [B][24][N]
  double* st;   // [B][kMpcStItems][64]
  int* gait;    // [B][N_gait][4]
  int* flags;   // [B] set once an instance has been set up (num_iter == 0 seen)
  int* iters;
  int* status;
  double* rho_out;
  double* pri;
  double* dua;
  int* rho_updates;
  const int* order;  // optional [B]: block i solves instance order[i] (longest-first scheduling)
  double* prof;  // optional [B][kMpcProfItems] phase cycle counters / residual trace (diagnostic builds only)
  // sequences (qrw_mpc_solve_sequence): xref / fsteps / out hold seq_K call-major blocks, the queue hands tasks out
  int seq_K;
  int* queue;        // [seq_K * B]: the levels' FIFOs one after the other
  unsigned* qctr;    // [kSeqQctrWords] per-level heads / tails, error flag, level sizes and bases (mpc_kernel.hip)
  int* seq_hot;      // [B] priority level of each instance (0 = longest chains)
  int* seq_first;    // [B] call-0 tasks dealt by workgroup index
  int seq_groups;    // workgroups resident at a time = call-0 tasks dealt by index
  int* seq_iters;    // optional [seq_K][B]
  // preemptive launch (mpc_preemptive_launch): time slices of pre_chunk iterations; parked solves wait in pre_levels FIFOs
  // (level 0: parked after the first slice; 1..: by predicted remaining iterations, most first)
  int pre_chunk;      // iterations per time slice (cut at the next multiple of 200)
  int pre_cmax;       // most slices one solve can need = workgroups per instance in the grid
  int pre_cap;        // slots in pre_queue = B * (pre_cmax - 1)
  int* pre_queue;     // [pre_levels][pre_cap] parked instances, one FIFO per priority level, -1 = not filled yet
  unsigned* pre_ctr;  // [kPreCtrWords] takers that took a solve / parks / finished / error / progress / claims / per-level head and tail
  int* pause_it;      // [B] iteration a solve was parked at, 0 = not parked
  int pre_levels;     // 1: one FIFO (plain round robin); > 1: level 0 = solves parked after their first slice, levels 1.. by the
                      // remaining iterations predicted from the residuals' decay (most first), pre_bin iterations per level
  int pre_bin;
  int pre_block0;     // index of this launch's first workgroup in the B * pre_cmax grid (0; B for the takers' own launch of a
                      // batch that is resident all at once, mpc_preemptive_launch)
};
constexpr int kPreMaxLevels = 9;
// pre_ctr words (one layout for the kernel, mpc_kernel.hip, and the host readers, qrw_api.hip): takers that took a solve, solves
// parked in total, finished instances, error (3, 4: a taker's solve, reserved for it, did not arrive; 2: a level's queue overran;
// 9: forced by a test), progress (chunks ended: the give-up clock restarts on it), claims (taker workgroups spoken for: by a solve
// parked for them, or by leaving empty-handed); head of level l at kPreLevelWord + 2 l, its tail at + 2 l + 1
constexpr int kPreTicketWord = 0, kPreParksWord = 16, kPreDoneWord = 32, kPreErrWord = 33, kPreProgressWord = 34, kPreClaimsWord = 40;
constexpr int kPreLevelWord = 48;
constexpr int kPreCtrWords = kPreLevelWord + 2 * kPreMaxLevels + 14;
static_assert(kPreProgressWord < kPreClaimsWord && kPreClaimsWord < kPreLevelWord && kPreLevelWord + 2 * kPreMaxLevels <= kPreCtrWords, "pre_ctr too small");
// resident_slots: two-wavefront instances the device holds at a time (a batch that fits is launched as two grids, see there)
int mpc_preemptive_launch(const MpcArgs& a, int resident_slots, hipStream_t stream);
int mpc_pre_error_flush(const unsigned* pre_ctr, unsigned* host_word, hipStream_t stream);  // error word -> host-mapped word, if set

int mpc_launch(const MpcArgs& a, hipStream_t stream);
constexpr int kSeqErrWord = 4 * 16;              // kSeqLevels * kSeqStride: the error flag's word in qctr (mpc_kernel.hip)
constexpr int kSeqQctrWords = kSeqErrWord + 32;  // + error / diagnostics / level totals / level bases
int mpc_sequence_launch(const MpcArgs& a, hipStream_t stream);
int mpc_order_launch(const int* iters, float* ema, int* order, int B, hipStream_t stream);

// WBC persistent state: st[instance][item]
enum WbcStateItem {
  kWsX = 0,     // 12 scaled x
  kWsZ = 12,    // 20 scaled z
  kWsY = 32,    // 20 scaled y
  kWsRho = 52,
  kWsInit = 53,
  kWsKsc = 54,  // k_since_contact (4)
  kWsG = 58,    // previous call's linear cost g (12): OSQP rescales with the OLD q inside osqp_update_P
  kWbcStItems = 70
};

struct WbcArgs {
  int B;
  double dt;
  const double *q, *dq, *f_cmd, *contacts, *pgoals, *vgoals, *agoals;
  double *tau_ff, *qdes, *vdes, *f_with_delta, *ddq_res, *feet;
  double* st;  // [B][kWbcStItems]
  int* iters;
  int* status;
  double Y[6];  // diagonal of the neutral-configuration CRBA base block (constant)
  // optional stand-alone modes (host-API pieces): see wbc_kernel.hip
  int mode;
  int lanes16;  // full compute (mode 0): 1 = wbc16_kernel (sixteen lanes per instance), 0 = wbc_kernel (one quad per instance)
  const double *in0, *in1, *in2, *in3, *in4, *in5, *in6, *in7, *in8;
  double *out0, *out1, *out2, *out3, *out4;
  // optional fused tail of a control iteration (Controller result + security_check, controller_glue.h)
  double* c_cs;  // [kCtrlStItems][B] controller state, null = off
  const double *c_qfilt, *c_vsecu;
  double* c_result;
  int32_t* c_err;
};

int wbc_launch(const WbcArgs& a, hipStream_t stream);
int pinv6_launch(const double* d_M18, double* d_Yinv, int B, hipStream_t stream);  // pseudoInverse<> of M[:6,:6], [B][6][6]

}  // namespace qrw

namespace qrw {
int sweeps_selftest(double* max_err);

// ---- planners (planner_kernel.hip)
enum PlannerMode { kPlanInit = 1, kPlanGait = 2, kPlanFootsteps = 4, kPlanTraj = 8, kPlanState = 16, kPlanOutputs = 32 };
struct PlannerArgs {
  int B, n_steps, N_gait, k_mpc, mode, k, k_footsteps, refresh, code_scalar, q_ld;
  int xref_steps;  // horizon steps of xref to write (columns 1..xref_steps); 0 = all n_steps
  double dt_mpc, dt_wbc, T_gait, T_mpc, h_ref, k_feedback, g, L, max_height, lock_time, z_average;
  double shoulders[12], init_target[12], init_pos[12];
  const double *q7, *hv, *vref, *target_in;
  const int32_t* code;
  double* ps;  // [items][B]
  double *xref, *fsteps, *gait, *target, *feet_pva, *contacts;
};
int planner_state_items(int N_gait);

// ---- controller glue (controller_kernel.hip)
enum ControllerMode { kCtrlInit = 1, kCtrlUpdateState = 2, kCtrlWbcInputs = 3, kCtrlResult = 4, kCtrlMpcShift = 5 };
constexpr int kCtrlStItems = 58;
struct ControllerArgs {
  int B, n_steps, mode, n_gait;
  double dt_wbc, h_ref;
  const double *in0, *in1, *in2, *in3, *in4;
  double *out0, *out1, *out2, *out3, *out4;
  int32_t* iout;
  double* cs;  // [kCtrlStItems][B]
};
int controller_launch(const ControllerArgs& a, hipStream_t stream);
int control_pre_launch(const ControllerArgs& cu, const PlannerArgs& p, const ControllerArgs& cw, int with_wbc_inputs,
                       hipStream_t stream);
int planner_item_offset(int N_gait, int which);
int planner_launch(const PlannerArgs& a, hipStream_t stream);
}
