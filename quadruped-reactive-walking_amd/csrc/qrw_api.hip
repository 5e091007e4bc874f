// Host side of libqrw_hip.so: the C ABI declared in include/qrw_hip.h.
// Owns the per-instance persistent solver state in HBM and launches the gfx950 kernels.
// There is NO CPU fallback: every entry point needs a HIP device and fails loudly otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/qrw_hip.h"
#include "../../include/qrw_hip_test.h"
#include "../../include/qrw_solo12_model.h"
#include "kat_table.h"
#include "qrw_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s", what);
  g_err = buf;
  return code;
}

#define HIP_OK(expr, what)                         \
  do {                                             \
    hipError_t e__ = (expr);                       \
    if (e__ != hipSuccess) return fail(-10, what, e__); \
  } while (0)

}  // namespace

struct qrw_handle_s {
  qrw_config cfg;
  // MPC
  double* mpc_st = nullptr;
  int* mpc_gait = nullptr;
  int* mpc_flags = nullptr;
  int *mpc_iters = nullptr, *mpc_status = nullptr, *mpc_rho_updates = nullptr, *mpc_order = nullptr;
  float* mpc_ema = nullptr;  // moving average of each instance's iteration counts (longest-first block order)
  bool mpc_have_order = false;
  double *mpc_rho = nullptr, *mpc_pri = nullptr, *mpc_dua = nullptr, *mpc_prof = nullptr;
  // sequences (qrw_mpc_solve_sequence): task queue, counters, argument block re-read by the persistent kernel
  int* seq_queue = nullptr;
  size_t seq_queue_len = 0;
  unsigned* seq_ctr = nullptr;
  int* seq_hot = nullptr;
  int* seq_first = nullptr;
  int seq_groups = 0;
  // preemptive launch of qrw_mpc_solve (N > 16, more instances than resident slots): FIFO of parked solves, counters
  int* pre_queue = nullptr;
  unsigned* pre_ctr = nullptr;
  int* pause_it = nullptr;
  int pre_chunk = 0, pre_cmax = 0, pre_min_batch = 0, pre_levels = 1, pre_bin = 200;
  int pre_slots = 0;  // two-wavefront instances resident at a time (2 per compute unit)
  unsigned* pre_err_host = nullptr;  // pinned, host-mapped: the kernel's give-up code of a time-sliced launch (read on entry of the next)
  unsigned* pre_err_dev = nullptr;   // its device address
  bool force_giveup = false;         // tests (QRW_PREEMPT_FORCE_GIVEUP=1): every time-sliced launch starts with its error word set
  // WBC
  double* wbc_st = nullptr;
  int *wbc_iters = nullptr, *wbc_status = nullptr;
  int wbc_lanes16 = 1;  // full WBC compute with sixteen lanes per instance (wbc16_kernel); qrw_wbc_set_lanes / QRW_WBC16=0: one quad
  double Y[6];
  // controller glue
  double* ctrl_st = nullptr;
  // planners
  double* plan_st = nullptr;
  qrw_planner_config pcfg;
  bool plan_ready = false;
  // staging for the host-buffer entry points
  double* stage = nullptr;
  size_t stage_doubles = 0;
  int32_t* stage_i = nullptr;
  // Synchronisation is per handle, never device-wide: the *_host entry points and the getters run their copies and launches on
  // `host_stream` (private, non-blocking: it neither waits for nor stalls the legacy default stream or any other handle's
  // streams) and wait for exactly two things -- the last launch of the state family they read (wait_family) and their own copies.
  hipEvent_t order_ev = nullptr;     // qrw_stream_wait_stream (created on first use)
  qrw_iteration_buffers iter_bufs;  // qrw_iteration_bind
  bool iter_bound = false;
  hipStream_t host_stream = nullptr;
  // (atomics: qrw_stream_destroy, called from whatever thread owns the stream, clears these in every live handle)
  std::atomic<hipStream_t> last_stream[3] = {};
  std::atomic<bool> launched[3] = {};
};

namespace {
// state families of a handle: which launches a getter has to wait for
// (the controller-glue kernels write state no getter reads: their launches are not tracked)
enum Family { kFamMpc = 0, kFamWbc = 1, kFamPlan = 2, kFamCount = 3 };
inline void note_launch(qrw_handle_s* h, int fam, hipStream_t s) {
  h->last_stream[fam].store(s, std::memory_order_relaxed);
  h->launched[fam].store(true, std::memory_order_release);
}
// Wait for the family's most recent launch (whatever stream the caller gave it), nothing else on the device.  Only the LAST
// stream of a family is remembered: a caller that launches one family of one handle on several streams that are not ordered
// among themselves has to order them itself before it calls a getter.  A stream the caller has destroyed since (its work had to
// be complete for that) is answered with an error by the runtime: cleared, and the conservative device-wide wait taken instead.
hipError_t wait_family(qrw_handle_s* h, int fam) {
  if (!h->launched[fam].load(std::memory_order_acquire)) return hipSuccess;
  const hipStream_t last = h->last_stream[fam].load(std::memory_order_relaxed);
  if (last == h->host_stream) return hipSuccess;  // (host_stream work is waited for below / was at its call)
  hipError_t e = hipStreamSynchronize(last);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  h->launched[fam] = false;
  return hipDeviceSynchronize();
}
// live handles: qrw_stream_destroy forgets a destroyed stream in every handle that launched on it last
std::mutex g_handles_mutex;
std::vector<qrw_handle_s*> g_handles;
inline hipError_t d2h(qrw_handle_s* h, void* dst, const void* src, size_t bytes) {
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->host_stream);
}
inline hipError_t h2d(qrw_handle_s* h, void* dst, const void* src, size_t bytes) {
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->host_stream);
}
inline hipError_t host_done(qrw_handle_s* h) { return hipStreamSynchronize(h->host_stream); }
// Every entry point that queues copies from / to the CALLER'S host buffers (or the handle's staging area) on host_stream holds one
// of these: on ANY return -- the error paths too -- host_stream has drained, so nothing touches the caller's memory or h->stage
// after the call has come back.  On the normal path the function's own host_done() has already waited; this second
// wait on an idle stream costs ~1 us.
struct HostStreamGuard {
  qrw_handle_s* h;
  explicit HostStreamGuard(qrw_handle_s* handle) : h(handle) {}
  ~HostStreamGuard() {
    if (h && h->host_stream) (void)hipStreamSynchronize(h->host_stream);
  }
  HostStreamGuard(const HostStreamGuard&) = delete;
  HostStreamGuard& operator=(const HostStreamGuard&) = delete;
};
}  // namespace

extern "C" const char* qrw_last_error(void) { return g_err.c_str(); }

// Every entry point works on the device its handle was created on and leaves the caller's current device alone.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

static void base_inertia_diag(double Y[6]) {
  // diag of crba(q_neutral)[:6,:6] (scripts/QP_WBC.py:89-93): total mass and the composite
  // rotational inertia about the base origin with every joint at zero (all link frames axis-aligned).
  const qrw_solo12_model& Mo = QRW_SOLO12_MODEL;
  double m = 0, I[3] = {0, 0, 0};
  auto add = [&](const qrw_link_inertial& L, const double o[3]) {
    const double c[3] = {o[0] + L.com[0], o[1] + L.com[1], o[2] + L.com[2]};
    const double n2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    m += L.mass;
    I[0] += L.inertia[0] + L.mass * (n2 - c[0] * c[0]);
    I[1] += L.inertia[3] + L.mass * (n2 - c[1] * c[1]);
    I[2] += L.inertia[5] + L.mass * (n2 - c[2] * c[2]);
  };
  const double z[3] = {0, 0, 0};
  add(Mo.base, z);
  for (int l = 0; l < 4; l++) {
    const qrw_leg_model& G = Mo.leg[l];
    double o0[3], o1[3], o2[3], o3[3];
    for (int i = 0; i < 3; i++) {
      o0[i] = G.haa_xyz[i];
      o1[i] = o0[i] + G.hfe_xyz[i];
      o2[i] = o1[i] + G.kfe_xyz[i];
      o3[i] = o2[i] + G.foot_xyz[i];
    }
    add(G.shoulder, o0);
    add(G.upper, o1);
    add(G.lower, o2);
    add(G.foot, o3);
  }
  Y[0] = Y[1] = Y[2] = m;
  Y[3] = I[0];
  Y[4] = I[1];
  Y[5] = I[2];
}


// Tests only: every compute unit's LDS filled with a bit pattern (one 160 000-byte workgroup at a time per compute unit, far more
// workgroups than compute units), then the known-answer solve of mpc_solve_kernel for horizon N in launch form `mode` WITHOUT the
// per-process cache of qrw_create.  A kernel that reads LDS it has not written (an operand "multiplied by zero", a lane whose
// result is dropped later) passes on the zeros or small numbers other kernels usually leave behind and fails on NaN / Inf patterns.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned long long pattern, unsigned long long* sink) {
  __shared__ unsigned long long s[20000];
  for (int i = threadIdx.x; i < 20000; i += 256) s[i] = pattern;
  __syncthreads();
  if (sink && s[(threadIdx.x * 77 + blockIdx.x) % 20000] != pattern) *sink = 1ull;  // (keeps the stores alive)
}
// ... and every vector register (256 architectural + 256 accumulation, one 512-register wavefront per SIMD, several rounds over the
// chip): a new wavefront inherits whatever the last one left in the register file.
#include "reg_poison_asm.h"
__global__ __launch_bounds__(64, 1) void reg_poison_kernel(int* sink) {
  asm volatile(QRW_REG_POISON_ASM ::: QRW_REG_POISON_CLOBBERS);
  if (sink && threadIdx.x == 9999) *sink = 1;
}
// QRW_DEBUG_POISON_LDS=1 (diagnostic): every MPC / WBC / fused control-iteration launch of the library is preceded by that fill with NaN, on the
// launch's stream -- the whole GPU test suite can then be run with "no kernel may depend on LDS leftovers" as an extra condition.
static void debug_poison_lds(hipStream_t stream) {
  static const bool on = getenv("QRW_DEBUG_POISON_LDS") && atoi(getenv("QRW_DEBUG_POISON_LDS")) != 0;
  if (on) {
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 0, stream, ~0ull, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(reg_poison_kernel, dim3(4096), dim3(64), 0, stream, (int*)nullptr);
  }
}

// Known-answer check of THIS build of mpc_solve_kernel, run by qrw_create once per process, device and INSTANTIATION the
// handle will launch (and by qrw_selftest_sweeps for all of them).  Why it exists: the kernel lives at the edge of the register
// file, and one combination of compiler options (docs/HISTORY.md 6b) has produced a library in which every solve diverges; the parity
// tests catch that, a deployment that only rebuilds the library would not -- and each of the ten instantiations is its own piece
// of generated code (the N = 32 forms sit closest to the register limit).  Case: the reference's four-stance immobile scenario
// (scripts/test_mpc.py:54-62: height 0.2447..., feet at (+-0.195, +-0.147)), first MPC call, at the handle's own horizon N.
// Expected values per N: kat_table.h, generated from the CPU restatement by scripts/make_kat_table.py (N = 16: 350 ADMM
// iterations with one rho adaptation at iteration 200, rho 1.03905792582975e-3, four equal vertical forces summing to
// 24.5347812847 N against m g = 24.525 N, horizontal forces zero; N = 32: 375 iterations).
//   mode kKatPlain:    mpc_launch                (<1,true> N = 16, <1,false> N < 16, <2,true> N = 32, <2,false> 16 < N < 32)
//   mode kKatSliced:   mpc_preemptive_launch     (N > 16: the solve is cut after 200 iterations, parked, taken from the queue by
//                                                 another workgroup and finished -- the time-sliced form config 4 runs by default)
//   mode kKatSequence: mpc_sequence_launch, K = 1 (the SEQ instantiations of qrw_mpc_solve_sequence)
enum KatMode { kKatPlain = 0, kKatSliced = 1, kKatSequence = 2, kKatModes = 3 };
static const char* const kKatModeName[kKatModes] = {"one launch per call", "time-sliced launch", "sequence launch"};
struct KatResult { int iters = -1, status = 0, want_iters = 0; double err = 0.0, rho = 0.0, want_rho = 0.0; };
static int mpc_known_answer_check(int N, int mode, KatResult* res) {
  if (N < 1 || N > qrw::kMpcMaxN || mode < 0 || mode >= kKatModes || (mode == kKatSliced && N <= 16)) return -13;
  const int Ng = N > 20 ? N : 20, T = qrw::mpc_threads(N);
  const qrw::KatRow& want = qrw::kKatTable[N];
  std::vector<double> hx(12 * (N + 1), 0.0), hf((size_t)Ng * 12, 0.0), hout(24 * N);
  for (int c = 0; c <= N; c++) hx[2 * (N + 1) + c] = 0.24474949993103629;
  const double feet[12] = {0.195, 0.147, 0., 0.195, -0.147, 0., -0.195, 0.147, 0., -0.195, -0.147, 0.};
  for (int k = 0; k < N; k++) for (int i = 0; i < 12; i++) hf[k * 12 + i] = feet[i];
  struct Buf { void* p = nullptr; ~Buf() { if (p) hipFree(p); } };
  Buf bx, bf, bo, bst, bg, bi, bd, bq, bc;
  const size_t n_int = 16, n_dbl = 16;
  constexpr int kChunk = 200, kCmax = 4000 / kChunk, kLevels = 2;  // sliced: one parked solve at a time, 19 queue-fed workgroups
  const size_t q_ints = (size_t)kLevels * (kCmax - 1) + 8;         // (also covers the sequence launch's one-task queue)
  const size_t c_words = qrw::kPreCtrWords > qrw::kSeqQctrWords ? qrw::kPreCtrWords : qrw::kSeqQctrWords;
  if (hipMalloc(&bx.p, hx.size() * sizeof(double)) != hipSuccess || hipMalloc(&bf.p, hf.size() * sizeof(double)) != hipSuccess ||
      hipMalloc(&bo.p, hout.size() * sizeof(double)) != hipSuccess || hipMalloc(&bst.p, (size_t)qrw::kMpcStItems * T * sizeof(double)) != hipSuccess ||
      hipMalloc(&bg.p, (size_t)Ng * 4 * sizeof(int)) != hipSuccess || hipMalloc(&bi.p, n_int * sizeof(int)) != hipSuccess ||
      hipMalloc(&bd.p, n_dbl * sizeof(double)) != hipSuccess || hipMalloc(&bq.p, q_ints * sizeof(int)) != hipSuccess ||
      hipMalloc(&bc.p, c_words * sizeof(unsigned)) != hipSuccess)
    return -10;
  // a private non-blocking stream: the check neither waits for nor stalls the work other streams of the process have queued
  // (the fills and copies below are ordered before the launch by the stream itself)
  struct Stream { hipStream_t s = nullptr; ~Stream() { if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); } } } st;
  if (hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking) != hipSuccess) return -10;
  hipMemcpyAsync(bx.p, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice, st.s);
  hipMemcpyAsync(bf.p, hf.data(), hf.size() * sizeof(double), hipMemcpyHostToDevice, st.s);
  hipMemsetAsync(bo.p, 0xFF, hout.size() * sizeof(double), st.s);
  hipMemsetAsync(bst.p, 0, (size_t)qrw::kMpcStItems * T * sizeof(double), st.s);
  hipMemsetAsync(bg.p, 0, (size_t)Ng * 4 * sizeof(int), st.s);
  hipMemsetAsync(bi.p, 0, n_int * sizeof(int), st.s);
  hipMemsetAsync(bd.p, 0, n_dbl * sizeof(double), st.s);
  hipMemsetAsync(bq.p, 0xFF, q_ints * sizeof(int), st.s);
  hipMemsetAsync(bc.p, 0, c_words * sizeof(unsigned), st.s);
  qrw::MpcArgs a;
  memset(&a, 0, sizeof(a));
  a.B = 1; a.N = N; a.N_gait = Ng; a.dt = 0.02;
  a.xref = (const double*)bx.p; a.fsteps = (const double*)bf.p; a.num_iter = nullptr; a.num_iter_scalar = 0;
  a.out = (double*)bo.p; a.st = (double*)bst.p; a.gait = (int*)bg.p;
  int* ip = (int*)bi.p; double* dp = (double*)bd.p;
  a.flags = ip; a.iters = ip + 1; a.status = ip + 2; a.rho_updates = ip + 3;
  a.rho_out = dp; a.pri = dp + 1; a.dua = dp + 2; a.prof = nullptr; a.order = nullptr;
  int lrc;
  if (mode == kKatSliced) {
    a.pre_chunk = kChunk; a.pre_cmax = kCmax; a.pre_cap = kCmax - 1; a.pre_levels = kLevels; a.pre_bin = 200;
    a.pre_queue = (int*)bq.p; a.pre_ctr = (unsigned*)bc.p; a.pause_it = ip + 8;
    lrc = qrw::mpc_preemptive_launch(a, /*resident_slots=*/1 << 20, st.s);  // (one instance: always the two-grid form)
  } else if (mode == kKatSequence) {
    a.seq_K = 1; a.queue = (int*)bq.p; a.qctr = (unsigned*)bc.p; a.seq_hot = ip + 9; a.seq_first = ip + 10; a.seq_groups = 1; a.seq_iters = ip + 11;
    lrc = qrw::mpc_sequence_launch(a, st.s);
  } else {
#ifdef QRW_DEBUG_POISON
    if (const char* sel = getenv("QRW_DEBUG_POISON_SEL")) a.pre_bin = atoi(sel);  // diagnostic build: mpc_kernel.hip, top of the kernel
#endif
    lrc = qrw::mpc_launch(a, st.s);
  }
  if (lrc != 0) return -11;
  int hi[n_int]; double hd[n_dbl]; unsigned hc[8] = {0};
  hipMemcpyAsync(hi, bi.p, sizeof(hi), hipMemcpyDeviceToHost, st.s);
  hipMemcpyAsync(hd, bd.p, sizeof(hd), hipMemcpyDeviceToHost, st.s);
  hipMemcpyAsync(hout.data(), bo.p, hout.size() * sizeof(double), hipMemcpyDeviceToHost, st.s);
  if (mode == kKatSliced) hipMemcpyAsync(hc, (unsigned*)bc.p + qrw::kPreErrWord, sizeof(unsigned), hipMemcpyDeviceToHost, st.s);
  if (mode == kKatSequence) hipMemcpyAsync(hc, (unsigned*)bc.p + qrw::kSeqErrWord, sizeof(unsigned), hipMemcpyDeviceToHost, st.s);
  if (hipStreamSynchronize(st.s) != hipSuccess) return -12;
  double err = 0.0, fz = 0.0;
  for (int j = 0; j < 4; j++) {
    fz += hout[(12 + 3 * j + 2) * N];
    err = fmax(err, fabs(hout[(12 + 3 * j + 2) * N] - hout[14 * N]));      // equal vertical forces
    err = fmax(err, fabs(hout[(12 + 3 * j) * N]) + fabs(hout[(12 + 3 * j + 1) * N]));  // no horizontal force
  }
  err = fmax(err, fabs(fz - want.fz));
  err = fmax(err, fabs(hd[0] / want.rho - 1.0));
  if (hc[0] != 0) err = 1e300;       // the queue of the time-sliced / sequence launch gave up
  if (!(err == err)) err = 1e300;  // NaN
  if (res) { res->iters = hi[1]; res->status = hi[2]; res->err = err; res->rho = hd[0]; res->want_iters = want.iters; res->want_rho = want.rho; }
  // The pass criterion is the ANSWER (status solved, forces and rho to 1e-8).  The iteration count is reported, and
  // only checked loosely: the table's value with the shipped toolchain; a legitimate compiler change may move a borderline
  // termination test by one check interval (25), a miscompiled kernel diverges (status != solved) or lands far away.
  return (hi[2] == qrw::kStatusSolved && err < 1e-8 && hi[1] >= want.iters - 50 && hi[1] <= want.iters + 50) ? 0 : 1;
}
// per device, horizon and launch form: 0 not run, 1 passed, -1 failed; guarded by a mutex (qrw_create may be called from several threads)
static std::mutex g_kat_mutex;
static signed char g_kat_state[64][qrw::kMpcMaxN + 1][kKatModes] = {};
static KatResult g_kat_last;  // result of the most recent check (read under the mutex by the caller that ran it)
static int mpc_known_answer_once(int device, int N, int mode, KatResult* out) {
  if (device < 0 || device >= 64 || N < 1 || N > qrw::kMpcMaxN) return 0;
  std::lock_guard<std::mutex> lock(g_kat_mutex);
  signed char& state = g_kat_state[device][N][mode];
  if (state == 0) {
    const char* skip = getenv("QRW_SKIP_SELFTEST");  // escape hatch (e.g. bring-up on a new toolchain): say so, loudly
    if (skip && skip[0] == '1') {
      fprintf(stderr, "libqrw_hip: QRW_SKIP_SELFTEST=1, the known-answer self-test of mpc_solve_kernel is SKIPPED\n");
      state = 1;
    } else {
      const int rc = mpc_known_answer_check(N, mode, &g_kat_last);
      if (rc < 0) g_kat_last.status = rc;
      state = (rc == 0) ? 1 : -1;
    }
  }
  if (out) *out = g_kat_last;
  return state == 1 ? 0 : 1;
}
static int kat_fail(const char* who, int N, int mode, const KatResult& r) {
  char msg[640];
  snprintf(msg, sizeof(msg),
           "%s: the known-answer self-test of mpc_solve_kernel failed on this device for horizon N = %d, %s (got %d ADMM iterations, "
           "status %d, rho %.10g, error %.3g; expected %d, solved, %.10g, < 1e-8): this build of libqrw_hip.so computes wrong results "
           "(a code-generation problem seen with non-shipped compiler options, docs/HISTORY.md 6b); rebuild with the Makefile's flags "
           "(QRW_SKIP_SELFTEST=1 skips this check)", who, N, kKatModeName[mode], r.iters, r.status, r.rho, r.err, r.want_iters, r.want_rho);
  return fail(-20, msg);
}

// the launch form qrw_mpc_solve uses for this handle (decided once, at creation)
static bool mpc_time_sliced(const qrw_handle_s* h) { return h->pre_chunk > 0 && h->cfg.batch > h->pre_min_batch && h->pre_cmax > 1; }

extern "C" int qrw_create(const qrw_config* cfg, qrw_handle* out) {
  if (!cfg || !out) return fail(-1, "qrw_create: null argument");
  if (cfg->batch < 1) return fail(-1, "qrw_create: batch must be >= 1");
  if (cfg->n_steps < 1 || cfg->n_steps > qrw::kMpcMaxN)
    return fail(-1, "qrw_create: n_steps must be in 1..32 (16 horizon steps per wavefront, at most two wavefronts)");
  if (cfg->N_gait < cfg->n_steps) return fail(-1, "qrw_create: N_gait must be >= n_steps");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(-2, "qrw_create: no HIP device (this library has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(-2, "qrw_create: bad device ordinal");
  DeviceScope dev_scope__(cfg->device);
  qrw_handle h = new qrw_handle_s();
  h->cfg = *cfg;
  const size_t B = (size_t)cfg->batch;
  const int N = cfg->n_steps;
  if (hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking) != hipSuccess) {
    h->host_stream = nullptr;
    qrw_destroy(h);
    return fail(-10, "qrw_create: hipStreamCreateWithFlags", hipGetLastError());
  }
#define ALLOC(ptr, bytes)                                   \
  do {                                                      \
    hipError_t e__ = hipMalloc((void**)&(ptr), (bytes));    \
    if (e__ != hipSuccess) {                                \
      qrw_destroy(h);                                       \
      return fail(-10, "hipMalloc " #ptr, e__);             \
    }                                                       \
    hipMemsetAsync((ptr), 0, (bytes), h->host_stream);      \
  } while (0)
  ALLOC(h->mpc_st, B * qrw::kMpcStItems * qrw::mpc_threads(N) * sizeof(double));
  ALLOC(h->mpc_gait, B * cfg->N_gait * 4 * sizeof(int));
  ALLOC(h->mpc_flags, B * sizeof(int));
  ALLOC(h->mpc_iters, B * sizeof(int));
  ALLOC(h->mpc_status, B * sizeof(int));
  ALLOC(h->mpc_rho_updates, B * sizeof(int));
  ALLOC(h->mpc_order, B * sizeof(int));
  ALLOC(h->mpc_ema, B * sizeof(float));
  ALLOC(h->mpc_rho, B * sizeof(double));
  ALLOC(h->mpc_pri, B * sizeof(double));
  ALLOC(h->mpc_dua, B * sizeof(double));
  ALLOC(h->mpc_prof, B * qrw::kMpcProfItems * sizeof(double));
  if (N > 16) {
    // round-robin time slicing of the solves of one call (mpc_kernel.hip, PRE): slices of QRW_PREEMPT_CHUNK iterations
    // (default 600, rounded up to a multiple of 200; 0 switches it off), used when the batch exceeds the resident slots
    const char* ce = getenv("QRW_PREEMPT_CHUNK");
    int chunk = ce ? atoi(ce) : 600;
    if (chunk > 0) {
      chunk = ((chunk + 199) / 200) * 200;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) { qrw_destroy(h); return fail(-10, "hipGetDeviceProperties"); }
      h->pre_chunk = chunk;
      h->pre_cmax = (4000 + chunk - 1) / chunk;
      h->pre_min_batch = 2 * prop.multiProcessorCount;  // two two-wavefront instances per compute unit are resident
      h->pre_slots = h->pre_min_batch;
      if (const char* mb = getenv("QRW_PREEMPT_MIN_BATCH")) h->pre_min_batch = atoi(mb);  // tests: slice small batches too
      ALLOC(h->pause_it, B * sizeof(int));
      ALLOC(h->pre_ctr, qrw::kPreCtrWords * sizeof(unsigned));
      // priority levels of the parked solves (mpc_kernel.hip, PRE): QRW_PREEMPT_LEVELS = 1 is one FIFO (plain round robin),
      // the default 9 = first-slice FIFO + 8 levels of QRW_PREEMPT_BIN (200) predicted remaining iterations each (everything beyond
      // 1400 shares the first of them; 400 and 200 measure 91.4 k and 91.9 k steps/s on one box, the simulation 1.033 and 1.028)
      h->pre_levels = qrw::kPreMaxLevels;
      if (const char* le = getenv("QRW_PREEMPT_LEVELS")) h->pre_levels = atoi(le);
      if (h->pre_levels < 1) h->pre_levels = 1;
      if (h->pre_levels > qrw::kPreMaxLevels) h->pre_levels = qrw::kPreMaxLevels;
      if (const char* be = getenv("QRW_PREEMPT_BIN")) h->pre_bin = atoi(be);
      if (h->pre_bin < 25) h->pre_bin = 25;
      ALLOC(h->pre_queue, (size_t)h->pre_levels * B * (size_t)(h->pre_cmax > 1 ? h->pre_cmax - 1 : 1) * sizeof(int));
      // the launch's error word, mirrored where the host can read it without synchronising the device
      if (hipHostMalloc((void**)&h->pre_err_host, 64, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer((void**)&h->pre_err_dev, h->pre_err_host, 0) != hipSuccess) {
        qrw_destroy(h);
        return fail(-10, "qrw_create: hipHostMalloc of the time-sliced launch's error word");
      }
      h->pre_err_host[0] = 0u;
      // tests: the launch starts with its error word set, so every queue-fed workgroup leaves at once ("somebody gave up
      // already") and no parked solve is ever finished -- the state a queue that gave up leaves behind, without waiting 2 s for it
      if (const char* ge = getenv("QRW_PREEMPT_FORCE_GIVEUP")) h->force_giveup = (ge[0] == '1');
    }
  }
  {  // known answer through the instantiation of mpc_solve_kernel this handle's qrw_mpc_solve will launch
    const int mode = mpc_time_sliced(h) ? kKatSliced : kKatPlain;
    KatResult r;
    if (mpc_known_answer_once(cfg->device, N, mode, &r) != 0) {
      qrw_destroy(h);
      return kat_fail("qrw_create", N, mode, r);
    }
  }
  ALLOC(h->wbc_st, B * qrw::kWbcStItems * sizeof(double));
  ALLOC(h->wbc_iters, B * sizeof(int));
  ALLOC(h->wbc_status, B * sizeof(int));
  ALLOC(h->ctrl_st, B * (size_t)qrw::kCtrlStItems * sizeof(double));
  ALLOC(h->plan_st, B * (size_t)qrw::planner_state_items(cfg->N_gait) * sizeof(double));
  // staging: the largest host-API call moves M (324) + Jc (216) + ... per instance
  h->stage_doubles = B * (size_t)(12 * (N + 1) + cfg->N_gait * 12 + 24 * N + 1024);
  ALLOC(h->stage, h->stage_doubles * sizeof(double));
  ALLOC(h->stage_i, B * sizeof(int32_t));
#undef ALLOC
  base_inertia_diag(h->Y);
  if (const char* e16 = getenv("QRW_WBC16")) h->wbc_lanes16 = (e16[0] != '0');
  {  // the zero fills above are complete before the caller can launch on any stream (this handle's own work only: no device-wide wait)
    const hipError_t e = host_done(h);
    if (e != hipSuccess) { qrw_destroy(h); return fail(-10, "qrw_create sync", e); }
  }
  {
    std::lock_guard<std::mutex> lock(g_handles_mutex);
    g_handles.push_back(h);
  }
  *out = h;
  return 0;
}

extern "C" int qrw_destroy(qrw_handle h) {
  if (!h) return 0;
  {
    std::lock_guard<std::mutex> lock(g_handles_mutex);
    for (size_t i = 0; i < g_handles.size(); i++)
      if (g_handles[i] == h) { g_handles.erase(g_handles.begin() + i); break; }
  }
  DeviceScope dev_scope__(h->cfg.device);
  hipFree(h->mpc_st); hipFree(h->mpc_gait); hipFree(h->mpc_flags); hipFree(h->mpc_iters);
  hipFree(h->mpc_status); hipFree(h->mpc_rho_updates); hipFree(h->mpc_order); hipFree(h->mpc_ema); hipFree(h->mpc_rho); hipFree(h->mpc_pri);
  hipFree(h->mpc_dua); hipFree(h->mpc_prof); hipFree(h->wbc_st); hipFree(h->wbc_iters); hipFree(h->wbc_status);
  hipFree(h->plan_st); hipFree(h->ctrl_st);
  hipFree(h->seq_queue); hipFree(h->seq_ctr); hipFree(h->seq_hot); hipFree(h->seq_first);
  hipFree(h->pre_queue); hipFree(h->pre_ctr); hipFree(h->pause_it);
  if (h->pre_err_host) hipHostFree(h->pre_err_host);
  hipFree(h->stage); hipFree(h->stage_i);
  if (h->host_stream) hipStreamDestroy(h->host_stream);
  if (h->order_ev) hipEventDestroy(h->order_ev);
  delete h;
  return 0;
}

extern "C" int64_t qrw_state_bytes(qrw_handle h) {
  if (!h) return 0;
  const int64_t B = h->cfg.batch;
  return B * ((int64_t)qrw::kMpcStItems * qrw::mpc_threads(h->cfg.n_steps) * 8 + h->cfg.N_gait * 16 + 6 * 4 + 3 * 8 + qrw::kWbcStItems * 8 + 8);
}

extern "C" int qrw_mpc_solve(qrw_handle h, const double* d_xref, const double* d_fsteps, const int32_t* d_num_iter,
                             int32_t num_iter_scalar, double* d_out, void* stream) {
  if (!h || !d_xref || !d_fsteps || !d_out) return fail(-1, "qrw_mpc_solve: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  if (h->pre_err_host) {
    // A time-sliced launch whose queue gave up (never expected) wrote its code into a host-mapped word: seen here, at the next
    // call that follows the failed launch's end, without a device sync.  Reported ONCE (this call launches nothing); the
    // unfinished instances' results of that launch are NaN, and the next solve starts them cold (mpc_kernel.hip, `aborted`).
    const unsigned code = __atomic_exchange_n(h->pre_err_host, 0u, __ATOMIC_ACQ_REL);
    if (code != 0u) {
      char msg[256];
      snprintf(msg, sizeof(msg), "qrw_mpc_solve: an earlier time-sliced MPC launch of this handle gave up (code %u: %s); its unfinished "
               "instances hold NaN results and restart cold at the next call", code,
               code == 2u ? "a priority level's queue overran" : code == 9u ? "forced by QRW_PREEMPT_FORCE_GIVEUP" : "a parked solve reserved for a taker workgroup did not arrive");
      return fail(-12, msg);
    }
  }
  qrw::MpcArgs a;
  memset(&a, 0, sizeof(a));
  a.B = h->cfg.batch; a.N = h->cfg.n_steps; a.N_gait = h->cfg.N_gait; a.dt = h->cfg.dt_mpc;
  a.xref = d_xref; a.fsteps = d_fsteps; a.num_iter = d_num_iter; a.num_iter_scalar = num_iter_scalar;
  a.out = d_out; a.st = h->mpc_st; a.gait = h->mpc_gait; a.flags = h->mpc_flags; a.iters = h->mpc_iters;
  a.status = h->mpc_status; a.rho_out = h->mpc_rho; a.pri = h->mpc_pri; a.dua = h->mpc_dua;
  a.rho_updates = h->mpc_rho_updates;
  a.prof = h->mpc_prof;
  a.order = h->mpc_have_order ? h->mpc_order : nullptr;
  if (mpc_time_sliced(h)) {
    // more instances than resident slots at N > 16: the solves are time-sliced round robin inside the launch (same results)
    a.pre_chunk = h->pre_chunk; a.pre_cmax = h->pre_cmax; a.pre_cap = h->cfg.batch * (h->pre_cmax - 1);
    a.pre_queue = h->pre_queue; a.pre_ctr = h->pre_ctr; a.pause_it = h->pause_it;
    a.pre_levels = h->pre_levels; a.pre_bin = h->pre_bin;
    HIP_OK(hipMemsetAsync(h->pre_queue, 0xFF, (size_t)a.pre_levels * a.pre_cap * sizeof(int), (hipStream_t)stream), "qrw_mpc_solve: queue reset");
    HIP_OK(hipMemsetAsync(h->pre_ctr, 0, qrw::kPreCtrWords * sizeof(unsigned), (hipStream_t)stream), "qrw_mpc_solve: counter reset");
    // a solve that a given-up queue left unfinished (never expected; qrw_mpc_get_stats reports it) must not leave the previous
    // call's numbers in the caller's buffer: NaN (all-ones bytes) until the finishing slice writes the result (~10 us per call)
    HIP_OK(hipMemsetAsync(d_out, 0xFF, (size_t)h->cfg.batch * 24 * (size_t)h->cfg.n_steps * sizeof(double), (hipStream_t)stream),
           "qrw_mpc_solve: result prefill");
    if (h->force_giveup) {
      static const unsigned forced = 9u;
      HIP_OK(hipMemcpyAsync(h->pre_ctr + qrw::kPreErrWord, &forced, sizeof(unsigned), hipMemcpyHostToDevice, (hipStream_t)stream),
             "qrw_mpc_solve: forced give-up");
    }
    debug_poison_lds((hipStream_t)stream);
    if (qrw::mpc_preemptive_launch(a, h->pre_slots, (hipStream_t)stream) != 0 ||
        qrw::mpc_pre_error_flush(h->pre_ctr, h->pre_err_dev, (hipStream_t)stream) != 0)
      return fail(-11, "qrw_mpc_solve: kernel launch failed", hipGetLastError());
  } else {
    debug_poison_lds((hipStream_t)stream);
    if (qrw::mpc_launch(a, (hipStream_t)stream) != 0) return fail(-11, "qrw_mpc_solve: kernel launch failed", hipGetLastError());
  }
  note_launch(h, kFamMpc, (hipStream_t)stream);
  // next launch's block order = this solve's iteration counts, longest first (same stream: ordered after the solve)
  static const int order_min = getenv("QRW_ORDER_MIN") ? atoi(getenv("QRW_ORDER_MIN")) : 1024;  // experiments only
  if (h->cfg.batch > order_min) {
    if (qrw::mpc_order_launch(h->mpc_iters, h->mpc_ema, h->mpc_order, h->cfg.batch, (hipStream_t)stream) != 0)
      return fail(-11, "qrw_mpc_solve: order kernel launch failed", hipGetLastError());
    h->mpc_have_order = true;
  }
  return 0;
}

extern "C" int qrw_mpc_solve_host(qrw_handle h, const double* h_xref, const double* h_fsteps, const int32_t* h_num_iter,
                                  int32_t num_iter_scalar, double* h_out) {
  if (!h || !h_xref || !h_fsteps || !h_out) return fail(-1, "qrw_mpc_solve_host: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch, N = h->cfg.n_steps, Ng = h->cfg.N_gait;
  double* dx = h->stage;
  double* df = dx + B * 12 * (N + 1);
  double* dout = df + B * Ng * 12;
  HIP_OK(wait_family(h, kFamMpc), "qrw_mpc_solve_host: earlier solve of this handle");
  HIP_OK(h2d(h, dx, h_xref, B * 12 * (N + 1) * sizeof(double)), "H2D xref");
  HIP_OK(h2d(h, df, h_fsteps, B * Ng * 12 * sizeof(double)), "H2D fsteps");
  if (h_num_iter) HIP_OK(h2d(h, h->stage_i, h_num_iter, B * sizeof(int32_t)), "H2D num_iter");
  int rc = qrw_mpc_solve(h, dx, df, h_num_iter ? h->stage_i : nullptr, num_iter_scalar, dout, (void*)h->host_stream);
  if (rc) return rc;
  HIP_OK(d2h(h, h_out, dout, B * 24 * N * sizeof(double)), "D2H result");
  HIP_OK(host_done(h), "qrw_mpc_solve_host sync");
  return 0;
}

extern "C" int qrw_mpc_solve_sequence(qrw_handle h, int32_t K, const double* d_xref, const double* d_fsteps, int32_t first_num_iter,
                                      double* d_out, int32_t* d_iters, void* stream) {
  if (!h || !d_xref || !d_fsteps || !d_out || K < 1) return fail(-1, "qrw_mpc_solve_sequence: bad argument");
  DeviceScope dev_scope__(h->cfg.device);
  {  // the SEQ instantiation's known answer, once per process, device and horizon
    KatResult r;
    if (mpc_known_answer_once(h->cfg.device, h->cfg.n_steps, kKatSequence, &r) != 0) return kat_fail("qrw_mpc_solve_sequence", h->cfg.n_steps, kKatSequence, r);
  }
  const size_t need = (size_t)K * (size_t)h->cfg.batch;
  if (need > (size_t)0x7fffffff) return fail(-1, "qrw_mpc_solve_sequence: K * batch too large");
  if (need > h->seq_queue_len) {  // grown on demand (first call / longer sequence), never on a steady-state call
    HIP_OK(hipStreamSynchronize((hipStream_t)stream), "qrw_mpc_solve_sequence sync");
    hipFree(h->seq_queue);
    h->seq_queue = nullptr;
    h->seq_queue_len = 0;
    if (hipMalloc((void**)&h->seq_queue, need * sizeof(int)) != hipSuccess) return fail(-10, "qrw_mpc_solve_sequence: hipMalloc queue");
    h->seq_queue_len = need;
  }
  if (!h->seq_ctr || !h->seq_first || !h->seq_hot) {  // (each on its own: a call that failed half-way is retried from where it failed)
    if (!h->seq_ctr && hipMalloc((void**)&h->seq_ctr, qrw::kSeqQctrWords * sizeof(unsigned)) != hipSuccess) { h->seq_ctr = nullptr; return fail(-10, "qrw_mpc_solve_sequence: hipMalloc counters"); }
    if (!h->seq_first && hipMalloc((void**)&h->seq_first, (size_t)h->cfg.batch * sizeof(int)) != hipSuccess) { h->seq_first = nullptr; return fail(-10, "qrw_mpc_solve_sequence: hipMalloc first tasks"); }
    if (!h->seq_hot && hipMalloc((void**)&h->seq_hot, (size_t)h->cfg.batch * sizeof(int)) != hipSuccess) { h->seq_hot = nullptr; return fail(-10, "qrw_mpc_solve_sequence: hipMalloc hot flags"); }
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, h->cfg.device), "hipGetDeviceProperties");
    // resident workgroups: one 512-register wavefront per SIMD (N <= 16: 4 workgroups per CU; N > 16: 2 of two wavefronts).
    // Only a scheduling hint: the first seq_groups workgroups take call 0 of the seq_groups longest-ranked instances by
    // index instead of through the queues.  On a CU-masked stream (or beside another stream group) fewer workgroups are
    // resident at a time; the dealt ones then simply start in several rounds.  What forward progress does rest on is that
    // workgroups START IN INDEX ORDER (every workgroup with a dealt task starts before any queue-fed one can hold the
    // last free slot) -- how the hardware dispatcher works, and what the one-call kernel's longest-first order already
    // assumes; tests/test_gpu_mpc.py runs a sequence on a masked stream beside a second stream to pin it.
    h->seq_groups = prop.multiProcessorCount * (h->cfg.n_steps <= 16 ? 4 : 2);
  }
  qrw::MpcArgs a;
  memset(&a, 0, sizeof(a));
  a.B = h->cfg.batch; a.N = h->cfg.n_steps; a.N_gait = h->cfg.N_gait; a.dt = h->cfg.dt_mpc;
  a.xref = d_xref; a.fsteps = d_fsteps; a.num_iter = nullptr; a.num_iter_scalar = first_num_iter;
  a.out = d_out; a.st = h->mpc_st; a.gait = h->mpc_gait; a.flags = h->mpc_flags; a.iters = h->mpc_iters;
  a.status = h->mpc_status; a.rho_out = h->mpc_rho; a.pri = h->mpc_pri; a.dua = h->mpc_dua;
  a.rho_updates = h->mpc_rho_updates;
  a.prof = h->mpc_prof;  // only diagnostic builds (-DQRW_SEQ_STATS) write it
  a.order = h->mpc_have_order ? h->mpc_order : nullptr;
  a.seq_K = K; a.queue = h->seq_queue; a.qctr = h->seq_ctr; a.seq_hot = h->seq_hot; a.seq_first = h->seq_first; a.seq_iters = d_iters;
  a.seq_groups = h->seq_groups < h->cfg.batch ? h->seq_groups : h->cfg.batch;
  // a task that never runs (queue give-up, see qrw_mpc_sequence_error) must not leave plausible numbers behind: results
  // are pre-filled with NaN (all-ones bytes) and the iteration counts with -1
  HIP_OK(hipMemsetAsync(d_out, 0xFF, need * 24 * (size_t)h->cfg.n_steps * sizeof(double), (hipStream_t)stream), "qrw_mpc_solve_sequence prefill");
  if (d_iters) HIP_OK(hipMemsetAsync(d_iters, 0xFF, need * sizeof(int32_t), (hipStream_t)stream), "qrw_mpc_solve_sequence prefill iters");
  debug_poison_lds((hipStream_t)stream);
  if (qrw::mpc_sequence_launch(a, (hipStream_t)stream) != 0)
    return fail(-11, "qrw_mpc_solve_sequence: kernel launch failed", hipGetLastError());
  note_launch(h, kFamMpc, (hipStream_t)stream);
  if (h->cfg.batch > 1024) {
    if (qrw::mpc_order_launch(h->mpc_iters, h->mpc_ema, h->mpc_order, h->cfg.batch, (hipStream_t)stream) != 0)
      return fail(-11, "qrw_mpc_solve_sequence: order kernel launch failed", hipGetLastError());
    h->mpc_have_order = true;
  }
  return 0;
}

extern "C" int qrw_mpc_sequence_error(qrw_handle h, int32_t* timed_out) {
  if (!h || !timed_out) return fail(-1, "qrw_mpc_sequence_error: null argument");
  DeviceScope dev_scope__(h->cfg.device);
  HostStreamGuard host_guard__(h);
  *timed_out = 0;
  if (!h->seq_ctr) return 0;
  // like every other getter: waits for this handle's last MPC launch (the sequence in flight), not for the device
  HIP_OK(wait_family(h, kFamMpc), "qrw_mpc_sequence_error sync");
  unsigned c[qrw::kSeqQctrWords];
  HIP_OK(d2h(h, c, h->seq_ctr, sizeof(c)), "qrw_mpc_sequence_error");
  HIP_OK(host_done(h), "qrw_mpc_sequence_error");
  *timed_out = (int32_t)c[qrw::kSeqErrWord];
  if (getenv("QRW_SEQ_STATS")) {
    fprintf(stderr, "qrw sequence: levels taken/queued");
    for (int l = 0; l < qrw::kSeqErrWord / 16; l++) fprintf(stderr, " %u/%u", c[16 * l], c[16 * l + 1]);
    fprintf(stderr, ", looking-for-work ticks (100 MHz, summed) %u\n", c[qrw::kSeqErrWord + 1]);
  }
  return 0;
}

extern "C" int qrw_mpc_copy_iters(qrw_handle h, int32_t* d_iters, void* stream) {
  if (!h || !d_iters) return fail(-1, "qrw_mpc_copy_iters: null argument");
  DeviceScope dev_scope__(h->cfg.device);
  HIP_OK(hipMemcpyAsync(d_iters, h->mpc_iters, (size_t)h->cfg.batch * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream),
         "qrw_mpc_copy_iters");
  return 0;
}

extern "C" int qrw_mpc_get_gait(qrw_handle h, int32_t b, double* h_gait, double* h_Sgait) {
  if (!h || b < 0 || b >= h->cfg.batch) return fail(-1, "qrw_mpc_get_gait: bad argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const int N = h->cfg.n_steps, Ng = h->cfg.N_gait;
  HIP_OK(wait_family(h, kFamMpc), "sync");
  if (h_gait) {
    std::vector<int> g(Ng * 4);
    HIP_OK(d2h(h, g.data(), h->mpc_gait + (size_t)b * Ng * 4, Ng * 4 * sizeof(int)), "D2H gait");
    HIP_OK(host_done(h), "D2H gait");
    for (int i = 0; i < Ng * 4; i++) h_gait[i] = (double)g[i];
  }
  if (h_Sgait) {
    const int T = qrw::mpc_threads(N);
    std::vector<double> s(3 * T);
    HIP_OK(d2h(h, s.data(), h->mpc_st + ((size_t)b * qrw::kMpcStItems + qrw::kStS) * T, 3 * T * sizeof(double)), "D2H S");
    HIP_OK(host_done(h), "D2H S");
    for (int k = 0; k < N; k++)
      for (int j = 0; j < 4; j++)
        for (int t = 0; t < 3; t++) h_Sgait[12 * k + 3 * j + t] = s[t * T + 4 * k + j];
  }
  return 0;
}

extern "C" int qrw_mpc_get_stats(qrw_handle h, int32_t* h_iters, int32_t* h_status, double* h_rho, double* h_pri_res,
                                 double* h_dua_res) {
  if (!h) return fail(-1, "qrw_mpc_get_stats: null handle");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch;
  HIP_OK(wait_family(h, kFamMpc), "sync");  // this handle's last solve only: another handle's launch in flight is not waited for
  if (h->pre_ctr) {  // a time-sliced launch whose queue gave up (never expected) left solves unfinished: say so, loudly
    unsigned c[qrw::kPreCtrWords];
    HIP_OK(d2h(h, c, h->pre_ctr, sizeof(c)), "D2H pre_ctr");
    HIP_OK(host_done(h), "D2H pre_ctr");
    if (c[qrw::kPreErrWord] != 0) return fail(-12, "qrw_mpc_get_stats: the last time-sliced MPC launch left solves unfinished (a parked solve did not reach its taker "
                                     "within 2 s, or a queue overran); results of that call are incomplete");
  }
  if (h_iters) HIP_OK(d2h(h, h_iters, h->mpc_iters, B * sizeof(int)), "D2H iters");
  if (h_status) HIP_OK(d2h(h, h_status, h->mpc_status, B * sizeof(int)), "D2H status");
  if (h_rho) HIP_OK(d2h(h, h_rho, h->mpc_rho, B * sizeof(double)), "D2H rho");
  if (h_pri_res) HIP_OK(d2h(h, h_pri_res, h->mpc_pri, B * sizeof(double)), "D2H pri");
  if (h_dua_res) HIP_OK(d2h(h, h_dua_res, h->mpc_dua, B * sizeof(double)), "D2H dua");
  HIP_OK(host_done(h), "qrw_mpc_get_stats");
  return 0;
}

// Diagnostic: bookkeeping of the last time-sliced qrw_mpc_solve (N > 16, batch above the resident slots; zeros otherwise).
extern "C" int qrw_mpc_get_slice_stats(qrw_handle h, int32_t* levels, int32_t* chunk, uint32_t* h_parks_per_level /* [9] */,
                                       uint32_t* h_takers, uint32_t* h_finished) {
  if (!h || !levels || !chunk || !h_parks_per_level || !h_takers || !h_finished) return fail(-1, "qrw_mpc_get_slice_stats: null argument");
  DeviceScope dev_scope__(h->cfg.device);
  HostStreamGuard host_guard__(h);
  *levels = h->pre_ctr ? h->pre_levels : 0;
  *chunk = h->pre_ctr ? h->pre_chunk : 0;
  for (int l = 0; l < qrw::kPreMaxLevels; l++) h_parks_per_level[l] = 0;
  *h_takers = *h_finished = 0;
  if (!h->pre_ctr) return 0;
  HIP_OK(wait_family(h, kFamMpc), "qrw_mpc_get_slice_stats sync");
  unsigned c[qrw::kPreCtrWords];
  HIP_OK(d2h(h, c, h->pre_ctr, sizeof(c)), "qrw_mpc_get_slice_stats");
  HIP_OK(host_done(h), "qrw_mpc_get_slice_stats");
  for (int l = 0; l < qrw::kPreMaxLevels; l++) h_parks_per_level[l] = c[qrw::kPreLevelWord + 2 * l + 1];
  *h_takers = c[qrw::kPreTicketWord];
  *h_finished = c[qrw::kPreDoneWord];
  return 0;
}

extern "C" int qrw_test_poke_aborted(qrw_handle h, int32_t parked_at) {
  if (!h || parked_at < 1) return fail(-1, "qrw_test_poke_aborted: bad argument");
  if (!h->pause_it || !mpc_time_sliced(h)) return fail(-1, "qrw_test_poke_aborted: not a time-sliced handle");
  DeviceScope dev_scope__(h->cfg.device);
  HIP_OK(wait_family(h, kFamMpc), "qrw_test_poke_aborted sync");
  const size_t B = h->cfg.batch, T = qrw::mpc_threads(h->cfg.n_steps);
  std::vector<int> pit(B, parked_at);
  HIP_OK(hipMemcpy(h->pause_it, pit.data(), B * sizeof(int), hipMemcpyHostToDevice), "qrw_test_poke_aborted pause_it");
  // the iterate slots (items kStXX .. kStYC + 4) and rho: anything but what the last finished solve left
  std::vector<double> st((size_t)qrw::kMpcStItems * T);
  for (size_t b = 0; b < B; b++) {
    double* d = h->mpc_st + b * qrw::kMpcStItems * T;
    HIP_OK(hipMemcpy(st.data(), d, st.size() * sizeof(double), hipMemcpyDeviceToHost), "qrw_test_poke_aborted D2H");
    for (size_t e = 0; e < (size_t)qrw::kStB * T; e++) st[e] = 3.7 * st[e] + 1.0 + 0.01 * (double)(e % 17);
    for (size_t e = 0; e < T; e++) st[(size_t)qrw::kStRho * T + e] = 0.37;
    HIP_OK(hipMemcpy(d, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice), "qrw_test_poke_aborted H2D");
  }
  return 0;
}

// Tests only: is the poisoning itself effective on this device?  After both fills a probe wavefront reads, without writing them first,
// one LDS word, v255 and a255: out[0..2] (all-ones if the leftovers are what a new wavefront sees; the hardware clears neither).
__global__ __launch_bounds__(64, 1) void poison_probe_kernel(unsigned* out) {
  __shared__ unsigned s[40000];  // (the whole LDS of the compute unit: the word read below is inside the allocation)
  unsigned v, acc, w;
  const unsigned addr = (unsigned)(size_t)(&s[39000]) + 4u * threadIdx.x;
  // (through asm: the compiler folds a plain read of never-written shared memory to a constant)
  asm volatile("v_mov_b32 %0, v255\n\tv_accvgpr_read_b32 %1, a255\n\tds_read_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=v"(v), "=v"(acc), "=v"(w) : "v"(addr) : "v255", "a255", "memory");
  if (threadIdx.x == 5) { out[0] = w; out[1] = v; out[2] = acc; }
}
extern "C" int qrw_test_poison_probe(uint32_t* h_out3) {
  if (!h_out3) return fail(-1, "qrw_test_poison_probe: null argument");
  std::lock_guard<std::mutex> lock(g_kat_mutex);
  unsigned* d = nullptr;
  HIP_OK(hipMalloc((void**)&d, 3 * sizeof(unsigned)), "qrw_test_poison_probe alloc");
  hipMemset(d, 0, 3 * sizeof(unsigned));
  hipLaunchKernelGGL(lds_poison_kernel, dim3(4096), dim3(256), 0, nullptr, ~0ull, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(reg_poison_kernel, dim3(8192), dim3(64), 0, nullptr, (int*)nullptr);
  hipLaunchKernelGGL(poison_probe_kernel, dim3(1), dim3(64), 0, nullptr, d);
  const hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h_out3, d, 3 * sizeof(unsigned), hipMemcpyDeviceToHost);
  hipFree(d);
  HIP_OK(e, "qrw_test_poison_probe");
  return 0;
}

extern "C" int qrw_test_known_answer(int32_t N, int32_t mode, uint64_t lds_pattern, int32_t poison, int32_t* iters, int32_t* status,
                                     double* rho, double* err) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-2, "qrw_test_known_answer: no HIP device");
  std::lock_guard<std::mutex> lock(g_kat_mutex);
  if (poison) {
    hipLaunchKernelGGL(lds_poison_kernel, dim3(4096), dim3(256), 0, nullptr, (unsigned long long)lds_pattern, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(reg_poison_kernel, dim3(8192), dim3(64), 0, nullptr, (int*)nullptr);
    HIP_OK(hipGetLastError(), "qrw_test_known_answer poison launch");
    HIP_OK(hipDeviceSynchronize(), "qrw_test_known_answer poison");
  }
  KatResult r;
  const int rc = mpc_known_answer_check(N, mode, &r);
  if (iters) *iters = r.iters;
  if (status) *status = r.status;
  if (rho) *rho = r.rho;
  if (err) *err = r.err;
  return rc;
}

extern "C" int qrw_mpc_get_order(qrw_handle h, int32_t* h_order, float* h_ema, int32_t* has_order) {
  if (!h || !has_order) return fail(-1, "qrw_mpc_get_order: null argument");
  DeviceScope dev_scope__(h->cfg.device);
  HostStreamGuard host_guard__(h);
  HIP_OK(wait_family(h, kFamMpc), "qrw_mpc_get_order sync");
  *has_order = h->mpc_have_order ? 1 : 0;
  const size_t B = h->cfg.batch;
  if (h_order) HIP_OK(d2h(h, h_order, h->mpc_order, B * sizeof(int32_t)), "qrw_mpc_get_order");
  if (h_ema) HIP_OK(d2h(h, h_ema, h->mpc_ema, B * sizeof(float)), "qrw_mpc_get_order ema");
  HIP_OK(host_done(h), "qrw_mpc_get_order");
  return 0;
}

extern "C" int qrw_mpc_get_state(qrw_handle h, int32_t b, double* h_x, double* h_z, double* h_y, double* h_D,
                                 double* h_E, double* h_c) {
  if (!h || b < 0 || b >= h->cfg.batch) return fail(-1, "qrw_mpc_get_state: bad argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const int N = h->cfg.n_steps;
  const int T = qrw::mpc_threads(N);
  std::vector<double> s((size_t)qrw::kMpcStItems * T);
  HIP_OK(wait_family(h, kFamMpc), "sync");
  HIP_OK(d2h(h, s.data(), h->mpc_st + (size_t)b * qrw::kMpcStItems * T, s.size() * sizeof(double)), "D2H state");
  HIP_OK(host_done(h), "D2H state");
  auto at = [&](int item, int k, int j) { return s[(size_t)item * T + 4 * k + j]; };
  for (int k = 0; k < N; k++)
    for (int j = 0; j < 4; j++) {
      for (int t = 0; t < 3; t++) {
        const int i = 12 * k + 3 * j + t;
        if (h_x) { h_x[i] = at(qrw::kStXX + t, k, j); h_x[12 * N + i] = at(qrw::kStXF + t, k, j); }
        if (h_D) { h_D[i] = at(qrw::kStDX + t, k, j); h_D[12 * N + i] = at(qrw::kStDF + t, k, j); }
        if (h_z) { h_z[i] = at(qrw::kStZD + t, k, j); h_z[12 * N + i] = 0.0; }
        if (h_y) { h_y[i] = at(qrw::kStYD + t, k, j); h_y[12 * N + i] = at(qrw::kStYS + t, k, j); }
        if (h_E) { h_E[i] = at(qrw::kStED + t, k, j); h_E[12 * N + i] = at(qrw::kStES + t, k, j); }
      }
      for (int c = 0; c < 5; c++) {
        const int r = 24 * N + 20 * k + 5 * j + c;
        if (h_z) h_z[r] = at(qrw::kStZC + c, k, j);
        if (h_y) h_y[r] = at(qrw::kStYC + c, k, j);
        if (h_E) h_E[r] = at(qrw::kStEC + c, k, j);
      }
    }
  if (h_c) *h_c = s[(size_t)qrw::kStC * T];
  return 0;
}

static void wbc_common(qrw_handle h, qrw::WbcArgs& a) {
  memset(&a, 0, sizeof(a));
  a.B = h->cfg.batch;
  a.dt = h->cfg.dt_wbc;
  a.st = h->wbc_st;
  a.iters = h->wbc_iters;
  a.status = h->wbc_status;
  for (int i = 0; i < 6; i++) a.Y[i] = h->Y[i];
  a.lanes16 = h->wbc_lanes16;
}

extern "C" int qrw_wbc_set_lanes(qrw_handle h, int32_t lanes) {
  if (!h || (lanes != 4 && lanes != 16)) return fail(-1, "qrw_wbc_set_lanes: lanes must be 4 or 16");
  h->wbc_lanes16 = (lanes == 16);
  return 0;
}

extern "C" int qrw_wbc_compute(qrw_handle h, const double* d_q, const double* d_dq, const double* d_f_cmd,
                               const double* d_contacts, const double* d_pgoals, const double* d_vgoals,
                               const double* d_agoals, double* d_tau_ff, double* d_qdes, double* d_vdes,
                               double* d_f_with_delta, double* d_ddq_res, double* d_feet, void* stream) {
  if (!h || !d_q || !d_dq || !d_f_cmd || !d_contacts || !d_pgoals || !d_vgoals || !d_agoals)
    return fail(-1, "qrw_wbc_compute: null input");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  qrw::WbcArgs a;
  wbc_common(h, a);
  a.mode = 0;
  a.q = d_q; a.dq = d_dq; a.f_cmd = d_f_cmd; a.contacts = d_contacts;
  a.pgoals = d_pgoals; a.vgoals = d_vgoals; a.agoals = d_agoals;
  a.tau_ff = d_tau_ff; a.qdes = d_qdes; a.vdes = d_vdes; a.f_with_delta = d_f_with_delta;
  a.ddq_res = d_ddq_res; a.feet = d_feet;
  debug_poison_lds((hipStream_t)stream);
  if (qrw::wbc_launch(a, (hipStream_t)stream) != 0) return fail(-11, "qrw_wbc_compute: kernel launch failed", hipGetLastError());
  note_launch(h, kFamWbc, (hipStream_t)stream);
  return 0;
}

extern "C" int qrw_wbc_compute_result(qrw_handle h, const double* d_q, const double* d_dq, const double* d_f_cmd,
                                      const double* d_contacts, const double* d_pgoals, const double* d_vgoals,
                                      const double* d_agoals, double* d_tau_ff, double* d_qdes, double* d_vdes,
                                      double* d_f_with_delta, double* d_ddq_res, double* d_feet, const double* d_q_filt,
                                      const double* d_v_secu, double* d_result, int32_t* d_error_flag, void* stream) {
  if (!h || !d_q || !d_dq || !d_f_cmd || !d_contacts || !d_pgoals || !d_vgoals || !d_agoals || !d_q_filt || !d_v_secu ||
      !d_result)
    return fail(-1, "qrw_wbc_compute_result: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  qrw::WbcArgs a;
  wbc_common(h, a);
  a.mode = 0;
  a.q = d_q; a.dq = d_dq; a.f_cmd = d_f_cmd; a.contacts = d_contacts;
  a.pgoals = d_pgoals; a.vgoals = d_vgoals; a.agoals = d_agoals;
  a.tau_ff = d_tau_ff; a.qdes = d_qdes; a.vdes = d_vdes; a.f_with_delta = d_f_with_delta;
  a.ddq_res = d_ddq_res; a.feet = d_feet;
  a.c_cs = h->ctrl_st; a.c_qfilt = d_q_filt; a.c_vsecu = d_v_secu; a.c_result = d_result; a.c_err = d_error_flag;
  debug_poison_lds((hipStream_t)stream);
  if (qrw::wbc_launch(a, (hipStream_t)stream) != 0)
    return fail(-11, "qrw_wbc_compute_result: kernel launch failed", hipGetLastError());
  note_launch(h, kFamWbc, (hipStream_t)stream);
  return 0;
}

namespace {
struct Stager {
  qrw_handle h;
  size_t off = 0;
  bool ok = true;
  explicit Stager(qrw_handle hh) : h(hh) {}
  double* in(const double* src, size_t n) {
    double* d = out(n);
    if (d && src && h2d(h, d, src, n * sizeof(double)) != hipSuccess) ok = false;
    return d;
  }
  double* out(size_t n) {
    if (off + n > h->stage_doubles) { ok = false; return nullptr; }
    double* d = h->stage + off;
    off += n;
    return d;
  }
  // (asynchronous on the handle's host stream: the caller ends with host_done)
  bool back(double* dst, const double* d, size_t n) {
    if (!dst) return true;
    return d2h(h, dst, d, n * sizeof(double)) == hipSuccess;
  }
};
}  // namespace

extern "C" int qrw_wbc_compute_host(qrw_handle h, const double* h_q, const double* h_dq, const double* h_f_cmd,
                                    const double* h_contacts, const double* h_pgoals, const double* h_vgoals,
                                    const double* h_agoals, double* h_tau_ff, double* h_qdes, double* h_vdes,
                                    double* h_f_with_delta, double* h_ddq_res, double* h_feet) {
  if (!h) return fail(-1, "qrw_wbc_compute_host: null handle");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch;
  HIP_OK(wait_family(h, kFamWbc), "qrw_wbc_compute_host: earlier WBC step of this handle");
  Stager s(h);
  double *q = s.in(h_q, B * 19), *dq = s.in(h_dq, B * 18), *f = s.in(h_f_cmd, B * 12), *c = s.in(h_contacts, B * 4);
  double *pg = s.in(h_pgoals, B * 12), *vg = s.in(h_vgoals, B * 12), *ag = s.in(h_agoals, B * 12);
  double *tau = s.out(B * 12), *qd = s.out(B * 19), *vd = s.out(B * 18), *fw = s.out(B * 12), *dd = s.out(B * 6),
         *ft = s.out(B * 36);
  if (!s.ok) return fail(-12, "qrw_wbc_compute_host: staging failed");
  int rc = qrw_wbc_compute(h, q, dq, f, c, pg, vg, ag, tau, qd, vd, fw, dd, ft, (void*)h->host_stream);
  if (rc) return rc;
  if (!(s.back(h_tau_ff, tau, B * 12) && s.back(h_qdes, qd, B * 19) && s.back(h_vdes, vd, B * 18) &&
        s.back(h_f_with_delta, fw, B * 12) && s.back(h_ddq_res, dd, B * 6) && s.back(h_feet, ft, B * 36)))
    return fail(-12, "qrw_wbc_compute_host: D2H failed");
  HIP_OK(host_done(h), "wbc sync");
  return 0;
}

extern "C" int qrw_wbc_get_stats(qrw_handle h, int32_t* h_iters, int32_t* h_status, double* h_rho,
                                 double* h_k_since_contact) {
  if (!h) return fail(-1, "qrw_wbc_get_stats: null handle");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch;
  HIP_OK(wait_family(h, kFamWbc), "sync");
  if (h_iters) HIP_OK(d2h(h, h_iters, h->wbc_iters, B * sizeof(int)), "D2H iters");
  if (h_status) HIP_OK(d2h(h, h_status, h->wbc_status, B * sizeof(int)), "D2H status");
  HIP_OK(host_done(h), "qrw_wbc_get_stats");
  if (h_rho || h_k_since_contact) {
    std::vector<double> s(B * qrw::kWbcStItems);
    HIP_OK(d2h(h, s.data(), h->wbc_st, s.size() * sizeof(double)), "D2H wbc state");
    HIP_OK(host_done(h), "D2H wbc state");
    for (size_t b = 0; b < B; b++) {
      if (h_rho) h_rho[b] = s[b * qrw::kWbcStItems + qrw::kWsRho];
      if (h_k_since_contact)
        for (int i = 0; i < 4; i++) h_k_since_contact[b * 4 + i] = s[b * qrw::kWbcStItems + qrw::kWsKsc + i];
    }
  }
  return 0;
}

extern "C" int qrw_fixed_feet_host(qrw_handle h, const double* h_q12, const double* h_dq12, double* h_posf, double* h_vf,
                                   double* h_wf, double* h_af, double* h_Jf) {
  if (!h || !h_q12 || !h_dq12) return fail(-1, "qrw_fixed_feet_host: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch;
  Stager s(h);
  qrw::WbcArgs a;
  wbc_common(h, a);
  a.mode = 1;
  a.in0 = s.in(h_q12, B * 12); a.in1 = s.in(h_dq12, B * 12);
  a.out0 = s.out(B * 12); a.out1 = s.out(B * 12); a.out2 = s.out(B * 12); a.out3 = s.out(B * 12); a.out4 = s.out(B * 144);
  if (!s.ok) return fail(-12, "qrw_fixed_feet_host: staging failed");
  if (qrw::wbc_launch(a, h->host_stream) != 0) return fail(-11, "qrw_fixed_feet_host: launch failed", hipGetLastError());
  if (!(s.back(h_posf, a.out0, B * 12) && s.back(h_vf, a.out1, B * 12) && s.back(h_wf, a.out2, B * 12) &&
        s.back(h_af, a.out3, B * 12) && s.back(h_Jf, a.out4, B * 144)))
    return fail(-12, "qrw_fixed_feet_host: D2H failed");
  HIP_OK(host_done(h), "sync");
  return 0;
}

extern "C" int qrw_invkin_host(qrw_handle h, const double* h_contacts, const double* h_goals, const double* h_vgoals,
                               const double* h_agoals, const double* h_posf, const double* h_vf, const double* h_wf,
                               const double* h_af, const double* h_Jf, double* h_ddq, double* h_dq_cmd,
                               double* h_q_step) {
  if (!h) return fail(-1, "qrw_invkin_host: null handle");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch;
  Stager s(h);
  qrw::WbcArgs a;
  wbc_common(h, a);
  a.mode = 2;
  a.in0 = s.in(h_contacts, B * 4); a.in1 = s.in(h_goals, B * 12); a.in2 = s.in(h_vgoals, B * 12);
  a.in3 = s.in(h_agoals, B * 12); a.in4 = s.in(h_posf, B * 12); a.in5 = s.in(h_vf, B * 12);
  a.in6 = s.in(h_wf, B * 12); a.in7 = s.in(h_af, B * 12); a.in8 = s.in(h_Jf, B * 144);
  a.out0 = s.out(B * 12); a.out1 = s.out(B * 12); a.out2 = s.out(B * 12);
  if (!s.ok) return fail(-12, "qrw_invkin_host: staging failed");
  if (qrw::wbc_launch(a, h->host_stream) != 0) return fail(-11, "qrw_invkin_host: launch failed", hipGetLastError());
  if (!(s.back(h_ddq, a.out0, B * 12) && s.back(h_dq_cmd, a.out1, B * 12) && s.back(h_q_step, a.out2, B * 12)))
    return fail(-12, "qrw_invkin_host: D2H failed");
  HIP_OK(host_done(h), "sync");
  return 0;
}

extern "C" int qrw_qpwbc_host(qrw_handle h, const double* h_M, const double* h_Jc, const double* h_f_cmd,
                              const double* h_RNEA, double* h_f_res, double* h_ddq_res, double* h_H) {
  if (!h || !h_M || !h_Jc || !h_f_cmd || !h_RNEA) return fail(-1, "qrw_qpwbc_host: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch;
  HIP_OK(wait_family(h, kFamWbc), "qrw_qpwbc_host: earlier WBC step of this handle");
  Stager s(h);
  qrw::WbcArgs a;
  wbc_common(h, a);
  a.mode = 3;
  a.in0 = s.in(h_M, B * 324); a.in1 = s.in(h_Jc, B * 216); a.in2 = s.in(h_f_cmd, B * 12); a.in3 = s.in(h_RNEA, B * 6);
  a.out0 = s.out(B * 12); a.out1 = s.out(B * 6); a.out2 = h_H ? s.out(B * 144) : nullptr;
  // M[:6,:6] diagonal in every instance (what scripts/QP_WBC.py:93 hands over): the kernel inverts it in place; otherwise the
  // general pseudoInverse<> of include/qrw/InvKin.hpp:60-66 is computed first (pinv6_kernel)
  bool diagonal = true;
  for (size_t b = 0; b < B && diagonal; b++)
    for (int i = 0; i < 6 && diagonal; i++)
      for (int c = 0; c < 6; c++)
        if (i != c && h_M[b * 324 + i * 18 + c] != 0.0) { diagonal = false; break; }
  double* d_yinv = nullptr;
  if (!diagonal) d_yinv = s.out(B * 36);
  if (!s.ok) return fail(-12, "qrw_qpwbc_host: staging failed");
  if (d_yinv) {
    if (qrw::pinv6_launch(a.in0, d_yinv, (int)B, h->host_stream) != 0) return fail(-11, "qrw_qpwbc_host: launch failed", hipGetLastError());
    a.in4 = d_yinv;
  }
  if (qrw::wbc_launch(a, h->host_stream) != 0) return fail(-11, "qrw_qpwbc_host: launch failed", hipGetLastError());
  note_launch(h, kFamWbc, h->host_stream);
  if (!(s.back(h_f_res, a.out0, B * 12) && s.back(h_ddq_res, a.out1, B * 6) && s.back(h_H, a.out2, B * 144)))
    return fail(-12, "qrw_qpwbc_host: D2H failed");
  HIP_OK(host_done(h), "sync");
  return 0;
}

extern "C" int qrw_get_base_inertia_diag(qrw_handle h, double* h_Y6) {
  if (!h || !h_Y6) return fail(-1, "qrw_get_base_inertia_diag: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  for (int i = 0; i < 6; i++) h_Y6[i] = h->Y[i];
  return 0;
}

extern "C" int qrw_selftest_sweeps(double* max_err) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-2, "qrw_selftest_sweeps: no HIP device");
  std::lock_guard<std::mutex> lock(g_kat_mutex);  // one self-test at a time (qrw_create's known-answer check takes the same lock)
  const int rc = qrw::sweeps_selftest(max_err);
  if (rc != 0) return rc;
  // the whole solve, not only its sweeps: every instantiation of mpc_solve_kernel (compile-time and runtime horizon, one and two
  // wavefronts, one launch per call / time-sliced / sequence)
  static const int kat_cases[][2] = {{16, kKatPlain}, {12, kKatPlain}, {32, kKatPlain}, {24, kKatPlain}, {32, kKatSliced}, {24, kKatSliced},
                                     {16, kKatSequence}, {12, kKatSequence}, {32, kKatSequence}, {24, kKatSequence}};
  for (const auto& kc : kat_cases) {
    KatResult kat;
    const int krc = mpc_known_answer_check(kc[0], kc[1], &kat);
    if (krc != 0) {
      char msg[320];
      snprintf(msg, sizeof(msg), "qrw_selftest_sweeps: known-answer MPC solve failed for N = %d, %s (rc %d: %d iterations, status %d, rho %.10g, "
               "error %.3g; expected %d iterations, rho %.10g)", kc[0], kKatModeName[kc[1]], krc, kat.iters, kat.status, kat.rho, kat.err,
               kat.want_iters, kat.want_rho);
      return fail(2, msg);
    }
  }
  return 0;
}

// Diagnostic (profiling builds only, -DQRW_PROFILE_PHASES): per-instance shader-clock totals of the MPC kernel phases.
extern "C" int qrw_mpc_get_phase_cycles(qrw_handle h, double* h_prof /* [B][10] */) {
  if (!h || !h_prof) return fail(-1, "qrw_mpc_get_phase_cycles: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  HIP_OK(wait_family(h, kFamMpc), "sync");
  HIP_OK(d2h(h, h_prof, h->mpc_prof, (size_t)h->cfg.batch * qrw::kMpcProfItems * sizeof(double)), "D2H prof");
  HIP_OK(host_done(h), "D2H prof");
  return 0;
}

// ------------------------------------------------------------------ planners
static void planner_common(qrw_handle h, qrw::PlannerArgs& a) {
  memset(&a, 0, sizeof(a));
  a.B = h->cfg.batch; a.n_steps = h->cfg.n_steps; a.N_gait = h->cfg.N_gait; a.k_mpc = h->pcfg.k_mpc;
  a.dt_mpc = h->cfg.dt_mpc; a.dt_wbc = h->cfg.dt_wbc; a.T_gait = h->cfg.T_gait; a.T_mpc = h->cfg.dt_mpc * h->cfg.n_steps;
  a.h_ref = h->pcfg.h_ref; a.k_feedback = 0.03; a.g = 9.81; a.L = 0.155;  // FootstepPlanner.cpp:5-7
  a.max_height = h->pcfg.max_height; a.lock_time = h->pcfg.lock_time;
  for (int i = 0; i < 12; i++) { a.shoulders[i] = h->pcfg.shoulders[i]; a.init_target[i] = h->pcfg.init_target[i]; a.init_pos[i] = h->pcfg.init_foot_pos[i]; }
  a.ps = h->plan_st;
  a.q_ld = 7;
}

extern "C" int qrw_planner_init(qrw_handle h, const qrw_planner_config* pc, void* stream) {
  if (!h || !pc) return fail(-1, "qrw_planner_init: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  if (pc->k_mpc < 1) return fail(-1, "qrw_planner_init: k_mpc must be >= 1");
  if (h->cfg.N_gait > 63) return fail(-1, "qrw_planner_init: N_gait must be <= 63 (gait matrices are 64-bit column masks)");
  // Gait::initialize throws when the matrices are too small (src/Gait.cpp:30-31)
  const long per = lround(h->cfg.T_gait / h->cfg.dt_mpc);
  if (h->cfg.n_steps > h->cfg.N_gait || per > h->cfg.N_gait || h->cfg.n_steps + 1 > h->cfg.N_gait)
    return fail(-3, "Sizes of matrices are too small for considered durations. Increase N_gait in config file.");
  h->pcfg = *pc;
  qrw::PlannerArgs a;
  planner_common(h, a);
  a.mode = qrw::kPlanInit;
  if (qrw::planner_launch(a, (hipStream_t)stream) != 0) return fail(-11, "qrw_planner_init: launch failed", hipGetLastError());
  note_launch(h, kFamPlan, (hipStream_t)stream);
  h->plan_ready = true;
  return 0;
}

extern "C" int qrw_planner_step(qrw_handle h, int32_t k, const double* d_q7, int32_t q_ld, const double* d_hv,
                                const double* d_vref, const int32_t* d_code, int32_t code_scalar, double* d_xref,
                                double* d_fsteps, double* d_gait, double* d_target, double* d_feet_pva,
                                double* d_contacts, void* stream) {
  if (!h || !h->plan_ready) return fail(-1, "qrw_planner_step: planner not initialised");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  if (!d_q7 || !d_hv || !d_vref) return fail(-1, "qrw_planner_step: null input");
  if (q_ld < 7) return fail(-1, "qrw_planner_step: q_ld must be at least 7");
  qrw::PlannerArgs a;
  planner_common(h, a);
  a.mode = qrw::kPlanGait | qrw::kPlanFootsteps | qrw::kPlanTraj | qrw::kPlanState;
  a.k = k;
  a.refresh = ((k % a.k_mpc) == 0 && k != 0) ? 1 : 0;     // scripts/Controller.py:225
  a.k_footsteps = a.k_mpc - k % a.k_mpc;                  // scripts/Controller.py:226
  a.q7 = d_q7; a.q_ld = q_ld; a.hv = d_hv; a.vref = d_vref; a.code = d_code; a.code_scalar = code_scalar;
  a.xref = d_xref; a.fsteps = d_fsteps; a.gait = d_gait; a.target = d_target; a.feet_pva = d_feet_pva;
  a.contacts = d_contacts;
  if (qrw::planner_launch(a, (hipStream_t)stream) != 0) return fail(-11, "qrw_planner_step: launch failed", hipGetLastError());
  note_launch(h, kFamPlan, (hipStream_t)stream);
  return 0;
}

extern "C" int qrw_planner_call_host(qrw_handle h, int32_t mode, int32_t k, int32_t k_footsteps, int32_t refresh,
                                     const double* h_q7, const double* h_v6, const double* h_vref6, int32_t code,
                                     const double* h_target_in, double z_average, double* h_xref, double* h_fsteps,
                                     double* h_gait, double* h_target, double* h_feet_pva) {
  if (!h || !h->plan_ready) return fail(-1, "qrw_planner_call_host: planner not initialised");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const size_t B = h->cfg.batch, N = h->cfg.n_steps, Ng = h->cfg.N_gait;
  HIP_OK(wait_family(h, kFamPlan), "qrw_planner_call_host: earlier planner step of this handle");
  Stager s(h);
  qrw::PlannerArgs a;
  planner_common(h, a);
  a.mode = mode & ~qrw::kPlanInit;
  a.k = k; a.k_footsteps = k_footsteps; a.refresh = refresh; a.code_scalar = code; a.z_average = z_average;
  a.q7 = h_q7 ? s.in(h_q7, B * 7) : nullptr;
  a.hv = h_v6 ? s.in(h_v6, B * 6) : nullptr;
  a.vref = h_vref6 ? s.in(h_vref6, B * 6) : nullptr;
  a.target_in = h_target_in ? s.in(h_target_in, B * 12) : nullptr;
  a.xref = h_xref ? s.out(B * 12 * (N + 1)) : nullptr;
  a.fsteps = h_fsteps ? s.out(B * Ng * 12) : nullptr;
  a.gait = h_gait ? s.out(B * Ng * 4) : nullptr;
  a.target = h_target ? s.out(B * 12) : nullptr;
  a.feet_pva = h_feet_pva ? s.out(B * 36) : nullptr;
  if (!s.ok) return fail(-12, "qrw_planner_call_host: staging failed");
  if (qrw::planner_launch(a, h->host_stream) != 0) return fail(-11, "qrw_planner_call_host: launch failed", hipGetLastError());
  note_launch(h, kFamPlan, h->host_stream);
  if (!(s.back(h_xref, a.xref, B * 12 * (N + 1)) && s.back(h_fsteps, a.fsteps, B * Ng * 12) && s.back(h_gait, a.gait, B * Ng * 4) &&
        s.back(h_target, a.target, B * 12) && s.back(h_feet_pva, a.feet_pva, B * 36)))
    return fail(-12, "qrw_planner_call_host: D2H failed");
  HIP_OK(host_done(h), "planner sync");
  return 0;
}

extern "C" int qrw_planner_get_host(qrw_handle h, int32_t which, int32_t b, int32_t count, double* h_out) {
  if (!h || !h_out || b < 0 || b >= h->cfg.batch || count < 1) return fail(-1, "qrw_planner_get_host: bad argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  HostStreamGuard host_guard__(h);
  const int off = qrw::planner_item_offset(h->cfg.N_gait, which);
  const size_t B = h->cfg.batch;
  HIP_OK(wait_family(h, kFamPlan), "sync");
  if (which >= 0 && which <= 2) {  // a gait matrix: stored as four 64-bit column masks, returned as N_gait x 4 doubles
    if (count > h->cfg.N_gait * 4) return fail(-1, "qrw_planner_get_host: bad item");
    double raw[4];
    HIP_OK(hipMemcpy2DAsync(raw, sizeof(double), h->plan_st + (size_t)off * B + b, B * sizeof(double), sizeof(double), 4,
                            hipMemcpyDeviceToHost, h->host_stream), "D2H planner state");
    HIP_OK(host_done(h), "D2H planner state");
    for (int e = 0; e < count; e++) {
      unsigned long long mask;
      memcpy(&mask, &raw[e % 4], sizeof(mask));
      h_out[e] = ((mask >> (e / 4)) & 1ull) ? 1.0 : 0.0;
    }
    return 0;
  }
  if (off < 0 || off + count > qrw::planner_state_items(h->cfg.N_gait)) return fail(-1, "qrw_planner_get_host: bad item");
  HIP_OK(hipMemcpy2DAsync(h_out, sizeof(double), h->plan_st + (size_t)off * B + b, B * sizeof(double), sizeof(double), count,
                          hipMemcpyDeviceToHost, h->host_stream), "D2H planner state");
  HIP_OK(host_done(h), "D2H planner state");
  return 0;
}

// ------------------------------------------------------------------ controller glue
static void ctrl_common(qrw_handle h, qrw::ControllerArgs& a, int mode) {
  memset(&a, 0, sizeof(a));
  a.B = h->cfg.batch; a.n_steps = h->cfg.n_steps; a.mode = mode; a.dt_wbc = h->cfg.dt_wbc; a.h_ref = h->pcfg.h_ref;
  a.n_gait = h->cfg.N_gait;
  a.cs = h->ctrl_st;
}
extern "C" int qrw_mpc_result_shift(qrw_handle h, const double* d_gait, double* d_x_f_mpc, void* stream) {
  if (!h || !d_gait || !d_x_f_mpc) return fail(-1, "qrw_mpc_result_shift: null argument");
  DeviceScope dev_scope__(h->cfg.device);
  qrw::ControllerArgs a;
  ctrl_common(h, a, qrw::kCtrlMpcShift);
  a.in0 = d_gait;
  a.out0 = d_x_f_mpc;
  return qrw::controller_launch(a, (hipStream_t)stream) ? fail(-11, "qrw_mpc_result_shift: launch failed", hipGetLastError()) : 0;
}
extern "C" int qrw_controller_init(qrw_handle h, const double* d_q_init12, double h_ref, void* stream) {
  if (!h) return fail(-1, "qrw_controller_init: null handle");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  h->pcfg.h_ref = h_ref;
  qrw::ControllerArgs a;
  ctrl_common(h, a, qrw::kCtrlInit);
  a.in0 = d_q_init12;
  return qrw::controller_launch(a, (hipStream_t)stream) ? fail(-11, "qrw_controller_init: launch failed", hipGetLastError()) : 0;
}
extern "C" int qrw_controller_update_state(qrw_handle h, const double* d_joy_vref, const double* d_q_filt,
                                           const double* d_v_filt, const double* d_rpy, double* d_q, double* d_v,
                                           double* d_hv, double* d_vref, double* d_oRh_oTh, void* stream) {
  if (!h || !d_joy_vref || !d_q_filt || !d_v_filt || !d_rpy || !d_q || !d_v || !d_hv)
    return fail(-1, "qrw_controller_update_state: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  qrw::ControllerArgs a;
  ctrl_common(h, a, qrw::kCtrlUpdateState);
  a.in0 = d_joy_vref; a.in1 = d_q_filt; a.in2 = d_v_filt; a.in3 = d_rpy;
  a.out0 = d_q; a.out1 = d_v; a.out2 = d_hv; a.out3 = d_vref; a.out4 = d_oRh_oTh;
  return qrw::controller_launch(a, (hipStream_t)stream) ? fail(-11, "qrw_controller_update_state: launch failed", hipGetLastError()) : 0;
}
extern "C" int qrw_controller_wbc_inputs(qrw_handle h, const double* d_x_f_mpc, const double* d_xref,
                                         const double* d_feet_pva, const double* d_v, double* d_x_f_wbc, double* d_q_wbc,
                                         double* d_b_v, double* d_f_cmd, double* d_feet_cmd, void* stream) {
  if (!h || !d_x_f_mpc || !d_xref || !d_feet_pva || !d_v || !d_q_wbc || !d_b_v || !d_feet_cmd)
    return fail(-1, "qrw_controller_wbc_inputs: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  qrw::ControllerArgs a;
  ctrl_common(h, a, qrw::kCtrlWbcInputs);
  a.in0 = d_x_f_mpc; a.in1 = d_xref; a.in2 = d_feet_pva; a.in3 = d_v;
  a.out0 = d_x_f_wbc; a.out1 = d_q_wbc; a.out2 = d_b_v; a.out3 = d_f_cmd; a.out4 = d_feet_cmd;
  return qrw::controller_launch(a, (hipStream_t)stream) ? fail(-11, "qrw_controller_wbc_inputs: launch failed", hipGetLastError()) : 0;
}
extern "C" int qrw_controller_result(qrw_handle h, const double* d_tau_ff, const double* d_qdes, const double* d_vdes,
                                     const double* d_q_filt, const double* d_v_secu, double* d_result,
                                     int32_t* d_error_flag, void* stream) {
  if (!h || !d_tau_ff || !d_qdes || !d_vdes || !d_q_filt || !d_v_secu || !d_result)
    return fail(-1, "qrw_controller_result: null argument");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  qrw::ControllerArgs a;
  ctrl_common(h, a, qrw::kCtrlResult);
  a.in0 = d_tau_ff; a.in1 = d_qdes; a.in2 = d_vdes; a.in3 = d_q_filt; a.in4 = d_v_secu;
  a.out0 = d_result; a.iout = d_error_flag;
  return qrw::controller_launch(a, (hipStream_t)stream) ? fail(-11, "qrw_controller_result: launch failed", hipGetLastError()) : 0;
}

// ------------------------------------------------------------------ streams on a subset of the compute units
extern "C" int qrw_device_cu_count(int32_t device, int32_t* n_cus) {
  if (!n_cus) return fail(-1, "qrw_device_cu_count: null argument");
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(-10, "qrw_device_cu_count: hipGetDeviceProperties failed", e);
  *n_cus = prop.multiProcessorCount;
  return 0;
}
extern "C" int qrw_stream_create(int32_t device, int32_t first_cu, int32_t n_cus, void** stream) {
  if (!stream) return fail(-1, "qrw_stream_create: null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(-2, "qrw_stream_create: bad device ordinal");
  DeviceScope dev_scope__(device);
  hipError_t e = hipSuccess;
  hipStream_t s = nullptr;
  if (n_cus <= 0) {
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  } else {
    int32_t total = 0;
    if (qrw_device_cu_count(device, &total)) return -10;
    if (first_cu < 0 || first_cu + n_cus > total) return fail(-1, "qrw_stream_create: compute-unit range outside the device");
    const int words = (total + 31) / 32;
    uint32_t mask[64] = {0};
    if (words > 64) return fail(-1, "qrw_stream_create: device has more than 2048 compute units");
    for (int c = first_cu; c < first_cu + n_cus; c++) mask[c >> 5] |= 1u << (c & 31);
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  }
  if (e != hipSuccess) return fail(-10, "qrw_stream_create: stream creation failed", e);
  *stream = (void*)s;
  return 0;
}
extern "C" int qrw_stream_destroy(void* stream) {
  if (!stream) return 0;
  // whatever the handles launched on it is complete before the stream goes: their getters then have nothing to wait for
  (void)hipStreamSynchronize((hipStream_t)stream);
  {
    std::lock_guard<std::mutex> lock(g_handles_mutex);
    for (qrw_handle_s* h : g_handles)
      for (int f = 0; f < kFamCount; f++)
        if (h->launched[f].load(std::memory_order_acquire) && h->last_stream[f].load(std::memory_order_relaxed) == (hipStream_t)stream)
          h->launched[f].store(false, std::memory_order_release);
  }
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? 0 : fail(-10, "qrw_stream_destroy: hipStreamDestroy failed", e);
}

extern "C" int qrw_stream_wait_stream(qrw_handle h, void* waiter, void* signaller) {
  if (!h) return fail(-1, "qrw_stream_wait_stream: null handle");
  if (waiter == signaller) return 0;
  DeviceScope dev_scope__(h->cfg.device);
  if (!h->order_ev) HIP_OK(hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming), "qrw_stream_wait_stream: hipEventCreate");
  // one event serves every hand-over: hipStreamWaitEvent captures the record that is current when it is called
  HIP_OK(hipEventRecord(h->order_ev, (hipStream_t)signaller), "qrw_stream_wait_stream: hipEventRecord");
  HIP_OK(hipStreamWaitEvent((hipStream_t)waiter, h->order_ev, 0), "qrw_stream_wait_stream: hipStreamWaitEvent");
  return 0;
}

// ------------------------------------------------------------------ fused head of a control iteration
extern "C" int qrw_control_pre(qrw_handle h, int32_t k, const double* d_joy_vref, const double* d_q_filt,
                               const double* d_v_filt, const double* d_rpy, const int32_t* d_code, int32_t code_scalar,
                               const double* d_x_f_mpc, double* d_q, double* d_v, double* d_hv, double* d_vref,
                               double* d_oRh_oTh, double* d_xref, double* d_fsteps, double* d_gait, double* d_target,
                               double* d_feet_pva, double* d_contacts, double* d_x_f_wbc, double* d_q_wbc, double* d_b_v,
                               double* d_f_cmd, double* d_feet_cmd, void* stream) {
  if (!h || !h->plan_ready) return fail(-1, "qrw_control_pre: planner not initialised");
  DeviceScope dev_scope__(h->cfg.device);  // launches and copies go to the handle's GPU, the caller's current device is restored
  if (!d_joy_vref || !d_q_filt || !d_v_filt || !d_rpy || !d_q || !d_v || !d_hv || !d_vref || !d_xref || !d_feet_pva)
    return fail(-1, "qrw_control_pre: null argument");
  if (d_x_f_mpc && (!d_q_wbc || !d_b_v || !d_feet_cmd)) return fail(-1, "qrw_control_pre: null WBC target output");
  qrw::ControllerArgs cu, cw;
  ctrl_common(h, cu, qrw::kCtrlUpdateState);
  cu.in0 = d_joy_vref; cu.in1 = d_q_filt; cu.in2 = d_v_filt; cu.in3 = d_rpy;
  cu.out0 = d_q; cu.out1 = d_v; cu.out2 = d_hv; cu.out3 = d_vref; cu.out4 = d_oRh_oTh;
  qrw::PlannerArgs p;
  planner_common(h, p);
  p.mode = qrw::kPlanGait | qrw::kPlanFootsteps | qrw::kPlanTraj | qrw::kPlanState;
  p.k = k;
  p.refresh = ((k % p.k_mpc) == 0 && k != 0) ? 1 : 0;
  p.k_footsteps = p.k_mpc - k % p.k_mpc;
  p.q7 = d_q; p.q_ld = 19; p.hv = d_hv; p.vref = d_vref; p.code = d_code; p.code_scalar = code_scalar;
  p.xref = d_xref; p.fsteps = d_fsteps; p.gait = d_gait; p.target = d_target; p.feet_pva = d_feet_pva;
  p.contacts = d_contacts;
  // no fsteps wanted = this iteration does not solve: of xref only column 0 and horizon step 1 are read (WBC target assembly)
  p.xref_steps = (!d_fsteps && d_x_f_mpc) ? 1 : 0;
  ctrl_common(h, cw, qrw::kCtrlWbcInputs);
  cw.in0 = d_x_f_mpc; cw.in1 = d_xref; cw.in2 = d_feet_pva; cw.in3 = d_v;
  cw.out0 = d_x_f_wbc; cw.out1 = d_q_wbc; cw.out2 = d_b_v; cw.out3 = d_f_cmd; cw.out4 = d_feet_cmd;
  debug_poison_lds((hipStream_t)stream);
  if (qrw::control_pre_launch(cu, p, cw, d_x_f_mpc ? 1 : 0, (hipStream_t)stream) != 0)
    return fail(-11, "qrw_control_pre: launch failed", hipGetLastError());
  note_launch(h, kFamPlan, (hipStream_t)stream);
  return 0;
}

// ------------------------------------------------------------------ a non-solving iteration on bound buffers
extern "C" int qrw_iteration_bind(qrw_handle h, const qrw_iteration_buffers* b) {
  if (!h || !b) return fail(-1, "qrw_iteration_bind: null argument");
  if (!h->plan_ready) return fail(-1, "qrw_iteration_bind: planner not initialised");
  if (!b->d_joy_vref || !b->d_q_filt || !b->d_v_filt || !b->d_rpy || !b->d_v_secu || !b->d_q || !b->d_v || !b->d_hv || !b->d_vref ||
      !b->d_xref || !b->d_feet_pva || !b->d_contacts || !b->d_q_wbc || !b->d_b_v || !b->d_f_cmd || !b->d_feet_cmd || !b->d_result)
    return fail(-1, "qrw_iteration_bind: null buffer (only d_code, d_oRh_oTh, d_target, d_x_f_wbc and the WBC outputs other than "
                    "d_result may be NULL)");
  h->iter_bufs = *b;
  h->iter_bound = true;
  return 0;
}

extern "C" int qrw_iteration_step(qrw_handle h, int32_t k, const double* d_x_f_mpc, void* stream) {
  if (!h || !d_x_f_mpc) return fail(-1, "qrw_iteration_step: null argument");
  if (!h->iter_bound) return fail(-1, "qrw_iteration_step: no buffers bound (qrw_iteration_bind)");
  const qrw_iteration_buffers& b = h->iter_bufs;
  const size_t plane = (size_t)h->cfg.batch * 12;  // d_feet_cmd [3][B][3][4]: the planes are the WBC's pgoals / vgoals / agoals
  int rc = qrw_control_pre(h, k, b.d_joy_vref, b.d_q_filt, b.d_v_filt, b.d_rpy, b.d_code, b.code_scalar, d_x_f_mpc, b.d_q, b.d_v, b.d_hv,
                           b.d_vref, b.d_oRh_oTh, b.d_xref, nullptr, nullptr, b.d_target, b.d_feet_pva, b.d_contacts, b.d_x_f_wbc,
                           b.d_q_wbc, b.d_b_v, b.d_f_cmd, b.d_feet_cmd, stream);
  if (rc) return rc;
  return qrw_wbc_compute_result(h, b.d_q_wbc, b.d_b_v, b.d_f_cmd, b.d_contacts, b.d_feet_cmd, b.d_feet_cmd + plane, b.d_feet_cmd + 2 * plane,
                                b.d_tau_ff, b.d_qdes, b.d_vdes, b.d_f_with_delta, b.d_ddq_res, b.d_feet, b.d_q_filt, b.d_v_secu, b.d_result,
                                b.d_error_flag, stream);
}
