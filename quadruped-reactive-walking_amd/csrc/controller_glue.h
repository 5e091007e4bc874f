// Element-wise glue of Controller.compute (/root/reference/scripts/Controller.py) as per-instance device functions,
// shared by controller_kernel.hip (one launch per piece) and the fused control-iteration kernels (planner_kernel.hip:
// update_state + planners [+ WBC target assembly] in one launch).  ControllerArgs carries the operands (qrw_kernels.h).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "qrw_kernels.h"

namespace qrw {
namespace glue {

enum { cQX = 0, cQY = 1, cYAW = 2, cPCMD = 3, cVCMD = 15, cQDES = 27, cVDES = 39, cERR = 51, cVREF = 52 };

struct CS {
  double* base;
  size_t stride;
  __device__ __forceinline__ double& operator()(int item) const { return base[(size_t)item * stride]; }
};
__device__ __forceinline__ CS state_of(const ControllerArgs& a, int b) {
  CS s;
  s.base = a.cs + b;
  s.stride = (size_t)a.B;
  return s;
}
__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// Controller.updateState (scripts/Controller.py:381-426): in0 joystick v_ref [6], in1 q_filt [19], in2 v_filt [18], in3 RPY [3]
// Every operand is read before the first store: the compiler must keep a load behind any earlier store that might alias it,
// and with one thread per instance each such load is a full HBM round trip on the critical path.
__device__ __forceinline__ void update_state(const ControllerArgs& a, int b) {
  const CS s = state_of(a, b);
  const double dt = a.dt_wbc;
  double jv[6], qf[13], vf[18], rpy[3];  // qf: element 2 and the joints 7..18
  {
    const double* pj = a.in0 + (size_t)b * 6;
    const double* pq = a.in1 + (size_t)b * 19;
    const double* pv = a.in2 + (size_t)b * 18;
    const double* pr = a.in3 + (size_t)b * 3;
#pragma unroll
    for (int i = 0; i < 6; i++) jv[i] = pj[i];
    qf[0] = pq[2];
#pragma unroll
    for (int i = 0; i < 12; i++) qf[1 + i] = pq[7 + i];
#pragma unroll
    for (int i = 0; i < 18; i++) vf[i] = pv[i];
#pragma unroll
    for (int i = 0; i < 3; i++) rpy[i] = pr[i];
  }
  const double yaw0 = s(cYAW), qx0 = s(cQX), qy0 = s(cQY);
#pragma unroll
  for (int i = 0; i < 6; i++) s(cVREF + i) = jv[i];
  const double c0 = cos(yaw0), s0 = sin(yaw0);
  const double qx = qx0 + (c0 * jv[0] + -s0 * jv[1]) * dt;
  const double qy = qy0 + (s0 * jv[0] + c0 * jv[1]) * dt;
  s(cQX) = qx; s(cQY) = qy;
  const double yaw = yaw0 + jv[5] * dt;
  s(cYAW) = yaw;
  double* q = a.out0 + (size_t)b * 19;
  q[0] = qx; q[1] = qy; q[2] = qf[0];
  {  // EulerToQuaternion(roll, pitch, yaw_estim) (scripts/Estimator.py:672-684)
    const double sr = sin(rpy[0] / 2.), cr = cos(rpy[0] / 2.), sp = sin(rpy[1] / 2.), cp = cos(rpy[1] / 2.),
                 sy = sin(yaw / 2.), cy = cos(yaw / 2.);
    q[3] = sr * cp * cy - cr * sp * sy;
    q[4] = cr * sp * cy + sr * cp * sy;
    q[5] = cr * cp * sy - sr * sp * cy;
    q[6] = cr * cp * cy + sr * sp * sy;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) q[7 + i] = qf[1 + i];
  double* v = a.out1 + (size_t)b * 18;
#pragma unroll
  for (int i = 0; i < 18; i++) v[i] = vf[i];
  {  // hRb = EulerToRotation(roll, pitch, 0) = Ry(pitch) Rx(roll) (scripts/utils_mpc.py:87-107)
    const double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]);
    const double R[9] = {cp, sp * sr, sp * cr, 0.0, cr, -sr, -sp, cp * sr, cp * cr};
    double* hv = a.out2 + (size_t)b * 6;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      hv[r] = R[r * 3] * vf[0] + R[r * 3 + 1] * vf[1] + R[r * 3 + 2] * vf[2];
      hv[3 + r] = R[r * 3] * vf[3] + R[r * 3 + 1] * vf[4] + R[r * 3 + 2] * vf[5];
    }
  }
  if (a.out3) {
#pragma unroll
    for (int i = 0; i < 6; i++) a.out3[(size_t)b * 6 + i] = jv[i];
  }
  if (a.out4) {  // oRh (9) | oTh (3)
    const double c = cos(yaw), sn = sin(yaw);
    double* o = a.out4 + (size_t)b * 12;
    o[0] = c; o[1] = -sn; o[2] = 0; o[3] = sn; o[4] = c; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = 1;
    o[9] = qx; o[10] = qy; o[11] = 0.0;
  }
}

// WBC target assembly (scripts/Controller.py:258-296): in0 x_f_mpc [24][N], in1 xref [12][N+1], in2 feet pva [3][3][4], in3 v [18]
// Operands are read in two batches ahead of the stores (see update_state).
__device__ __forceinline__ void wbc_inputs(const ControllerArgs& a, int b) {
  const CS s = state_of(a, b);
  const double dt = a.dt_wbc, h_ref = a.h_ref;
  const int N = a.n_steps;
  const double* xf = a.in0 + (size_t)b * 24 * N;
  const double* xr = a.in1 + (size_t)b * 12 * (N + 1);
  const double* pva = a.in2 + (size_t)b * 36;
  double xf0[24], xr1[6], qdes[12], vref[6], vdes[12];
#pragma unroll
  for (int i = 0; i < 24; i++) xf0[i] = xf[i * N];                       // column 0 of the MPC result
#pragma unroll
  for (int i = 0; i < 6; i++) xr1[i] = xr[(6 + i) * (N + 1) + 1];       // rows 6..11, column 1 of xref
#pragma unroll
  for (int i = 0; i < 12; i++) { qdes[i] = s(cQDES + i); vdes[i] = s(cVDES + i); }
#pragma unroll
  for (int i = 0; i < 6; i++) vref[i] = s(cVREF + i);
  const double yaw = s(cYAW), qx = s(cQX), qy = s(cQY);
  double pcmd[12], vcmd[12], pv_[36];
#pragma unroll
  for (int i = 0; i < 12; i++) { pcmd[i] = s(cPCMD + i); vcmd[i] = s(cVCMD + i); }
#pragma unroll
  for (int i = 0; i < 36; i++) pv_[i] = pva[i];

  double* xw = a.out0 ? a.out0 + (size_t)b * 24 : nullptr;
  if (xw) {
    xw[0] = dt * xr1[0];
    xw[1] = dt * xr1[1];
    xw[2] = h_ref; xw[3] = 0.0; xw[4] = 0.0;
    xw[5] = dt * xr1[5];
#pragma unroll
    for (int i = 6; i < 12; i++) xw[i] = xr1[i - 6];
#pragma unroll
    for (int i = 12; i < 24; i++) xw[i] = xf0[i];
  }
  double* qw = a.out1 + (size_t)b * 19;
#pragma unroll
  for (int i = 0; i < 7; i++) qw[i] = (i == 2) ? h_ref : (i == 6) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 12; i++) qw[7 + i] = qdes[i];
  double* bv = a.out2 + (size_t)b * 18;
#pragma unroll
  for (int i = 0; i < 6; i++) bv[i] = vref[i];
#pragma unroll
  for (int i = 0; i < 12; i++) bv[6 + i] = vdes[i];
  if (a.out3) {
#pragma unroll
    for (int i = 0; i < 12; i++) a.out3[(size_t)b * 12 + i] = xf0[12 + i];
  }
  const double c = cos(yaw), sn = sin(yaw);
  const double w[3] = {vref[3], vref[4], vref[5]};
  const double vl[3] = {vref[0], vref[1], vref[2]};
  double* fc = a.out4 + (size_t)b * 12;  // planes p | v | a, each [B][3][4]
  const size_t pl = (size_t)a.B * 12;
#pragma unroll
  for (int f = 0; f < 4; f++) {
    const double pp[3] = {pcmd[f], pcmd[4 + f], pcmd[8 + f]};
    const double pv[3] = {vcmd[f], vcmd[4 + f], vcmd[8 + f]};
    const double pos[3] = {pv_[0 * 4 + f], pv_[1 * 4 + f], pv_[2 * 4 + f]};
    const double vel[3] = {pv_[12 + 0 * 4 + f], pv_[12 + 1 * 4 + f], pv_[12 + 2 * 4 + f]};
    const double acc[3] = {pv_[24 + 0 * 4 + f], pv_[24 + 1 * 4 + f], pv_[24 + 2 * 4 + f]};
    double wxp[3], wxwxp[3], wxv[3];
    cross3(w, pp, wxp);
    cross3(w, wxp, wxwxp);
    cross3(w, pv, wxv);
    // oRh' = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    const double ra[3] = {c * acc[0] + sn * acc[1], -sn * acc[0] + c * acc[1], acc[2]};
    const double rv[3] = {c * vel[0] + sn * vel[1], -sn * vel[0] + c * vel[1], vel[2]};
    const double dp[3] = {pos[0] - 0.0 - qx, pos[1] - 0.0 - qy, pos[2] - h_ref - 0.0};
    const double rp[3] = {c * dp[0] + sn * dp[1], -sn * dp[0] + c * dp[1], dp[2]};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double an = ra[r] - wxwxp[r] - 2 * wxv[r];
      const double vn = (rv[r] - vl[r]) - wxp[r];
      fc[2 * pl + r * 4 + f] = an;
      fc[pl + r * 4 + f] = vn;
      fc[r * 4 + f] = rp[r];
      s(cVCMD + r * 4 + f) = vn;
      s(cPCMD + r * 4 + f) = rp[r];
    }
  }
}

// Result + security_check (scripts/Controller.py:306-310,341-365): in0 tau_ff [12], in1 qdes [19], in2 vdes [18], in3 q_filt [19], in4 v_secu [12]
__device__ __forceinline__ void result(const ControllerArgs& a, int b) {
  const CS s = state_of(a, b);
  const double* tau = a.in0 + (size_t)b * 12;
  const double* qd = a.in1 + (size_t)b * 19;
  const double* vd = a.in2 + (size_t)b * 18;
  const double* qf = a.in3 + (size_t)b * 19;
  const double* vs = a.in4 + (size_t)b * 12;
  int err = (int)s(cERR);
  if (err == 0) {  // the WBC ran this iteration: keep its references for the next one (Controller.py:282,287)
    for (int i = 0; i < 12; i++) { s(cQDES + i) = qd[7 + i]; s(cVDES + i) = vd[6 + i]; }
    const double qsec[3] = {M_PI * 0.4, M_PI * 80 / 180, M_PI};
    bool e1 = false, e2 = false, e3 = false, e4 = false;
    for (int i = 0; i < 12; i++) {
      e1 = e1 || (fabs(qf[7 + i]) > qsec[i % 3]);
      e2 = e2 || (fabs(vs[i]) > 50);
      e3 = e3 || (fabs(tau[i]) > 8);
      // NOT in the reference: its three comparisons are blind to NaN (`NaN > 8` is false), so a non-finite command -- e.g. the
      // forces of an MPC solve that never finished -- would be passed to the motors.  A fourth sticky code stops the robot.
      e4 = e4 || !(fabs(tau[i]) <= 1.7976931348623157e308) || !(fabs(qd[7 + i]) <= 1.7976931348623157e308) ||
           !(fabs(vd[6 + i]) <= 1.7976931348623157e308);
    }
    if (e4) err = 4;
    if (e1) err = 1;
    if (e2) err = 2;
    if (e3) err = 3;
    s(cERR) = (double)err;
  }
  double* r = a.out0 + (size_t)b * 60;  // P | D | q_des | v_des | tau_ff
  for (int i = 0; i < 12; i++) {
    if (err == 0) {
      r[i] = 3.0; r[12 + i] = 0.2; r[24 + i] = qd[7 + i]; r[36 + i] = vd[6 + i]; r[48 + i] = 0.8 * tau[i];
    } else {
      r[i] = 0.0; r[12 + i] = 0.1; r[24 + i] = 0.0; r[36 + i] = 0.0; r[48 + i] = 0.0;
    }
  }
  if (a.iout) a.iout[b] = err;
}

// MPC_Wrapper.solve bookkeeping on the result the control loop currently reads (scripts/MPC_Wrapper.py:89-102), in the
// reference only observable in the asynchronous mode: past the first iterations the force rows 12..23 are rolled one
// horizon step to the left (np.roll: the first column wraps to the end), and when the last gait row before the first
// all-zero one is in another phase than row 0 the freed last column gets m g / n_contacts (mass 2.5, :98) on its
// stance feet.  in0 = gait [B][N_gait][4] (doubles, FootstepPlanner / Gait layout), out0 = x_f_mpc [B][24][N] in place.
__device__ __forceinline__ void mpc_result_shift(const ControllerArgs& a, int b) {
  const int N = a.n_steps, Ng = a.n_gait;
  double* xf = a.out0 + (size_t)b * 24 * N;
  const double* g = a.in0 + (size_t)b * Ng * 4;
  const int r_end = (12 + N < 24) ? 12 + N : 24;  // the reference slices rows 12:(12+n_steps) of the 24-row array (:90)
  for (int r = 12; r < r_end; r++) {
    const double first = xf[r * N];
    for (int c = 0; c + 1 < N; c++) xf[r * N + c] = xf[r * N + c + 1];
    xf[r * N + N - 1] = first;
  }
  int pt = 0;
  while (pt < Ng && (g[pt * 4] != 0.0 || g[pt * 4 + 1] != 0.0 || g[pt * 4 + 2] != 0.0 || g[pt * 4 + 3] != 0.0)) pt++;
  const int last = pt > 0 ? pt - 1 : Ng - 1;  // gait[pt-1] with Python's negative index when row 0 is already zero
  bool differs = false;
  double n_ctc = 0.0;
  for (int i = 0; i < 4; i++) {
    differs = differs || (g[i] != g[last * 4 + i]);
    n_ctc += g[last * 4 + i];
  }
  if (differs) {
    const double F = 9.81 * 2.5 / n_ctc;
    for (int i = 0; i < 4; i++) {
      xf[(12 + 3 * i) * N + N - 1] = 0.0;
      xf[(13 + 3 * i) * N + N - 1] = 0.0;
      xf[(14 + 3 * i) * N + N - 1] = (g[last * 4 + i] == 1.0) ? F : 0.0;
    }
  }
}

}  // namespace glue
}  // namespace qrw
