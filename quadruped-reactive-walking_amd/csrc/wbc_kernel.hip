// Whole-body-control step for a batch of Solo12 instances — gfx950 (MI355X).
//
// Replaces, per instance, wbc_controller.compute (/root/reference/scripts/QP_WBC.py:52-131):
//   Solo12InvKin.refreshAndCompute   scripts/solo12InvKin.py:44-69   (fixed-base foot kinematics)
//   InvKin::refreshAndCompute        src/InvKin.cpp:23-73
//   pin.crba (neutral, diagonal), computeJointJacobians/getFrameJacobian, pin.rnea x2
//                                    scripts/QP_WBC.py:89-116
//   QPWBC::run                       src/QPWBC.cpp:310-390 (compute_matrices :481-498, update_PQ
//                                    :520-537, call_solver :213-275 = OSQP 0.6.x, retrieve_result :277-297)
//
// Two kernels.  wbc16_kernel (round 4, below): SIXTEEN LANES per robot instance, the default for the full compute (27.9 us per
// 4096 robots against 43.2).  wbc_kernel: one quad per instance; backs the stand-alone modes 1-3 and QRW_WBC16=0.
// Mapping of wbc_kernel: lane = 4*instance_in_wave + foot, i.e. ONE QUAD PER ROBOT INSTANCE and 16 instances per
// wavefront.  Each lane runs the kinematics / Newton-Euler recursion of its own leg; the base
// wrench, the QP data and the 12-variable ADMM are shared inside the quad with DPP quad
// permutes (no LDS).  The 20 friction-cone rows split 5 per lane and only touch that lane's 3
// force variables, so A_qp x and A_qp' y are lane-local.
//
// Rigid-body formulation: classical Newton-Euler in base-frame coordinates (angular velocity /
// acceleration, CoM accelerations, per-link force and moment), NOT the spatial-algebra recursion
// Pinocchio uses — only the model constants (include/qrw_solo12_model.h) are shared with the test checker.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/qrw_solo12_model.h"
#include "qrw_device.h"
#include "controller_glue.h"
#include "qrw_kernels.h"

// Diagnostic build (-DQRW_PROFILE_WBC): shader-clock stamps of wbc_kernel's phases, per workgroup (lane 0), read back by
// qrw_wbc_get_phase_cycles -- scripts/gpu_wbc_phases.py, profiles/r4_wbc_phase_cycles.txt
#ifdef QRW_PROFILE_WBC
__device__ unsigned long long g_wbc_ph[16 * 8192];
#define WPH(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_wbc_ph[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int qrw_wbc_get_phase_cycles(unsigned long long* out, int n_blocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wbc_ph), (size_t)n_blocks * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define WPH(i)
#endif

namespace qrw {

namespace {

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct M3 {
  V3 r0, r1, r2;  // rows
};
__device__ __forceinline__ V3 mul(const M3& m, V3 v) { return mk(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
__device__ __forceinline__ V3 mulT(const M3& m, V3 v) {
  return mk(m.r0.x * v.x + m.r1.x * v.y + m.r2.x * v.z, m.r0.y * v.x + m.r1.y * v.y + m.r2.y * v.z,
            m.r0.z * v.x + m.r1.z * v.y + m.r2.z * v.z);
}
__device__ __forceinline__ V3 rotx(double c, double s, V3 v) { return mk(v.x, c * v.y - s * v.z, s * v.y + c * v.z); }
__device__ __forceinline__ V3 roty(double c, double s, V3 v) { return mk(c * v.x + s * v.z, v.y, -s * v.x + c * v.z); }

// Model constants of one leg.  The numbers come from include/qrw_solo12_model.h (the ONE data file):
// the kernel takes the FL leg of a compile-time copy of that initialiser as immediates and applies the mirror signs of
// the leg it works on; a static_assert proves that the other three legs of the data file ARE those mirror images, so
// a model file that breaks the symmetry cannot be compiled into a kernel that silently ignores it.
struct LegC {
  V3 haa, hfe, kfe, foot;
  double m[4];
  V3 com[4];
  double I[4][6];
};
constexpr qrw_solo12_model kModel = QRW_SOLO12_MODEL_INIT;
// sign of entry e of (com | inertia) of link `link` (0 shoulder, 1 upper, 2 lower, 3 foot) for mirror signs (sx, sy)
constexpr double mirror_sign(int link, bool inertia, int e, double sx, double sy) {
  if (!inertia) return (e == 0 && link == 0) ? sx : (e == 1 && link != 3) ? sy : 1.0;
  if (link == 0) return e == 1 ? sx * sy : 1.0;            // shoulder: ixy
  if (link == 1 || link == 2) return e == 4 ? sy : 1.0;    // upper / lower leg: iyz
  return 1.0;
}
constexpr bool same(double a, double b) { return a == b; }
constexpr bool link_is_mirror(const qrw_link_inertial& a, const qrw_link_inertial& fl, int link, double sx, double sy) {
  bool ok = same(a.mass, fl.mass);
  for (int e = 0; e < 3; e++) ok = ok && same(a.com[e], mirror_sign(link, false, e, sx, sy) * fl.com[e]);
  for (int e = 0; e < 6; e++) ok = ok && same(a.inertia[e], mirror_sign(link, true, e, sx, sy) * fl.inertia[e]);
  return ok;
}
constexpr bool leg_is_mirror(const qrw_leg_model& a, const qrw_leg_model& fl, double sx, double sy) {
  bool ok = same(a.haa_xyz[0], sx * fl.haa_xyz[0]) && same(a.haa_xyz[1], sy * fl.haa_xyz[1]) && same(a.haa_xyz[2], fl.haa_xyz[2]);
  const double* o[3] = {a.hfe_xyz, a.kfe_xyz, a.foot_xyz};
  const double* f[3] = {fl.hfe_xyz, fl.kfe_xyz, fl.foot_xyz};
  for (int i = 0; i < 3; i++) ok = ok && same(o[i][0], f[i][0]) && same(o[i][1], sy * f[i][1]) && same(o[i][2], f[i][2]);
  return ok && link_is_mirror(a.shoulder, fl.shoulder, 0, sx, sy) && link_is_mirror(a.upper, fl.upper, 1, sx, sy) &&
         link_is_mirror(a.lower, fl.lower, 2, sx, sy) && link_is_mirror(a.foot, fl.foot, 3, sx, sy);
}
static_assert(leg_is_mirror(kModel.leg[0], kModel.leg[0], 1.0, 1.0) && leg_is_mirror(kModel.leg[1], kModel.leg[0], 1.0, -1.0) &&
                  leg_is_mirror(kModel.leg[2], kModel.leg[0], -1.0, 1.0) && leg_is_mirror(kModel.leg[3], kModel.leg[0], -1.0, -1.0),
              "include/qrw_solo12_model.h: legs FR/HL/HR are not the mirror images of FL that wbc_kernel assumes");

__device__ __forceinline__ LegC leg_consts(int j) {
  LegC L;
  const double sx = (j < 2) ? 1.0 : -1.0, sy = (j & 1) ? -1.0 : 1.0;
  constexpr qrw_leg_model FL = kModel.leg[0];
  L.haa = mk(sx * FL.haa_xyz[0], sy * FL.haa_xyz[1], FL.haa_xyz[2]);
  L.hfe = mk(FL.hfe_xyz[0], sy * FL.hfe_xyz[1], FL.hfe_xyz[2]);
  L.kfe = mk(FL.kfe_xyz[0], sy * FL.kfe_xyz[1], FL.kfe_xyz[2]);
  L.foot = mk(FL.foot_xyz[0], sy * FL.foot_xyz[1], FL.foot_xyz[2]);
  constexpr qrw_link_inertial LK[4] = {FL.shoulder, FL.upper, FL.lower, FL.foot};
#pragma unroll
  for (int l = 0; l < 4; l++) {
    L.m[l] = LK[l].mass;
    L.com[l] = mk(mirror_sign(l, false, 0, sx, sy) * LK[l].com[0], mirror_sign(l, false, 1, sx, sy) * LK[l].com[1], LK[l].com[2]);
#pragma unroll
    for (int e = 0; e < 6; e++) L.I[l][e] = mirror_sign(l, true, e, sx, sy) * LK[l].inertia[e];
  }
  return L;
}

// inertia (about the CoM, link axes) applied in base coordinates: R I R' w
__device__ __forceinline__ V3 inertia_apply(const double I[6], const M3& R, V3 w) {
  const V3 wl = mulT(R, w);
  const V3 hl = mk(I[0] * wl.x + I[1] * wl.y + I[2] * wl.z, I[1] * wl.x + I[3] * wl.y + I[4] * wl.z,
                   I[2] * wl.x + I[4] * wl.y + I[5] * wl.z);
  return mul(R, hl);
}

struct LegKin {
  V3 p0, p1, p2, pf;  // joint origins and foot, base frame
  V3 a0, a1;          // joint axes (a2 = a1)
  M3 R0, R1, R2;      // link rotations (base <- link)
  V3 J0, J1, J2;      // columns of the foot Jacobian (base frame)
};

__device__ __forceinline__ LegKin leg_kinematics(const LegC& C, const double q[3]) {
  LegKin K;
  const double c0 = cos(q[0]), s0 = sin(q[0]);
  const double c1 = cos(q[1]), s1 = sin(q[1]);
  const double c12 = cos(q[1] + q[2]), s12 = sin(q[1] + q[2]);
  // R0 = Rx(q0); R1 = R0 Ry(q1); R2 = R0 Ry(q1 + q2)
  K.R0.r0 = mk(1, 0, 0); K.R0.r1 = mk(0, c0, -s0); K.R0.r2 = mk(0, s0, c0);
  K.R1.r0 = mk(c1, 0, s1); K.R1.r1 = mk(s0 * s1, c0, -s0 * c1); K.R1.r2 = mk(-c0 * s1, s0, c0 * c1);
  K.R2.r0 = mk(c12, 0, s12); K.R2.r1 = mk(s0 * s12, c0, -s0 * c12); K.R2.r2 = mk(-c0 * s12, s0, c0 * c12);
  K.p0 = C.haa;
  K.p1 = K.p0 + mul(K.R0, C.hfe);
  K.p2 = K.p1 + mul(K.R1, C.kfe);
  K.pf = K.p2 + mul(K.R2, C.foot);
  K.a0 = mk(1, 0, 0);
  K.a1 = mk(0, c0, s0);
  K.J0 = cross(K.a0, K.pf - K.p0);
  K.J1 = cross(K.a1, K.pf - K.p1);
  K.J2 = cross(K.a1, K.pf - K.p2);
  return K;
}

// Newton-Euler for one leg hanging from a base that moves with (w_b, alpha_b) and whose origin has the
// classical acceleration a_b (gravity already folded in). Returns the leg's contribution to the base
// force / moment (about the base origin) and the three joint torques.
__device__ __forceinline__ void leg_newton_euler(const LegC& C, const LegKin& K, const double dq[3], const double ddq[3],
                                                 V3 wb, V3 alb, V3 ab, V3& Fsum, V3& Msum, double tau[3]) {
  // angular velocity / acceleration of the three links
  const V3 w0 = wb + dq[0] * K.a0;
  const V3 al0 = alb + dq[0] * cross(wb, K.a0) + ddq[0] * K.a0;
  const V3 w1 = w0 + dq[1] * K.a1;
  const V3 al1 = al0 + dq[1] * cross(w0, K.a1) + ddq[1] * K.a1;
  const V3 w2 = w1 + dq[2] * K.a1;
  const V3 al2 = al1 + dq[2] * cross(w1, K.a1) + ddq[2] * K.a1;
  // accelerations of the joint origins
  const V3 ap0 = ab + cross(alb, K.p0) + cross(wb, cross(wb, K.p0));
  const V3 d01 = K.p1 - K.p0, d12 = K.p2 - K.p1;
  const V3 ap1 = ap0 + cross(al0, d01) + cross(w0, cross(w0, d01));
  const V3 ap2 = ap1 + cross(al1, d12) + cross(w1, cross(w1, d12));
  // bodies: shoulder (link 0), upper leg (link 1), lower leg and foot (both on link 2)
  V3 F[4], Nn[4], c[4];
  {
    const V3 r = mul(K.R0, C.com[0]);
    c[0] = K.p0 + r;
    F[0] = C.m[0] * (ap0 + cross(al0, r) + cross(w0, cross(w0, r)));
    Nn[0] = inertia_apply(C.I[0], K.R0, al0) + cross(w0, inertia_apply(C.I[0], K.R0, w0));
  }
  {
    const V3 r = mul(K.R1, C.com[1]);
    c[1] = K.p1 + r;
    F[1] = C.m[1] * (ap1 + cross(al1, r) + cross(w1, cross(w1, r)));
    Nn[1] = inertia_apply(C.I[1], K.R1, al1) + cross(w1, inertia_apply(C.I[1], K.R1, w1));
  }
  {
    const V3 r = mul(K.R2, C.com[2]);
    c[2] = K.p2 + r;
    F[2] = C.m[2] * (ap2 + cross(al2, r) + cross(w2, cross(w2, r)));
    Nn[2] = inertia_apply(C.I[2], K.R2, al2) + cross(w2, inertia_apply(C.I[2], K.R2, w2));
  }
  {
    const V3 r = mul(K.R2, C.foot + C.com[3]);
    c[3] = K.p2 + r;
    F[3] = C.m[3] * (ap2 + cross(al2, r) + cross(w2, cross(w2, r)));
    Nn[3] = inertia_apply(C.I[3], K.R2, al2) + cross(w2, inertia_apply(C.I[3], K.R2, w2));
  }
  // joint torques: axis . (moment of all outboard bodies about the joint origin)
  V3 m2 = Nn[2] + cross(c[2] - K.p2, F[2]) + Nn[3] + cross(c[3] - K.p2, F[3]);
  tau[2] = dot(K.a1, m2);
  V3 m1 = Nn[1] + cross(c[1] - K.p1, F[1]) + Nn[2] + cross(c[2] - K.p1, F[2]) + Nn[3] + cross(c[3] - K.p1, F[3]);
  tau[1] = dot(K.a1, m1);
  V3 m0 = Nn[0] + cross(c[0] - K.p0, F[0]) + Nn[1] + cross(c[1] - K.p0, F[1]) + Nn[2] + cross(c[2] - K.p0, F[2]) +
          Nn[3] + cross(c[3] - K.p0, F[3]);
  tau[0] = dot(K.a0, m0);
  Fsum = F[0] + F[1] + F[2] + F[3];
  Msum = Nn[0] + cross(c[0], F[0]) + Nn[1] + cross(c[1], F[1]) + Nn[2] + cross(c[2], F[2]) + Nn[3] + cross(c[3], F[3]);
}

__device__ __forceinline__ V3 quad_sum3(V3 v) { return mk(quad_sum(v.x), quad_sum(v.y), quad_sum(v.z)); }

__device__ __forceinline__ void inv3x3(V3 c0, V3 c1, V3 c2, M3& inv) {
  // matrix with COLUMNS c0,c1,c2; cofactor inverse, rows returned
  const double a = c0.x, b = c1.x, c = c2.x, d = c0.y, e = c1.y, f = c2.y, g = c0.z, h = c1.z, i = c2.z;
  const double C00 = e * i - f * h, C10 = f * g - d * i, C20 = d * h - e * g;
  const double id = 1.0 / (a * C00 + b * C10 + c * C20);
  inv.r0 = mk(C00 * id, (c * h - b * i) * id, (b * f - c * e) * id);
  inv.r1 = mk(C10 * id, (a * i - c * g) * id, (c * d - a * f) * id);
  inv.r2 = mk(C20 * id, (b * g - a * h) * id, (a * e - b * d) * id);
}

// broadcast one lane's 3-vector to its quad, lane index compile-time
template <int J>
__device__ __forceinline__ void qb(const double in[3], double out[3]) {
#pragma unroll
  for (int t = 0; t < 3; t++) out[t] = quad_bcast<J>(in[t]);
}

// ------------------------------------------------------------------------------------------
// 12-variable / 20-row box QP of QPWBC solved with OSQP-0.6-style ADMM inside one quad.
//   min 1/2 x'Hx + g'x   s.t.  l <= G x <= u,  G block-diagonal (5x3 per foot), H rows 3j..3j+2 per lane
// State (scaled iterates, rho, previous g) persists across calls exactly like the OSQP workspace.
struct QpIo {
  double Hrow[3][12];  // rows of H owned by this lane
  double g[3];
  double lo[5], up[5];
};

__device__ __forceinline__ void cone_rows(const double f[3], double mu, double out[5]) {
  // G block of QPWBC.cpp:10-22: (-1,0,mu),(1,0,mu),(0,-1,mu),(0,1,mu),(0,0,1)
  out[0] = -f[0] + mu * f[2];
  out[1] = f[0] + mu * f[2];
  out[2] = -f[1] + mu * f[2];
  out[3] = f[1] + mu * f[2];
  out[4] = f[2];
}
__device__ __forceinline__ void cone_rows_t(const double w[5], double mu, double out[3]) {
  out[0] = -w[0] + w[1];
  out[1] = -w[2] + w[3];
  out[2] = mu * (w[0] + w[1] + w[2] + w[3]) + w[4];
}

__device__ void qp_solve(const QpIo& io, double* st, int j, bool valid, double sol[3], int& iter_out, int& status_out) {
  const double mu = 0.9;  // QPWBC.hpp:30
  const double sigma = 1e-6, alpha = 1.6;
  const double eps_abs = (double)(float)1e-5, eps_rel = (double)(float)1e-5;  // QPWBC.cpp:239-240
  const double eps_prim_inf = 1e-4, eps_dual_inf = 1e-4;
  // the warm-start state is read unconditionally, together with the flag that says whether it counts: one round trip
  // to HBM instead of two (padding quads read instance 0's)
  WPH(3);
  const double init_flag = st[kWsInit];
  double x[3], z[5], y[5], rho = st[kWsRho], gprev[3];
#pragma unroll
  for (int t = 0; t < 3; t++) { x[t] = st[kWsX + 3 * j + t]; gprev[t] = st[kWsG + 3 * j + t]; }
#pragma unroll
  for (int c = 0; c < 5; c++) { z[c] = st[kWsZ + 5 * j + c]; y[c] = st[kWsY + 5 * j + c]; }
  const bool first = !valid || !(init_flag != 0.0);  // padding quads always cold-start
  if (first) {
    rho = 0.1;
#pragma unroll
    for (int t = 0; t < 3; t++) { x[t] = 0.0; gprev[t] = io.g[t]; }
#pragma unroll
    for (int c = 0; c < 5; c++) z[c] = y[c] = 0.0;
  }
  // ---- scale_data: on setup q is the current g; inside osqp_update_P it is still the PREVIOUS call's g
  // (osqp_update_lin_cost runs afterwards, QPWBC.cpp:258-261)
  double D[3] = {1, 1, 1}, E[5] = {1, 1, 1, 1, 1}, cs = 1.0;
  const double Gabs[5][3] = {{1, 0, mu}, {1, 0, mu}, {0, 1, mu}, {0, 1, mu}, {0, 0, 1}};
  for (int pass = 0; pass < 10; pass++) {
    double Dall[12];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      Dall[t] = quad_bcast<0>(D[t]); Dall[3 + t] = quad_bcast<1>(D[t]);
      Dall[6 + t] = quad_bcast<2>(D[t]); Dall[9 + t] = quad_bcast<3>(D[t]);
    }
    double nD[3], nE[5];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      double v = 0.0;
#pragma unroll
      for (int cb = 0; cb < 12; cb++) v = fmax(v, fabs(cs * D[t] * io.Hrow[t][cb] * Dall[cb]));
#pragma unroll
      for (int c = 0; c < 5; c++) v = fmax(v, E[c] * Gabs[c][t] * D[t]);
      nD[t] = v;
    }
#pragma unroll
    for (int c = 0; c < 5; c++) {
      double v = 0.0;
#pragma unroll
      for (int t = 0; t < 3; t++) v = fmax(v, E[c] * Gabs[c][t] * D[t]);
      nE[c] = v;
    }
#pragma unroll
    // rsqrt (1-2 ulp) instead of OSQP's 1.0 / sqrt(): the equilibration is a preconditioner, iteration counts and results
    // still agree with the oracle (tests/test_gpu_wbc.py), and 80 square-root-and-divide chains per call go (as mpc_kernel.hip)
    for (int t = 0; t < 3; t++) D[t] *= rsqrt(limit_scaling(nD[t]));
#pragma unroll
    for (int c = 0; c < 5; c++) E[c] *= rsqrt(limit_scaling(nE[c]));
#pragma unroll
    for (int t = 0; t < 3; t++) {
      Dall[t] = quad_bcast<0>(D[t]); Dall[3 + t] = quad_bcast<1>(D[t]);
      Dall[6 + t] = quad_bcast<2>(D[t]); Dall[9 + t] = quad_bcast<3>(D[t]);
    }
    double colsum = 0.0, qn = 0.0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      double v = 0.0;
#pragma unroll
      for (int cb = 0; cb < 12; cb++) v = fmax(v, fabs(cs * D[t] * io.Hrow[t][cb] * Dall[cb]));
      colsum += v;
      qn = fmax(qn, fabs(cs * D[t] * gprev[t]));
    }
    double ct = quad_sum(colsum) * (1.0 / 12.0);
    qn = limit_scaling(quad_max(qn));
    ct = fmax(ct, qn);
    ct = limit_scaling(ct);
    cs *= 1.0 / ct;
  }
  WPH(4);
  const double cinv = 1.0 / cs;
  double iD[3], iE[5], ls[5], us[5];
#pragma unroll
  for (int t = 0; t < 3; t++) iD[t] = 1.0 / D[t];
#pragma unroll
  for (int c = 0; c < 5; c++) { iE[c] = 1.0 / E[c]; ls[c] = E[c] * io.lo[c]; us[c] = E[c] * io.up[c]; }
  rho = fmin(fmax(rho, kRhoMin), kRhoMax);

  double Ki[3][12];
  bool need_factor = true;
  int iter, status = kStatusUnsolved;
  double last_np = 0.0, last_nd = 0.0, pri_res = 0.0, dua_res = 0.0;
  const int max_iter = 4000;
  // padding quads (batch not a multiple of 16) skip the solve: their data is a copy of instance 0 with an identity
  // KKT inverse, which never converges and would hold the whole wavefront for max_iter iterations
  for (iter = 1; valid && iter <= max_iter; iter++) {
    if (need_factor) {  // Khat = c H + sigma D^-2 + rho G' E^2 G, inverted by Gauss-Jordan inside the quad
      need_factor = false;
      double om[5];
#pragma unroll
      for (int c = 0; c < 5; c++) om[c] = rho * E[c] * E[c];
      const double s4 = om[0] + om[1] + om[2] + om[3];
      const double cone[3][3] = {{om[0] + om[1], 0.0, mu * (om[1] - om[0])},
                                 {0.0, om[2] + om[3], mu * (om[3] - om[2])},
                                 {mu * (om[1] - om[0]), mu * (om[3] - om[2]), mu * mu * s4 + om[4]}};
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int cb = 0; cb < 12; cb++) {
          double v = cs * io.Hrow[t][cb];
          if (cb / 3 == j) {
            v += cone[t][cb % 3];
            if (cb % 3 == t) v += sigma * iD[t] * iD[t];
          }
          Ki[t][cb] = valid ? v : ((cb / 3 == j && cb % 3 == t) ? 1.0 : 0.0);
        }
#pragma unroll
      for (int p = 0; p < 12; p++) {
        const int jp = p / 3, tp = p % 3;
        double prow[12];
#pragma unroll
        for (int cb = 0; cb < 12; cb++) {
          const double src = Ki[tp][cb];
          prow[cb] = (jp == 0) ? quad_bcast<0>(src) : (jp == 1) ? quad_bcast<1>(src) : (jp == 2) ? quad_bcast<2>(src) : quad_bcast<3>(src);
        }
        const double d = 1.0 / prow[p];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const bool isp = (j == jp) && (t == tp);
          const double fcol = Ki[t][p];
#pragma unroll
          for (int cb = 0; cb < 12; cb++) {
            double v;
            if (cb == p) v = isp ? d : -fcol * d;
            else v = isp ? prow[cb] * d : Ki[t][cb] - fcol * prow[cb] * d;
            Ki[t][cb] = v;
          }
        }
      }
    }
    const double rho_inv = 1.0 / rho;
    // rhs (hatted): sigma x / D - c g + G' E (rho z - y)
    double w[5], gt[3], r[3], rall[12];
#pragma unroll
    for (int c = 0; c < 5; c++) w[c] = E[c] * (rho * (z[c] - rho_inv * y[c]));
    cone_rows_t(w, mu, gt);
#pragma unroll
    for (int t = 0; t < 3; t++) r[t] = sigma * x[t] * iD[t] - cs * io.g[t] + gt[t];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      rall[t] = quad_bcast<0>(r[t]); rall[3 + t] = quad_bcast<1>(r[t]);
      rall[6 + t] = quad_bcast<2>(r[t]); rall[9 + t] = quad_bcast<3>(r[t]);
    }
    double xh[3], zt[5], cv[5], dx[3], dy[5];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      double v = 0.0;
#pragma unroll
      for (int cb = 0; cb < 12; cb++) v += Ki[t][cb] * rall[cb];
      xh[t] = v;
    }
    cone_rows(xh, mu, cv);
#pragma unroll
    for (int c = 0; c < 5; c++) zt[c] = E[c] * cv[c];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const double xn = alpha * (xh[t] * iD[t]) + (1.0 - alpha) * x[t];
      dx[t] = xn - x[t];
      x[t] = xn;
    }
#pragma unroll
    for (int c = 0; c < 5; c++) {
      const double zr = alpha * zt[c] + (1.0 - alpha) * z[c];
      double zn = zr + rho_inv * y[c];
      zn = fmin(fmax(zn, ls[c]), us[c]);
      dy[c] = rho * (zr - zn);
      y[c] += dy[c];
      z[c] = zn;
    }
    if (iter % 25 == 0) {
      double xs[3], xall[12];
#pragma unroll
      for (int t = 0; t < 3; t++) xs[t] = D[t] * x[t];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        xall[t] = quad_bcast<0>(xs[t]); xall[3 + t] = quad_bcast<1>(xs[t]);
        xall[6 + t] = quad_bcast<2>(xs[t]); xall[9 + t] = quad_bcast<3>(xs[t]);
      }
      double pres = 0, nz = 0, nax = 0, pres_s = 0, nz_s = 0, nax_s = 0;
      cone_rows(xs, mu, cv);
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const double axs = E[c] * cv[c], rs = axs - z[c];
        pres_s = fmax(pres_s, fabs(rs)); nz_s = fmax(nz_s, fabs(z[c])); nax_s = fmax(nax_s, fabs(axs));
        pres = fmax(pres, fabs(iE[c] * rs)); nz = fmax(nz, fabs(iE[c] * z[c])); nax = fmax(nax, fabs(iE[c] * axs));
      }
      double ey[5], aty[3];
#pragma unroll
      for (int c = 0; c < 5; c++) ey[c] = E[c] * y[c];
      cone_rows_t(ey, mu, aty);
      double dres = 0, naty = 0, npx = 0, nq = 0, dres_s = 0, naty_s = 0, npx_s = 0, nq_s = 0;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        double px = 0.0;
#pragma unroll
        for (int cb = 0; cb < 12; cb++) px += io.Hrow[t][cb] * xall[cb];
        px *= cs;
        const double qh = cs * io.g[t];
        dres = fmax(dres, fabs(px + qh + aty[t])); naty = fmax(naty, fabs(aty[t])); npx = fmax(npx, fabs(px));
        nq = fmax(nq, fabs(qh));
        dres_s = fmax(dres_s, fabs(D[t] * (px + qh + aty[t]))); naty_s = fmax(naty_s, fabs(D[t] * aty[t]));
        npx_s = fmax(npx_s, fabs(D[t] * px)); nq_s = fmax(nq_s, fabs(D[t] * qh));
      }
      pres = quad_max(pres); nz = quad_max(nz); nax = quad_max(nax);
      dres = quad_max(dres); naty = quad_max(naty); npx = quad_max(npx); nq = quad_max(nq);
      pri_res = pres;
      dua_res = cinv * dres;
      last_np = fmax(nz, nax);
      last_nd = cinv * fmax(fmax(naty, npx), nq);
      bool done = false;
      if (pri_res > kOsqpInfty || dua_res > kOsqpInfty) {
        status = kStatusNonCvx;
        done = true;
      } else {
        const bool pok = pri_res < eps_abs + eps_rel * last_np;
        const bool dok = dua_res < eps_abs + eps_rel * last_nd;
        bool pinf = false, dinf = false;
        if (!pok) {  // is_primal_infeasible: all bounds finite here, so delta_y is not projected
          double ndy = 0.0, lhs = 0.0;
#pragma unroll
          for (int c = 0; c < 5; c++) {
            ndy = fmax(ndy, fabs(E[c] * dy[c]));
            lhs += us[c] * fmax(dy[c], 0.0) + ls[c] * fmin(dy[c], 0.0);
          }
          ndy = quad_max(ndy);
          lhs = quad_sum(lhs);
          if (ndy > eps_prim_inf && lhs < -eps_prim_inf * ndy) {
            double at[3], na = 0.0, edy[5];  // Dinv A_s' dy = G' (E dy)
#pragma unroll
            for (int c = 0; c < 5; c++) edy[c] = E[c] * dy[c];
            cone_rows_t(edy, mu, at);
#pragma unroll
            for (int t = 0; t < 3; t++) na = fmax(na, fabs(at[t]));
            na = quad_max(na);
            pinf = na < eps_prim_inf * ndy;
          }
        }
        if (!dok) {  // is_dual_infeasible
          double ndx = 0.0, qdx = 0.0;
#pragma unroll
          for (int t = 0; t < 3; t++) {
            ndx = fmax(ndx, fabs(D[t] * dx[t]));
            qdx += (cs * D[t] * io.g[t]) * dx[t];
          }
          ndx = quad_max(ndx);
          qdx = quad_sum(qdx);
          if (ndx > eps_dual_inf && qdx < -cs * eps_dual_inf * ndx) {
            double dxs[3], dall[12], npd = 0.0;
#pragma unroll
            for (int t = 0; t < 3; t++) dxs[t] = D[t] * dx[t];
#pragma unroll
            for (int t = 0; t < 3; t++) {
              dall[t] = quad_bcast<0>(dxs[t]); dall[3 + t] = quad_bcast<1>(dxs[t]);
              dall[6 + t] = quad_bcast<2>(dxs[t]); dall[9 + t] = quad_bcast<3>(dxs[t]);
            }
#pragma unroll
            for (int t = 0; t < 3; t++) {
              double v = 0.0;
#pragma unroll
              for (int cb = 0; cb < 12; cb++) v += io.Hrow[t][cb] * dall[cb];
              npd = fmax(npd, fabs(cs * v));
            }
            npd = quad_max(npd);
            if (npd < cs * eps_dual_inf * ndx) {
              double adx[5];
              bool ok = true;
              cone_rows(dxs, mu, adx);  // Einv A_s dx = G (D dx)
#pragma unroll
              for (int c = 0; c < 5; c++)
                if (adx[c] > eps_dual_inf * ndx || adx[c] < -eps_dual_inf * ndx) ok = false;
              const unsigned long long bl = __ballot(!ok);
              const int base = (threadIdx.x & 63) & ~3;
              dinf = ((bl >> base) & 0xFull) == 0;
            }
          }
        }
        if (pok && dok) { status = kStatusSolved; done = true; }
        else if (pinf) { status = kStatusPrimalInf; done = true; }
        else if (dinf) { status = kStatusDualInf; done = true; }
      }
      if (done) break;
      if (iter % 200 == 0) {
        pres_s = quad_max(pres_s); nz_s = quad_max(nz_s); nax_s = quad_max(nax_s);
        dres_s = quad_max(dres_s); naty_s = quad_max(naty_s); npx_s = quad_max(npx_s); nq_s = quad_max(nq_s);
        const double pn = pres_s / (fmax(nz_s, nax_s) + 1e-10);
        const double dn = dres_s / (fmax(fmax(naty_s, npx_s), nq_s) + 1e-10);
        double rho_new = rho * sqrt(pn / (dn + 1e-10));
        rho_new = fmin(fmax(rho_new, kRhoMin), kRhoMax);
        if (rho_new > rho * 5.0 || rho_new < rho / 5.0) {
          rho = rho_new;
          need_factor = true;
        }
      }
    }
  }
  WPH(5);
#ifdef QRW_PROFILE_WBC
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_wbc_ph[blockIdx.x * 16 + 8] = (unsigned long long)iter;
#endif
  if (iter > max_iter) iter = max_iter;
  if (status == kStatusUnsolved) {
    const bool pok = pri_res < 10 * eps_abs + 10 * eps_rel * last_np;
    const bool dok = dua_res < 10 * eps_abs + 10 * eps_rel * last_nd;
    status = (pok && dok) ? kStatusSolvedInaccurate : kStatusMaxIter;
  }
  const bool has_sol = (status == kStatusSolved || status == kStatusSolvedInaccurate || status == kStatusMaxIter);
#pragma unroll
  for (int t = 0; t < 3; t++) sol[t] = has_sol ? D[t] * x[t] : nan("");
  if (!has_sol) {
#pragma unroll
    for (int t = 0; t < 3; t++) x[t] = 0.0;
#pragma unroll
    for (int c = 0; c < 5; c++) z[c] = y[c] = 0.0;
  }
  if (valid) {
#pragma unroll
    for (int t = 0; t < 3; t++) { st[kWsX + 3 * j + t] = x[t]; st[kWsG + 3 * j + t] = io.g[t]; }
#pragma unroll
    for (int c = 0; c < 5; c++) { st[kWsZ + 5 * j + c] = z[c]; st[kWsY + 5 * j + c] = y[c]; }
    if (j == 0) { st[kWsRho] = rho; st[kWsInit] = 1.0; }
  }
  iter_out = iter;
  status_out = status;
}

// QP data from per-foot A blocks (6x3), gamma and f_cmd (QPWBC::compute_matrices + update_PQ + bounds)
__device__ __forceinline__ void qp_build(const double Aj[6][3], const double gamma[6], const double fc[3], int j,
                                         QpIo& io) {
  const double mu = 0.9;
  double Aall[6][12];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int t = 0; t < 3; t++) {
      Aall[i][t] = quad_bcast<0>(Aj[i][t]); Aall[i][3 + t] = quad_bcast<1>(Aj[i][t]);
      Aall[i][6 + t] = quad_bcast<2>(Aj[i][t]); Aall[i][9 + t] = quad_bcast<3>(Aj[i][t]);
    }
#pragma unroll
  for (int t = 0; t < 3; t++) {
#pragma unroll
    for (int cb = 0; cb < 12; cb++) {
      double v = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) v += (Aj[i][t] * 0.1) * Aall[i][cb];
      if (cb / 3 == j && cb % 3 == t) v += 5.0;
      io.Hrow[t][cb] = v;
    }
    double gv = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) gv += (Aj[i][t] * 0.1) * gamma[i];
    io.g[t] = gv;
  }
  double gf[5];
  cone_rows(fc, mu, gf);
#pragma unroll
  for (int c = 0; c < 5; c++) { io.lo[c] = -gf[c]; io.up[c] = -gf[c] + 25.0; }
}

}  // namespace

__global__ __launch_bounds__(64, 1) void wbc_kernel(WbcArgs a) {
  const int lane = threadIdx.x;
  const int j = lane & 3;
  const int b = blockIdx.x * 16 + (lane >> 2);
  const bool valid = b < a.B;
  const int bb = valid ? b : 0;
  WPH(0);
  const LegC C = leg_consts(j);
  double* st = a.st + (size_t)bb * kWbcStItems;

  if (a.mode == 1) {  // fixed-base feet kinematics only (qrw_fixed_feet_host)
    double q[3], dq[3];
#pragma unroll
    for (int t = 0; t < 3; t++) { q[t] = a.in0[bb * 12 + 3 * j + t]; dq[t] = a.in1[bb * 12 + 3 * j + t]; }
    const LegKin K = leg_kinematics(C, q);
    const V3 vf = dq[0] * K.J0 + dq[1] * K.J1 + dq[2] * K.J2;
    const V3 w0 = dq[0] * K.a0, w1 = w0 + dq[1] * K.a1, w2 = w1 + dq[2] * K.a1;
    const V3 da1 = dq[0] * cross(K.a0, K.a1);
    const V3 vp1 = cross(w0, K.p1 - K.p0), vp2 = vp1 + cross(w1, K.p2 - K.p1);
    const V3 acl = dq[0] * cross(K.a0, vf) + dq[1] * (cross(da1, K.pf - K.p1) + cross(K.a1, vf - vp1)) +
                   dq[2] * (cross(da1, K.pf - K.p2) + cross(K.a1, vf - vp2));
    const V3 af = acl - cross(w2, vf);
    if (valid) {
      double* o;
      o = a.out0 + bb * 12 + 3 * j; o[0] = K.pf.x; o[1] = K.pf.y; o[2] = K.pf.z;
      o = a.out1 + bb * 12 + 3 * j; o[0] = vf.x; o[1] = vf.y; o[2] = vf.z;
      o = a.out2 + bb * 12 + 3 * j; o[0] = w2.x; o[1] = w2.y; o[2] = w2.z;
      o = a.out3 + bb * 12 + 3 * j; o[0] = af.x; o[1] = af.y; o[2] = af.z;
      double* J = a.out4 + bb * 144;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 12; c++) J[(3 * j + r) * 12 + c] = 0.0;
      J[(3 * j + 0) * 12 + 3 * j + 0] = K.J0.x; J[(3 * j + 1) * 12 + 3 * j + 0] = K.J0.y; J[(3 * j + 2) * 12 + 3 * j + 0] = K.J0.z;
      J[(3 * j + 0) * 12 + 3 * j + 1] = K.J1.x; J[(3 * j + 1) * 12 + 3 * j + 1] = K.J1.y; J[(3 * j + 2) * 12 + 3 * j + 1] = K.J1.z;
      J[(3 * j + 0) * 12 + 3 * j + 2] = K.J2.x; J[(3 * j + 1) * 12 + 3 * j + 2] = K.J2.y; J[(3 * j + 2) * 12 + 3 * j + 2] = K.J2.z;
    }
    return;
  }
  if (a.mode == 2) {  // InvKin::refreshAndCompute from caller-supplied kinematics (qrw_invkin_host)
    const double ct = a.in0[bb * 4 + j];
    V3 goal = mk(a.in1[bb * 12 + 0 * 4 + j], a.in1[bb * 12 + 1 * 4 + j], a.in1[bb * 12 + 2 * 4 + j]);
    V3 vg = mk(a.in2[bb * 12 + 0 * 4 + j], a.in2[bb * 12 + 1 * 4 + j], a.in2[bb * 12 + 2 * 4 + j]);
    V3 ag = mk(a.in3[bb * 12 + 0 * 4 + j], a.in3[bb * 12 + 1 * 4 + j], a.in3[bb * 12 + 2 * 4 + j]);
    const double* pp = a.in4 + bb * 12 + 3 * j; V3 pos = mk(pp[0], pp[1], pp[2]);
    pp = a.in5 + bb * 12 + 3 * j; V3 vf = mk(pp[0], pp[1], pp[2]);
    pp = a.in6 + bb * 12 + 3 * j; V3 wf = mk(pp[0], pp[1], pp[2]);
    pp = a.in7 + bb * 12 + 3 * j; V3 af = mk(pp[0], pp[1], pp[2]);
    const double* Jf = a.in8 + bb * 144 + (3 * j) * 12 + 3 * j;
    M3 iJ;
    inv3x3(mk(Jf[0], Jf[12], Jf[24]), mk(Jf[1], Jf[13], Jf[25]), mk(Jf[2], Jf[14], Jf[26]), iJ);
    const V3 e = goal - pos;
    V3 acc = 100.0 * e - (2.0 * sqrt(100.0)) * (vf - vg) + ag;
    if (ct != 0.0) acc = 0.0 * acc;
    acc = acc - (af + cross(wf, vf));
    const V3 ddq = mul(iJ, acc), dqc = mul(iJ, vg), qs = mul(iJ, e);
    if (valid) {
      double* o;
      o = a.out0 + bb * 12 + 3 * j; o[0] = ddq.x; o[1] = ddq.y; o[2] = ddq.z;
      o = a.out1 + bb * 12 + 3 * j; o[0] = dqc.x; o[1] = dqc.y; o[2] = dqc.z;
      o = a.out2 + bb * 12 + 3 * j; o[0] = qs.x; o[1] = qs.y; o[2] = qs.z;
    }
    return;
  }
  if (a.mode == 3) {  // QPWBC::run from caller-supplied M (diagonal), Jc, f_cmd, RNEA (qrw_qpwbc_host)
    const double* M = a.in0 + (size_t)bb * 324;
    const double* Jc = a.in1 + (size_t)bb * 216;
    double Aj[6][3], fc[3], xf[6], gamma[6];
#pragma unroll
    for (int t = 0; t < 3; t++) fc[t] = a.in2[bb * 12 + 3 * j + t];
    if (a.in4) {
      // general Y: Yinv = pseudoInverse(M[:6,:6]) (include/qrw/InvKin.hpp:60-66) from pinv6_kernel; A = Yinv X, gamma = Yinv (X f - RNEA)
      const double* Yv = a.in4 + (size_t)bb * 36;
      double X[6][3], d6[6];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        double s_ = 0.0;
#pragma unroll
        for (int t = 0; t < 3; t++) { X[i][t] = Jc[(3 * j + t) * 18 + i]; s_ += X[i][t] * fc[t]; }
        xf[i] = quad_sum(s_);
        d6[i] = xf[i] - a.in3[bb * 6 + i];
      }
#pragma unroll
      for (int i = 0; i < 6; i++) {
        double g_ = 0.0;
#pragma unroll
        for (int t = 0; t < 3; t++) Aj[i][t] = 0.0;
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const double y = Yv[i * 6 + m];
          g_ += y * d6[m];
#pragma unroll
          for (int t = 0; t < 3; t++) Aj[i][t] += y * X[m][t];
        }
        gamma[i] = g_;
      }
    } else {
      double Yd[6], Yi[6], smax = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) { Yd[i] = M[i * 18 + i]; smax = fmax(smax, fabs(Yd[i])); }
      const double tol = 2.220446049250313e-16 * 6.0 * smax;  // pseudoInverse<>, InvKin.hpp:60-66, of a diagonal block
#pragma unroll
      for (int i = 0; i < 6; i++) Yi[i] = (fabs(Yd[i]) > tol) ? 1.0 / Yd[i] : 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        double s_ = 0.0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const double xv = Jc[(3 * j + t) * 18 + i];
          Aj[i][t] = Yi[i] * xv;
          s_ += xv * fc[t];
        }
        xf[i] = quad_sum(s_);
        gamma[i] = Yi[i] * (xf[i] - a.in3[bb * 6 + i]);
      }
    }
    QpIo io;
    qp_build(Aj, gamma, fc, j, io);
    double sol[3];
    int it, stt;
    qp_solve(io, st, j, valid, sol, it, stt);
    double dd[6];
#pragma unroll
    for (int i = 0; i < 6; i++) dd[i] = quad_sum(Aj[i][0] * sol[0] + Aj[i][1] * sol[1] + Aj[i][2] * sol[2]) + gamma[i];
    if (valid) {
#pragma unroll
      for (int t = 0; t < 3; t++) a.out0[bb * 12 + 3 * j + t] = sol[t] + fc[t];
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) a.out1[bb * 6 + i] = dd[i];
        a.iters[bb] = it;
        a.status[bb] = stt;
      }
      if (a.out2) {
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
          for (int cb = 0; cb < 12; cb++) a.out2[(size_t)bb * 144 + (3 * j + t) * 12 + cb] = io.Hrow[t][cb];
      }
    }
    return;
  }

  // ================================ full wbc_controller.compute ================================
  const double* qv = a.q + (size_t)bb * 19;
  const double* dqv = a.dq + (size_t)bb * 18;
  double q[3], dq[3], fc[3];
#pragma unroll
  for (int t = 0; t < 3; t++) { q[t] = qv[7 + 3 * j + t]; dq[t] = dqv[6 + 3 * j + t]; fc[t] = a.f_cmd[bb * 12 + 3 * j + t]; }
  const double contact = a.contacts[bb * 4 + j];
  const bool stance = (contact != 0.0);
  {  // k_since_contact (QP_WBC.py:65-66)
    double ks = st[kWsKsc + j];
    ks += contact;
    ks *= contact;
    if (valid) st[kWsKsc + j] = ks;
  }
  WPH(1);
  const LegKin K = leg_kinematics(C, q);
  // ---- fixed-base foot velocity and classical acceleration with zero joint acceleration
  const V3 vf = dq[0] * K.J0 + dq[1] * K.J1 + dq[2] * K.J2;
  V3 acl;
  {
    const V3 w0 = dq[0] * K.a0, w1 = w0 + dq[1] * K.a1;
    const V3 da1 = dq[0] * cross(K.a0, K.a1);
    const V3 vp1 = cross(w0, K.p1 - K.p0), vp2 = vp1 + cross(w1, K.p2 - K.p1);
    acl = dq[0] * cross(K.a0, vf) + dq[1] * (cross(da1, K.pf - K.p1) + cross(K.a1, vf - vp1)) +
          dq[2] * (cross(da1, K.pf - K.p2) + cross(K.a1, vf - vp2));
  }
  // ---- InvKin (src/InvKin.cpp:36-62); af + w x v of the reference IS the classical acceleration
  const V3 goal = mk(a.pgoals[bb * 12 + 0 * 4 + j], a.pgoals[bb * 12 + 1 * 4 + j], a.pgoals[bb * 12 + 2 * 4 + j]);
  const V3 vgoal = mk(a.vgoals[bb * 12 + 0 * 4 + j], a.vgoals[bb * 12 + 1 * 4 + j], a.vgoals[bb * 12 + 2 * 4 + j]);
  const V3 agoal = mk(a.agoals[bb * 12 + 0 * 4 + j], a.agoals[bb * 12 + 1 * 4 + j], a.agoals[bb * 12 + 2 * 4 + j]);
  M3 iJ;
  inv3x3(K.J0, K.J1, K.J2, iJ);
  const V3 perr = goal - K.pf;
  V3 afeet = 100.0 * perr - (2.0 * sqrt(100.0)) * (vf - vgoal) + agoal;
  if (stance) afeet = 0.0 * afeet;
  afeet = afeet - acl;
  const V3 ddq3 = mul(iJ, afeet), dqc3 = mul(iJ, vgoal), qs3 = mul(iJ, perr);
  const double ddq[3] = {ddq3.x, ddq3.y, ddq3.z};

  // ---- free-flyer quantities
  M3 Rb;
  {
    const double x = qv[3], y = qv[4], z = qv[5], w = qv[6];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                 tyz = tz * y, tzz = tz * z;
    Rb.r0 = mk(1 - (tyy + tzz), txy - twz, txz + twy);
    Rb.r1 = mk(txy + twz, 1 - (txx + tzz), tyz - twx);
    Rb.r2 = mk(txz - twy, tyz + twx, 1 - (txx + tyy));
  }
  const V3 vb = mk(dqv[0], dqv[1], dqv[2]), wb = mk(dqv[3], dqv[4], dqv[5]);
  const V3 grav = mulT(Rb, mk(0.0, 0.0, QRW_SOLO12_MODEL.gravity));  // -gravity in base coordinates
  const V3 ab0 = grav + cross(wb, vb);                               // classical base accel, ddq_base = 0
  const qrw_link_inertial& BL = QRW_SOLO12_MODEL.base;
  const double Ib[6] = {BL.inertia[0], BL.inertia[1], BL.inertia[2], BL.inertia[3], BL.inertia[4], BL.inertia[5]};
  const V3 cb = mk(BL.com[0], BL.com[1], BL.com[2]);
  M3 Id;
  Id.r0 = mk(1, 0, 0); Id.r1 = mk(0, 1, 0); Id.r2 = mk(0, 0, 1);
  // first rnea: base acceleration 0, joint acceleration ddq_cmd (QP_WBC.py:104); only [:6] is used
  V3 Fl, Ml;
  double tau1[3];
  leg_newton_euler(C, K, dq, ddq, wb, mk(0, 0, 0), ab0, Fl, Ml, tau1);
  V3 Fb = BL.mass * (ab0 + cross(wb, cross(wb, cb)));
  V3 Mb = inertia_apply(Ib, Id, mk(0, 0, 0)) + cross(wb, inertia_apply(Ib, Id, wb)) + cross(cb, Fb);
  const V3 F6 = quad_sum3(Fl) + Fb, M6 = quad_sum3(Ml) + Mb;
  const double rnea6[6] = {F6.x, F6.y, F6.z, M6.x, M6.y, M6.z};
  // ---- QP data: X = Jc[:, :6]', A = Yinv X (QPWBC.cpp:491-497)
  double Aj[6][3], gamma[6];
  {
    // rows of Jc for this foot: [Rb, -Rb skew(r)] -> X block = [Rb'; skew(r) Rb'] (6x3), zero in swing
    const V3 r = K.pf;
    const V3 c0 = mk(Rb.r0.x, Rb.r1.x, Rb.r2.x), c1 = mk(Rb.r0.y, Rb.r1.y, Rb.r2.y), c2 = mk(Rb.r0.z, Rb.r1.z, Rb.r2.z);
    // Rb' has rows c0,c1,c2 (columns of Rb); column t of Rb' is row t of Rb
    const V3 rt[3] = {Rb.r0, Rb.r1, Rb.r2};
    double X[6][3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const V3 col = mk(rt[t].x, rt[t].y, rt[t].z);  // Rb'(:,t) = (Rb(t,0), Rb(t,1), Rb(t,2))
      const V3 sc = cross(r, col);
      X[0][t] = col.x; X[1][t] = col.y; X[2][t] = col.z;
      X[3][t] = sc.x; X[4][t] = sc.y; X[5][t] = sc.z;
    }
    (void)c0; (void)c1; (void)c2;
    double xf[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double s_ = 0.0;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const double xv = stance ? X[i][t] : 0.0;
        Aj[i][t] = (1.0 / a.Y[i]) * xv;
        s_ += xv * fc[t];
      }
      xf[i] = quad_sum(s_);
      gamma[i] = (1.0 / a.Y[i]) * (xf[i] - rnea6[i]);
    }
  }
  // everything that does not depend on the QP leaves now, and the leg kinematics / base rotation are recomputed after
  // the solve instead of being carried across it: the ADMM loop needs most of the register file for itself
  if (valid) {
    if (a.qdes) {
      double* o = a.qdes + bb * 19;
      if (j == 0) for (int i = 0; i < 7; i++) o[i] = 0.0;  // q_cmd[:7] is never written (solo12InvKin.py:67)
      o[7 + 3 * j] = q[0] + qs3.x; o[8 + 3 * j] = q[1] + qs3.y; o[9 + 3 * j] = q[2] + qs3.z;
    }
    if (a.vdes) {
      double* o = a.vdes + bb * 18;
      if (j == 0) for (int i = 0; i < 6; i++) o[i] = 0.0;
      o[6 + 3 * j] = dqc3.x; o[7 + 3 * j] = dqc3.y; o[8 + 3 * j] = dqc3.z;
    }
    if (a.feet) {  // feet_pos, feet_err, feet_vel as 3x4 each (QP_WBC.py:73-80)
      double* o = a.feet + (size_t)bb * 36;
      o[0 * 4 + j] = K.pf.x; o[1 * 4 + j] = K.pf.y; o[2 * 4 + j] = K.pf.z;
      o[12 + 0 * 4 + j] = perr.x; o[12 + 1 * 4 + j] = perr.y; o[12 + 2 * 4 + j] = perr.z;
      o[24 + 0 * 4 + j] = vf.x; o[24 + 1 * 4 + j] = vf.y; o[24 + 2 * 4 + j] = vf.z;
    }
  }
  // The 27 values the epilogue needs again wait in LDS while the solve runs (lane-strided, conflict-free): carried in
  // registers they were what the compiler spilled to scratch (44 dwords per lane), and the ADMM loop wants the file.
  __shared__ double park[27 * 64];
  {
    double* pk = park + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      pk[(0 + t) * 64] = fc[t]; pk[(3 + t) * 64] = dq[t]; pk[(6 + t) * 64] = ddq[t];
      pk[(9 + t) * 64] = (t == 0) ? q[0] + qs3.x : (t == 1) ? q[1] + qs3.y : q[2] + qs3.z;   // q_des of this leg
      pk[(12 + t) * 64] = (t == 0) ? dqc3.x : (t == 1) ? dqc3.y : dqc3.z;                      // v_des of this leg
    }
    pk[15 * 64] = wb.x; pk[16 * 64] = wb.y; pk[17 * 64] = wb.z;
    pk[18 * 64] = vb.x; pk[19 * 64] = vb.y; pk[20 * 64] = vb.z;
#pragma unroll
    for (int i = 0; i < 6; i++) pk[(21 + i) * 64] = gamma[i];
  }
  WPH(2);
  QpIo io;
  qp_build(Aj, gamma, fc, j, io);
  double sol[3];
  int it, stt;
  // re-read after the solve (not carried across it): joint angles and base quaternion for the second evaluation of the
  // leg kinematics, and the operands of the fused result / security check
  qp_solve(io, st, j, valid, sol, it, stt);
  WPH(6);
  double q2[3], qq[4], c_err0 = 0.0, c_qf[3] = {0.0, 0.0, 0.0}, c_vs[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 3; t++) q2[t] = qv[7 + 3 * j + t];
#pragma unroll
  for (int t = 0; t < 4; t++) qq[t] = qv[3 + t];
  if (a.c_cs) {  // requested now, so that they do not cost a round trip of their own behind the torque stores
    c_err0 = a.c_cs[bb + (size_t)glue::cERR * (size_t)a.B];
#pragma unroll
    for (int t = 0; t < 3; t++) { c_qf[t] = a.c_qfilt[bb * 19 + 7 + 3 * j + t]; c_vs[t] = a.c_vsecu[bb * 12 + 3 * j + t]; }
  }
  double fcp[3], dqp[3], ddqp[3], qd_leg[3], vd_leg[3], gammap[6];
  V3 wbp, vbp;
  {
    int off = threadIdx.x;
    asm volatile("" : "+v"(off));  // opaque address: the values are read back, not kept alive across the solve
    const double* pk = park + off;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      fcp[t] = pk[(0 + t) * 64]; dqp[t] = pk[(3 + t) * 64]; ddqp[t] = pk[(6 + t) * 64];
      qd_leg[t] = pk[(9 + t) * 64]; vd_leg[t] = pk[(12 + t) * 64];
    }
    wbp = mk(pk[15 * 64], pk[16 * 64], pk[17 * 64]);
    vbp = mk(pk[18 * 64], pk[19 * 64], pk[20 * 64]);
#pragma unroll
    for (int i = 0; i < 6; i++) gammap[i] = pk[(21 + i) * 64];
  }
  double dd[6];
  const double fw[3] = {sol[0] + fcp[0], sol[1] + fcp[1], sol[2] + fcp[2]};
  double tau2[3], tff[3];
  {
    // recomputed (not carried across the solve): the opaque asm keeps the compiler from merging this evaluation with
    // the first one
    asm volatile("" : "+v"(q2[0]), "+v"(q2[1]), "+v"(q2[2]), "+v"(qq[0]), "+v"(qq[1]), "+v"(qq[2]), "+v"(qq[3]));
    const LegKin K2 = leg_kinematics(C, q2);
    M3 Rb2;
    {
      const double x = qq[0], y = qq[1], z = qq[2], w = qq[3];
      const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
      const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                   tyz = tz * y, tzz = tz * z;
      Rb2.r0 = mk(1 - (tyy + tzz), txy - twz, txz + twy);
      Rb2.r1 = mk(txy + twz, 1 - (txx + tzz), tyz - twx);
      Rb2.r2 = mk(txz - twy, tyz + twx, 1 - (txx + tyy));
    }
    // A = Yinv X again (same arithmetic as above), then ddq_res = A f_res + gamma (QPWBC.cpp:285-289)
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double acc = 0.0;
      const V3 rt[3] = {Rb2.r0, Rb2.r1, Rb2.r2};
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const V3 col = mk(rt[t].x, rt[t].y, rt[t].z);
        const V3 sc = cross(K2.pf, col);
        const double xi = (i == 0) ? col.x : (i == 1) ? col.y : (i == 2) ? col.z : (i == 3) ? sc.x : (i == 4) ? sc.y : sc.z;
        const double xv = stance ? xi : 0.0;
        const double Ait = (1.0 / a.Y[i]) * xv;
        acc = (t == 0) ? Ait * sol[0] : acc + Ait * sol[t];
      }
      dd[i] = quad_sum(acc) + gammap[i];
    }
    const V3 alb = mk(dd[3], dd[4], dd[5]);
    const V3 grav2 = mulT(Rb2, mk(0.0, 0.0, QRW_SOLO12_MODEL.gravity));
    const V3 ab02 = grav2 + cross(wbp, vbp);
    const V3 ab1 = ab02 + mk(dd[0], dd[1], dd[2]);
    leg_newton_euler(C, K2, dqp, ddqp, wbp, alb, ab1, Fl, Ml, tau2);
    // tau_ff = RNEA_delta - Jc[:, 6:]' f (QP_WBC.py:117): Jc joint block of a stance foot = Rb J_leg
    const V3 fbv = mulT(Rb2, mk(fw[0], fw[1], fw[2]));  // Rb' f
    tff[0] = tau2[0] - (stance ? dot(K2.J0, fbv) : 0.0);
    tff[1] = tau2[1] - (stance ? dot(K2.J1, fbv) : 0.0);
    tff[2] = tau2[2] - (stance ? dot(K2.J2, fbv) : 0.0);
  }
  if (valid) {
    if (a.tau_ff) { double* o = a.tau_ff + bb * 12 + 3 * j; o[0] = tff[0]; o[1] = tff[1]; o[2] = tff[2]; }
    if (a.f_with_delta) { double* o = a.f_with_delta + bb * 12 + 3 * j; o[0] = fw[0]; o[1] = fw[1]; o[2] = fw[2]; }
    if (a.ddq_res && j == 0)
      for (int i = 0; i < 6; i++) a.ddq_res[bb * 6 + i] = dd[i];
    if (j == 0) { a.iters[bb] = it; a.status[bb] = stt; }
    if (a.c_cs) {
      // fused tail of the control iteration: Controller result + security_check (scripts/Controller.py:306-310,
      // 341-365; same arithmetic as glue::result, one leg per lane, flags combined over the quad)
      double* cs = a.c_cs + bb;
      const size_t cB = (size_t)a.B;
      int err = (int)c_err0;
      const double qd[3] = {qd_leg[0], qd_leg[1], qd_leg[2]};
      const double vd[3] = {vd_leg[0], vd_leg[1], vd_leg[2]};
      const double qsec[3] = {M_PI * 0.4, M_PI * 80 / 180, M_PI};
      double e1 = 0.0, e2 = 0.0, e3 = 0.0, e4 = 0.0;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        if (fabs(c_qf[t]) > qsec[t]) e1 = 1.0;
        if (fabs(c_vs[t]) > 50) e2 = 1.0;
        if (fabs(tff[t]) > 8) e3 = 1.0;
        // not in the reference (its comparisons are blind to NaN): a non-finite command stops the robot, code 4 (glue::result)
        if (!(fabs(tff[t]) <= 1.7976931348623157e308) || !(fabs(qd[t]) <= 1.7976931348623157e308) ||
            !(fabs(vd[t]) <= 1.7976931348623157e308)) e4 = 1.0;
      }
      e1 = quad_max(e1); e2 = quad_max(e2); e3 = quad_max(e3); e4 = quad_max(e4);
      if (err == 0) {  // the WBC counts this iteration: keep its references for the next one
#pragma unroll
        for (int t = 0; t < 3; t++) {
          cs[(size_t)(glue::cQDES + 3 * j + t) * cB] = qd[t];
          cs[(size_t)(glue::cVDES + 3 * j + t) * cB] = vd[t];
        }
        if (e4 != 0.0) err = 4;
        if (e1 != 0.0) err = 1;
        if (e2 != 0.0) err = 2;
        if (e3 != 0.0) err = 3;
        if (j == 0) cs[(size_t)glue::cERR * cB] = (double)err;
      }
      double* r = a.c_result + (size_t)bb * 60;  // P | D | q_des | v_des | tau_ff
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int i = 3 * j + t;
        if (err == 0) {
          r[i] = 3.0; r[12 + i] = 0.2; r[24 + i] = qd[t]; r[36 + i] = vd[t]; r[48 + i] = 0.8 * tff[t];
        } else {
          r[i] = 0.0; r[12 + i] = 0.1; r[24 + i] = 0.0; r[36 + i] = 0.0; r[48 + i] = 0.0;
        }
      }
      if (j == 0 && a.c_err) a.c_err[bb] = err;
    }
  }
  WPH(7);
}

// =====================================================================================================================
// wbc16_kernel (round 4): the same wbc_controller.compute with SIXTEEN LANES PER ROBOT INSTANCE -- one 16-lane DPP row, four
// instances per wavefront, B / 4 wavefronts (1024 at batch 4096: every SIMD of the chip; wbc_kernel's one quad per instance
// fills a quarter of them and is bound by the issue rate of ONE wavefront's 18 k instructions, profiles/r4_wbc_phase_cycles.txt).
//   lane of the row = 4 * foot + t.  t = 0, 1, 2: force variable x, y, z of that foot (12 QP variables on 12 lanes: the lane holds
//   ITS row of H and of the KKT inverse, its x, D, g); t = 3: a pad lane.  The friction-cone rows of a foot sit on the lanes of
//   its quad: row c = t on lane t ("slot A", c = 0..3), the fifth row (f_z <= 25) on the pad lane as well ("slot B"; every lane
//   carries a slot B, only the pad lane's counts).
//   * per-foot phases (leg kinematics, InvKin, Newton-Euler, QP data, epilogue): every lane of a quad runs its foot's arithmetic
//     (the same instructions as wbc_kernel's one lane per foot; redundant, but no slower); sums over the feet are sums over the
//     four quads of the row (row_xor4 / row_xor8 below, same association as quad_sum).
//   * QP phases: a matrix-vector product with the 12 x 12 inverse is twelve v_fmac_f64_dpp with row_newbcast (the broadcast costs
//     no instruction; wbc_kernel: 36 FMAs + 24 DPP moves per lane); the inverse itself is an in-register Gauss-Jordan across the
//     row (as gj_invert12 of chain_sweep.h, for the padded lane numbering); the ten equilibration passes take a third of the
//     instructions (12 instead of 36 norm entries per lane, one or two instead of five row scalings).
//   Arithmetic: the per-foot phases, the QP data, the equilibration and the ADMM iteration evaluate wbc_kernel's expressions in
//   wbc_kernel's order; the KKT inverse is eliminated with m[c] += m_P[c] * f instead of Ki - (f m_P[c]) d and a fast reciprocal:
//   results agree with wbc_kernel to rounding (tests/test_gpu_wbc.py::test_wbc16_matches_the_quad_kernel), with the oracle as
//   before.  qrw_wbc_set_lanes(h, 4) / QRW_WBC16=0 select wbc_kernel for the full compute (the stand-alone modes always use it).
namespace {

template <int CTRL, int RM, int BM>
__device__ __forceinline__ double dpp_d(double old, double v) {  // lanes the masks leave out keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, RM, BM, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, RM, BM, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int BM>
__device__ __forceinline__ double dpp_all(double v) {  // every written lane has a source lane: no tied old value, no copy first
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, BM, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, BM, true);
  return __hiloint2double(hi, lo);
}
// value of lane LANE of this lane's 16-lane row (row_newbcast; scripts/ubench/dpp_row_probe.hip checks these controls)
template <int LANE>
__device__ __forceinline__ double row_bcast(double v) { return dpp_all<0x150 + LANE, 0xF>(v); }
// value of lane ^ 4 (row_shl:4 on banks 0, 2 + row_shr:4 on banks 1, 3; a bank = a quad of the row) / of lane ^ 8 (row_ror:8)
__device__ __forceinline__ double row_xor4(double v) { const double t = dpp_all<0x104, 0x5>(v); return dpp_d<0x114, 0xF, 0xA>(t, v); }
__device__ __forceinline__ double row_xor8(double v) { return dpp_all<0x128, 0xF>(v); }
// over the four feet (quads) of a row, for a value that is uniform inside each quad; (q + q^1) + (q^2 + q^3) like quad_sum
__device__ __forceinline__ double feet_sum(double v) { v += row_xor4(v); v += row_xor8(v); return v; }
__device__ __forceinline__ double feet_max(double v) { v = fmax(v, row_xor4(v)); v = fmax(v, row_xor8(v)); return v; }
__device__ __forceinline__ V3 feet_sum3(V3 v) { return mk(feet_sum(v.x), feet_sum(v.y), feet_sum(v.z)); }
__device__ __forceinline__ double row_max(double v) { return feet_max(quad_max(v)); }  // over all 16 lanes

// lane of the row that holds QP variable cb (0..11)
#define QRW_VL(cb) (4 * ((cb) / 3) + (cb) % 3)
// r + sum_cb m[cb] * x(lane of variable cb): the 12 x 12 row-times-vector product, broadcast folded into the FMAs
__device__ __forceinline__ double row_matvec12(double r, double x, const double (&m)[12]) {
  double a0 = r;
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %10 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %11 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %12 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %13 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(a0)
      : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]),
        "v"(m[9]), "v"(m[10]), "v"(m[11]));
  return a0;
}
// sum_i A_i(any lane of quad Q) * coef_i, i = 0..5 in order: one entry of H = 0.1 A'A (QPWBC::compute_matrices)
template <int Q>
__device__ __forceinline__ double h_entry(const double (&A)[6], const double (&coef)[6]) {
  double acc = 0.0;
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %2, %8 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %3, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %4, %10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %5, %11 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %6, %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(acc)
      : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(coef[0]), "v"(coef[1]), "v"(coef[2]), "v"(coef[3]),
        "v"(coef[4]), "v"(coef[5]), "n"(4 * Q));
  return acc;
}
// one pivot of the in-register Gauss-Jordan over the row: lane of variable v holds row v of the matrix in m[0..11]
#define QRW_GJ16(C) "v_fmac_f64_dpp %" #C ", %" #C ", %12 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
template <int P>
__device__ __forceinline__ void gj16_pivot(double (&m)[12], int v) {
  double piv = 0.0;
  const double one = 1.0;
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(piv) : "v"(m[P]), "v"(one), "n"(QRW_VL(P)));
  const double d = fast_rcp(piv);
  const double f = (v == P) ? (d - 1.0) : (-m[P] * d);
  asm("s_nop 1\n\t" QRW_GJ16(0) QRW_GJ16(1) QRW_GJ16(2) QRW_GJ16(3) QRW_GJ16(4) QRW_GJ16(5) QRW_GJ16(6) QRW_GJ16(7) QRW_GJ16(8)
      QRW_GJ16(9) QRW_GJ16(10) QRW_GJ16(11)
      : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]),
        "+v"(m[10]), "+v"(m[11])
      : "v"(f), "n"(QRW_VL(P)));
  m[P] = (v == P) ? d : f;
}
#undef QRW_GJ16
__device__ __forceinline__ void gj16_invert(double (&m)[12], int v) {
  gj16_pivot<0>(m, v); gj16_pivot<1>(m, v); gj16_pivot<2>(m, v); gj16_pivot<3>(m, v); gj16_pivot<4>(m, v); gj16_pivot<5>(m, v);
  gj16_pivot<6>(m, v); gj16_pivot<7>(m, v); gj16_pivot<8>(m, v); gj16_pivot<9>(m, v); gj16_pivot<10>(m, v); gj16_pivot<11>(m, v);
}
// the x, y, z values of this lane's foot (from lanes 0, 1, 2 of its quad)
__device__ __forceinline__ void foot_xyz(double v, double (&o)[3]) { o[0] = quad_bcast<0>(v); o[1] = quad_bcast<1>(v); o[2] = quad_bcast<2>(v); }
// the five cone-row values of this lane's foot: slot A of lanes 0..3, slot B of lane 3
__device__ __forceinline__ void foot_rows(double vA, double vB, double (&o)[5]) {
  o[0] = quad_bcast<0>(vA); o[1] = quad_bcast<1>(vA); o[2] = quad_bcast<2>(vA); o[3] = quad_bcast<3>(vA); o[4] = quad_bcast<3>(vB);
}

// QP of one instance on one 16-lane row.  H: this lane's row of H; g: its entry of the linear cost; lo / up: bounds of its two row
// slots.  st: the instance's persistent state.  sol: D x of this lane's variable.
__device__ void qp_solve16(const double (&H)[12], double g, const double (&lo)[2], const double (&up)[2], double* st, int j, int t,
                           bool valid, double& sol, int& iter_out, int& status_out) {
  const double mu = 0.9;
  const double sigma = 1e-6, alpha = 1.6;
  const double eps_abs = (double)(float)1e-5, eps_rel = (double)(float)1e-5;  // QPWBC.cpp:239-240
  const double eps_prim_inf = 1e-4, eps_dual_inf = 1e-4;
  const bool isvar = t < 3, isB = (t == 3);
  const int v = isvar ? 3 * j + t : -1;
  const int vs = isvar ? 3 * j + t : 3 * j;  // (pad lanes read foot j's x entry: never used)
  WPH(3);
  const double init_flag = st[kWsInit];
  double x = st[kWsX + vs], gprev = st[kWsG + vs], rho = st[kWsRho];
  double z[2] = {st[kWsZ + 5 * j + t], st[kWsZ + 5 * j + 4]}, y[2] = {st[kWsY + 5 * j + t], st[kWsY + 5 * j + 4]};
  const bool first = !valid || !(init_flag != 0.0);
  if (first) {
    rho = 0.1;
    x = 0.0; gprev = g;
    z[0] = z[1] = y[0] = y[1] = 0.0;
  }
  if (!isvar) { x = 0.0; gprev = 0.0; }
  // Gabs of this lane's variable against the five rows of its foot, and of its two row slots against x, y, z (QPWBC.cpp:10-22)
  const double gv[5] = {(t == 0) ? 1.0 : (t == 2) ? mu : 0.0, (t == 0) ? 1.0 : (t == 2) ? mu : 0.0, (t == 1) ? 1.0 : (t == 2) ? mu : 0.0,
                        (t == 1) ? 1.0 : (t == 2) ? mu : 0.0, (t == 2) ? 1.0 : 0.0};
  const double gA[3] = {(t < 2) ? 1.0 : 0.0, (t >= 2) ? 1.0 : 0.0, mu};  // slot A: row t
  // ---- scale_data (ten passes), wbc_kernel's expressions
  // The inf-norm of this lane's row of c D H D is (c D_v) * max_cb(|H[cb]| D_cb): hm = that maximum is taken ONCE per pass, after
  // the update of D -- the pass's cost normalisation and the next pass's column norm both read it (wbc_kernel evaluates the twelve
  // triple products twice per pass; same values up to the rounding of the last product, a preconditioner like the rsqrt below)
  double D = 1.0, E[2] = {1.0, 1.0}, cs = 1.0;
  double Habs[12], hm = 0.0;
#pragma unroll
  for (int cb = 0; cb < 12; cb++) { Habs[cb] = fabs(H[cb]); hm = fmax(hm, Habs[cb]); }
  for (int pass = 0; pass < 10; pass++) {
    double Ef[5], Df[3];
    foot_rows(E[0], E[1], Ef);
    foot_xyz(D, Df);
    double nD = (cs * D) * hm;
#pragma unroll
    for (int c = 0; c < 5; c++) nD = fmax(nD, Ef[c] * gv[c] * D);
    double nE[2] = {0.0, 0.0};
#pragma unroll
    for (int tt = 0; tt < 3; tt++) nE[0] = fmax(nE[0], E[0] * gA[tt] * Df[tt]);
    nE[1] = fmax(nE[1], E[1] * 1.0 * Df[2]);  // row 4: (0, 0, 1)
    D *= rsqrt(limit_scaling(nD));
    E[0] *= rsqrt(limit_scaling(nE[0]));
    E[1] *= rsqrt(limit_scaling(nE[1]));
    {
      double Dall[12];
      Dall[0] = row_bcast<QRW_VL(0)>(D); Dall[1] = row_bcast<QRW_VL(1)>(D); Dall[2] = row_bcast<QRW_VL(2)>(D);
      Dall[3] = row_bcast<QRW_VL(3)>(D); Dall[4] = row_bcast<QRW_VL(4)>(D); Dall[5] = row_bcast<QRW_VL(5)>(D);
      Dall[6] = row_bcast<QRW_VL(6)>(D); Dall[7] = row_bcast<QRW_VL(7)>(D); Dall[8] = row_bcast<QRW_VL(8)>(D);
      Dall[9] = row_bcast<QRW_VL(9)>(D); Dall[10] = row_bcast<QRW_VL(10)>(D); Dall[11] = row_bcast<QRW_VL(11)>(D);
      hm = 0.0;
#pragma unroll
      for (int cb = 0; cb < 12; cb++) hm = fmax(hm, Habs[cb] * Dall[cb]);
    }
    const double vcol = (cs * D) * hm;
    double vf3[3];
    foot_xyz(vcol, vf3);
    const double colsum = (vf3[0] + vf3[1]) + vf3[2];  // the foot's three columns, summed as wbc_kernel's lane does
    double qn = isvar ? fabs(cs * D * gprev) : 0.0;
    double ct = feet_sum(colsum) * (1.0 / 12.0);
    qn = limit_scaling(row_max(qn));
    ct = fmax(ct, qn);
    ct = limit_scaling(ct);
    cs *= 1.0 / ct;
  }
  WPH(4);
  const double cinv = 1.0 / cs;
  const double iD = 1.0 / D;
  const double iE[2] = {1.0 / E[0], 1.0 / E[1]};
  const double ls[2] = {E[0] * lo[0], E[1] * lo[1]}, us[2] = {E[0] * up[0], E[1] * up[1]};
  rho = fmin(fmax(rho, kRhoMin), kRhoMax);

  double Ki[12];
  bool need_factor = true;
  int iter, status = kStatusUnsolved;
  double last_np = 0.0, last_nd = 0.0, pri_res = 0.0, dua_res = 0.0;
  const int max_iter = 4000;
  for (iter = 1; valid && iter <= max_iter; iter++) {
    if (need_factor) {  // Khat = c H + sigma D^-2 + rho G' E^2 G; this lane's row; inverse by Gauss-Jordan across the row
      need_factor = false;
      double Ef[5], om[5];
      foot_rows(E[0], E[1], Ef);
#pragma unroll
      for (int c = 0; c < 5; c++) om[c] = rho * Ef[c] * Ef[c];
      const double s4 = om[0] + om[1] + om[2] + om[3];
      const double cone[3][3] = {{om[0] + om[1], 0.0, mu * (om[1] - om[0])},
                                 {0.0, om[2] + om[3], mu * (om[3] - om[2])},
                                 {mu * (om[1] - om[0]), mu * (om[3] - om[2]), mu * mu * s4 + om[4]}};
      const int tc = isvar ? t : 0;
      double crow[3];  // this lane's row of its foot's 3 x 3 block, sigma / D^2 on the diagonal
#pragma unroll
      for (int e = 0; e < 3; e++) {
        crow[e] = (tc == 0) ? cone[0][e] : (tc == 1) ? cone[1][e] : cone[2][e];
        if (e == tc) crow[e] += sigma * iD * iD;
      }
#pragma unroll
      for (int cb = 0; cb < 12; cb++) {
        const double kv = cs * H[cb] + ((cb / 3 == j) ? crow[cb % 3] : 0.0);
        Ki[cb] = isvar ? kv : 0.0;  // (pad lanes: a zero row, which the elimination leaves zero)
      }
      gj16_invert(Ki, v);
    }
    const double rho_inv = 1.0 / rho;
    // rhs (hatted): sigma x / D - c g + G' E (rho z - y)
    const double wA = E[0] * (rho * (z[0] - rho_inv * y[0])), wB = E[1] * (rho * (z[1] - rho_inv * y[1]));
    double w[5], gt3[3];
    foot_rows(wA, wB, w);
    cone_rows_t(w, mu, gt3);
    const double gt = (t == 0) ? gt3[0] : (t == 1) ? gt3[1] : gt3[2];
    const double r = isvar ? sigma * x * iD - cs * g + gt : 0.0;
    const double xh = row_matvec12(0.0, r, Ki);
    double xh3[3], cv5[5];
    foot_xyz(xh, xh3);
    cone_rows(xh3, mu, cv5);
    const double cvA = (t == 0) ? cv5[0] : (t == 1) ? cv5[1] : (t == 2) ? cv5[2] : cv5[3];
    const double zt[2] = {E[0] * cvA, E[1] * cv5[4]};
    const double xn = alpha * (xh * iD) + (1.0 - alpha) * x;
    const double dx = xn - x;
    x = xn;
    double dy[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; s_++) {
      const double zr = alpha * zt[s_] + (1.0 - alpha) * z[s_];
      double zn = zr + rho_inv * y[s_];
      zn = fmin(fmax(zn, ls[s_]), us[s_]);
      dy[s_] = rho * (zr - zn);
      y[s_] += dy[s_];
      z[s_] = zn;
    }
    if (iter % 25 == 0) {
      const double xs = D * x;
      double xs3[3];
      foot_xyz(xs, xs3);
      cone_rows(xs3, mu, cv5);
      const double cvsA = (t == 0) ? cv5[0] : (t == 1) ? cv5[1] : (t == 2) ? cv5[2] : cv5[3];
      const double cvs[2] = {cvsA, cv5[4]};
      double pres = 0, nz = 0, nax = 0, pres_s = 0, nz_s = 0, nax_s = 0;
#pragma unroll
      for (int s_ = 0; s_ < 2; s_++) {
        if (s_ == 1 && !isB) continue;  // slot B counts on the pad lane only
        const double axs = E[s_] * cvs[s_], rs = axs - z[s_];
        pres_s = fmax(pres_s, fabs(rs)); nz_s = fmax(nz_s, fabs(z[s_])); nax_s = fmax(nax_s, fabs(axs));
        pres = fmax(pres, fabs(iE[s_] * rs)); nz = fmax(nz, fabs(iE[s_] * z[s_])); nax = fmax(nax, fabs(iE[s_] * axs));
      }
      double ey5[5], aty3[3];
      foot_rows(E[0] * y[0], E[1] * y[1], ey5);
      cone_rows_t(ey5, mu, aty3);
      const double aty = (t == 0) ? aty3[0] : (t == 1) ? aty3[1] : aty3[2];
      double dres = 0, naty = 0, npx = 0, nq = 0, dres_s = 0, naty_s = 0, npx_s = 0, nq_s = 0;
      {
        double px = row_matvec12(0.0, xs, H);
        px *= cs;
        const double qh = cs * g;
        if (isvar) {
          dres = fabs(px + qh + aty); naty = fabs(aty); npx = fabs(px); nq = fabs(qh);
          dres_s = fabs(D * (px + qh + aty)); naty_s = fabs(D * aty); npx_s = fabs(D * px); nq_s = fabs(D * qh);
        }
      }
      pres = row_max(pres); nz = row_max(nz); nax = row_max(nax);
      dres = row_max(dres); naty = row_max(naty); npx = row_max(npx); nq = row_max(nq);
      pri_res = pres;
      dua_res = cinv * dres;
      last_np = fmax(nz, nax);
      last_nd = cinv * fmax(fmax(naty, npx), nq);
      bool done = false;
      if (pri_res > kOsqpInfty || dua_res > kOsqpInfty) {
        status = kStatusNonCvx;
        done = true;
      } else {
        const bool pok = pri_res < eps_abs + eps_rel * last_np;
        const bool dok = dua_res < eps_abs + eps_rel * last_nd;
        bool pinf = false, dinf = false;
        if (!pok) {  // is_primal_infeasible (all bounds finite)
          double ndy = 0.0, lhs = 0.0;
#pragma unroll
          for (int s_ = 0; s_ < 2; s_++) {
            if (s_ == 1 && !isB) continue;
            ndy = fmax(ndy, fabs(E[s_] * dy[s_]));
            lhs += us[s_] * fmax(dy[s_], 0.0) + ls[s_] * fmin(dy[s_], 0.0);
          }
          ndy = row_max(ndy);
          lhs = feet_sum(quad_sum(lhs));
          if (ndy > eps_prim_inf && lhs < -eps_prim_inf * ndy) {
            double edy5[5], at3[3];
            foot_rows(E[0] * dy[0], E[1] * dy[1], edy5);
            cone_rows_t(edy5, mu, at3);
            double na = isvar ? fabs((t == 0) ? at3[0] : (t == 1) ? at3[1] : at3[2]) : 0.0;
            na = row_max(na);
            pinf = na < eps_prim_inf * ndy;
          }
        }
        if (!dok) {  // is_dual_infeasible
          double ndx = isvar ? fabs(D * dx) : 0.0, qdx = isvar ? (cs * D * g) * dx : 0.0;
          ndx = row_max(ndx);
          qdx = feet_sum(quad_sum(qdx));
          if (ndx > eps_dual_inf && qdx < -cs * eps_dual_inf * ndx) {
            const double dxs = isvar ? D * dx : 0.0;
            double npd = fabs(cs * row_matvec12(0.0, dxs, H));
            npd = row_max(isvar ? npd : 0.0);
            if (npd < cs * eps_dual_inf * ndx) {
              double dxs3[3], adx[5];
              foot_xyz(dxs, dxs3);
              cone_rows(dxs3, mu, adx);  // Einv A_s dx = G (D dx)
              bool ok = true;
#pragma unroll
              for (int c = 0; c < 5; c++)
                if (adx[c] > eps_dual_inf * ndx || adx[c] < -eps_dual_inf * ndx) ok = false;
              dinf = (row_max(ok ? 0.0 : 1.0) == 0.0);
            }
          }
        }
        if (pok && dok) { status = kStatusSolved; done = true; }
        else if (pinf) { status = kStatusPrimalInf; done = true; }
        else if (dinf) { status = kStatusDualInf; done = true; }
      }
      if (done) break;
      if (iter % 200 == 0) {
        pres_s = row_max(pres_s); nz_s = row_max(nz_s); nax_s = row_max(nax_s);
        dres_s = row_max(dres_s); naty_s = row_max(naty_s); npx_s = row_max(npx_s); nq_s = row_max(nq_s);
        const double pn = pres_s / (fmax(nz_s, nax_s) + 1e-10);
        const double dn = dres_s / (fmax(fmax(naty_s, npx_s), nq_s) + 1e-10);
        double rho_new = rho * sqrt(pn / (dn + 1e-10));
        rho_new = fmin(fmax(rho_new, kRhoMin), kRhoMax);
        if (rho_new > rho * 5.0 || rho_new < rho / 5.0) {
          rho = rho_new;
          need_factor = true;
        }
      }
    }
  }
  WPH(5);
#ifdef QRW_PROFILE_WBC
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_wbc_ph[blockIdx.x * 16 + 8] = (unsigned long long)iter;
#endif
  if (iter > max_iter) iter = max_iter;
  if (status == kStatusUnsolved) {
    const bool pok = pri_res < 10 * eps_abs + 10 * eps_rel * last_np;
    const bool dok = dua_res < 10 * eps_abs + 10 * eps_rel * last_nd;
    status = (pok && dok) ? kStatusSolvedInaccurate : kStatusMaxIter;
  }
  const bool has_sol = (status == kStatusSolved || status == kStatusSolvedInaccurate || status == kStatusMaxIter);
  sol = has_sol ? D * x : nan("");
  if (!has_sol) { x = 0.0; z[0] = z[1] = y[0] = y[1] = 0.0; }
  if (valid) {
    if (isvar) { st[kWsX + v] = x; st[kWsG + v] = g; }
    st[kWsZ + 5 * j + t] = z[0]; st[kWsY + 5 * j + t] = y[0];
    if (isB) { st[kWsZ + 5 * j + 4] = z[1]; st[kWsY + 5 * j + 4] = y[1]; }
    if (j == 0 && t == 0) { st[kWsRho] = rho; st[kWsInit] = 1.0; }
  }
  iter_out = iter;
  status_out = status;
}

}  // namespace

__global__ __launch_bounds__(64, 1) void wbc16_kernel(WbcArgs a) {
  const int lane = threadIdx.x;
  const int r16 = lane & 15;
  const int j = r16 >> 2, t = r16 & 3;
  const int tc = (t < 3) ? t : 0;  // component this lane stores (pad lanes store nothing)
  const bool wr = t < 3;
  const int b = blockIdx.x * 4 + (lane >> 4);
  const bool valid = b < a.B;
  const int bb = valid ? b : 0;
  WPH(0);
  const LegC C = leg_consts(j);
  double* st = a.st + (size_t)bb * kWbcStItems;
  const double* qv = a.q + (size_t)bb * 19;
  const double* dqv = a.dq + (size_t)bb * 18;
  double q[3], dq[3], fc[3];
#pragma unroll
  for (int e = 0; e < 3; e++) { q[e] = qv[7 + 3 * j + e]; dq[e] = dqv[6 + 3 * j + e]; fc[e] = a.f_cmd[bb * 12 + 3 * j + e]; }
  const double contact = a.contacts[bb * 4 + j];
  const bool stance = (contact != 0.0);
  {  // k_since_contact (QP_WBC.py:65-66)
    double ks = st[kWsKsc + j];
    ks += contact;
    ks *= contact;
    if (valid && t == 0) st[kWsKsc + j] = ks;
  }
  WPH(1);
  const LegKin K = leg_kinematics(C, q);
  const V3 vf = dq[0] * K.J0 + dq[1] * K.J1 + dq[2] * K.J2;
  V3 acl;
  {
    const V3 w0 = dq[0] * K.a0, w1 = w0 + dq[1] * K.a1;
    const V3 da1 = dq[0] * cross(K.a0, K.a1);
    const V3 vp1 = cross(w0, K.p1 - K.p0), vp2 = vp1 + cross(w1, K.p2 - K.p1);
    acl = dq[0] * cross(K.a0, vf) + dq[1] * (cross(da1, K.pf - K.p1) + cross(K.a1, vf - vp1)) +
          dq[2] * (cross(da1, K.pf - K.p2) + cross(K.a1, vf - vp2));
  }
  const V3 goal = mk(a.pgoals[bb * 12 + 0 * 4 + j], a.pgoals[bb * 12 + 1 * 4 + j], a.pgoals[bb * 12 + 2 * 4 + j]);
  const V3 vgoal = mk(a.vgoals[bb * 12 + 0 * 4 + j], a.vgoals[bb * 12 + 1 * 4 + j], a.vgoals[bb * 12 + 2 * 4 + j]);
  const V3 agoal = mk(a.agoals[bb * 12 + 0 * 4 + j], a.agoals[bb * 12 + 1 * 4 + j], a.agoals[bb * 12 + 2 * 4 + j]);
  M3 iJ;
  inv3x3(K.J0, K.J1, K.J2, iJ);
  const V3 perr = goal - K.pf;
  V3 afeet = 100.0 * perr - (2.0 * sqrt(100.0)) * (vf - vgoal) + agoal;
  if (stance) afeet = 0.0 * afeet;
  afeet = afeet - acl;
  const V3 ddq3 = mul(iJ, afeet), dqc3 = mul(iJ, vgoal), qs3 = mul(iJ, perr);
  const double ddq[3] = {ddq3.x, ddq3.y, ddq3.z};
  M3 Rb;
  {
    const double x = qv[3], y = qv[4], z = qv[5], w = qv[6];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                 tyz = tz * y, tzz = tz * z;
    Rb.r0 = mk(1 - (tyy + tzz), txy - twz, txz + twy);
    Rb.r1 = mk(txy + twz, 1 - (txx + tzz), tyz - twx);
    Rb.r2 = mk(txz - twy, tyz + twx, 1 - (txx + tyy));
  }
  const V3 vb = mk(dqv[0], dqv[1], dqv[2]), wb = mk(dqv[3], dqv[4], dqv[5]);
  const V3 grav = mulT(Rb, mk(0.0, 0.0, QRW_SOLO12_MODEL.gravity));
  const V3 ab0 = grav + cross(wb, vb);
  const qrw_link_inertial& BL = QRW_SOLO12_MODEL.base;
  const double Ib[6] = {BL.inertia[0], BL.inertia[1], BL.inertia[2], BL.inertia[3], BL.inertia[4], BL.inertia[5]};
  const V3 cb = mk(BL.com[0], BL.com[1], BL.com[2]);
  M3 Id;
  Id.r0 = mk(1, 0, 0); Id.r1 = mk(0, 1, 0); Id.r2 = mk(0, 0, 1);
  V3 Fl, Ml;
  double tau1[3];
  leg_newton_euler(C, K, dq, ddq, wb, mk(0, 0, 0), ab0, Fl, Ml, tau1);
  const V3 Fb = BL.mass * (ab0 + cross(wb, cross(wb, cb)));
  const V3 Mb = inertia_apply(Ib, Id, mk(0, 0, 0)) + cross(wb, inertia_apply(Ib, Id, wb)) + cross(cb, Fb);
  const V3 F6 = feet_sum3(Fl) + Fb, M6 = feet_sum3(Ml) + Mb;
  const double rnea6[6] = {F6.x, F6.y, F6.z, M6.x, M6.y, M6.z};
  double Aj[6][3], gamma[6];
  {
    const V3 r = K.pf;
    const V3 rt[3] = {Rb.r0, Rb.r1, Rb.r2};
    double X[6][3];
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const V3 col = mk(rt[e].x, rt[e].y, rt[e].z);
      const V3 sc = cross(r, col);
      X[0][e] = col.x; X[1][e] = col.y; X[2][e] = col.z;
      X[3][e] = sc.x; X[4][e] = sc.y; X[5][e] = sc.z;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double s_ = 0.0;
#pragma unroll
      for (int e = 0; e < 3; e++) {
        const double xv = stance ? X[i][e] : 0.0;
        Aj[i][e] = (1.0 / a.Y[i]) * xv;
        s_ += xv * fc[e];
      }
      const double xf = feet_sum(s_);
      gamma[i] = (1.0 / a.Y[i]) * (xf - rnea6[i]);
    }
  }
  const double qd_leg = q[tc] + ((tc == 0) ? qs3.x : (tc == 1) ? qs3.y : qs3.z);   // this lane's component of the leg's q_des / v_des
  const double vd_leg = (tc == 0) ? dqc3.x : (tc == 1) ? dqc3.y : dqc3.z;
  if (valid) {
    if (a.qdes) {
      double* o = a.qdes + bb * 19;
      if (r16 < 7) o[r16] = 0.0;  // q_cmd[:7] is never written (solo12InvKin.py:67)
      if (wr) o[7 + 3 * j + t] = qd_leg;
    }
    if (a.vdes) {
      double* o = a.vdes + bb * 18;
      if (r16 < 6) o[r16] = 0.0;
      if (wr) o[6 + 3 * j + t] = vd_leg;
    }
    if (a.feet && wr) {  // feet_pos, feet_err, feet_vel as 3x4 each (QP_WBC.py:73-80): this lane's component
      double* o = a.feet + (size_t)bb * 36;
      o[t * 4 + j] = (t == 0) ? K.pf.x : (t == 1) ? K.pf.y : K.pf.z;
      o[12 + t * 4 + j] = (t == 0) ? perr.x : (t == 1) ? perr.y : perr.z;
      o[24 + t * 4 + j] = (t == 0) ? vf.x : (t == 1) ? vf.y : vf.z;
    }
  }
  WPH(2);
  // ---- QP data of this lane: its row of H = 0.1 A'A + 5 I (compute_matrices / update_PQ), its g, the bounds of its row slots
  double H[12], g, lo[2], up[2];
  {
    double coef[6], col0[6], col1[6], col2[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      coef[i] = ((tc == 0) ? Aj[i][0] : (tc == 1) ? Aj[i][1] : Aj[i][2]) * 0.1;
      col0[i] = Aj[i][0]; col1[i] = Aj[i][1]; col2[i] = Aj[i][2];
    }
    H[0] = h_entry<0>(col0, coef); H[1] = h_entry<0>(col1, coef); H[2] = h_entry<0>(col2, coef);
    H[3] = h_entry<1>(col0, coef); H[4] = h_entry<1>(col1, coef); H[5] = h_entry<1>(col2, coef);
    H[6] = h_entry<2>(col0, coef); H[7] = h_entry<2>(col1, coef); H[8] = h_entry<2>(col2, coef);
    H[9] = h_entry<3>(col0, coef); H[10] = h_entry<3>(col1, coef); H[11] = h_entry<3>(col2, coef);
#pragma unroll
    for (int cbi = 0; cbi < 12; cbi++)
      if (cbi / 3 == j && cbi % 3 == tc) H[cbi] += 5.0;
    double gv_ = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) gv_ += coef[i] * gamma[i];
    g = gv_;
    double gf[5];
    cone_rows(fc, 0.9, gf);
    const double gA = (t == 0) ? gf[0] : (t == 1) ? gf[1] : (t == 2) ? gf[2] : gf[3];
    lo[0] = -gA; up[0] = -gA + 25.0;
    lo[1] = -gf[4]; up[1] = -gf[4] + 25.0;
  }
  double sol;
  int it, stt;
  qp_solve16(H, g, lo, up, st, j, t, valid, sol, it, stt);
  WPH(6);
  // ---- epilogue: ddq_res = A f_res + gamma, second Newton-Euler with the base acceleration, torques, (fused) result check
  double sol3[3];
  foot_xyz(sol, sol3);
  const double fw[3] = {sol3[0] + fc[0], sol3[1] + fc[1], sol3[2] + fc[2]};
  double dd[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < 3; e++) acc = (e == 0) ? Aj[i][0] * sol3[0] : acc + Aj[i][e] * sol3[e];
    dd[i] = feet_sum(acc) + gamma[i];
  }
  double tau2[3], tff3[3];
  {
    const V3 alb = mk(dd[3], dd[4], dd[5]);
    const V3 ab1 = ab0 + mk(dd[0], dd[1], dd[2]);
    leg_newton_euler(C, K, dq, ddq, wb, alb, ab1, Fl, Ml, tau2);
    const V3 fbv = mulT(Rb, mk(fw[0], fw[1], fw[2]));  // Rb' f
    tff3[0] = tau2[0] - (stance ? dot(K.J0, fbv) : 0.0);
    tff3[1] = tau2[1] - (stance ? dot(K.J1, fbv) : 0.0);
    tff3[2] = tau2[2] - (stance ? dot(K.J2, fbv) : 0.0);
  }
  const double tff = (tc == 0) ? tff3[0] : (tc == 1) ? tff3[1] : tff3[2];
  if (valid) {
    if (a.tau_ff && wr) a.tau_ff[bb * 12 + 3 * j + t] = tff;
    if (a.f_with_delta && wr) a.f_with_delta[bb * 12 + 3 * j + t] = (t == 0) ? fw[0] : (t == 1) ? fw[1] : fw[2];
    if (a.ddq_res && r16 < 6) {
      const double dv = (r16 == 0) ? dd[0] : (r16 == 1) ? dd[1] : (r16 == 2) ? dd[2] : (r16 == 3) ? dd[3] : (r16 == 4) ? dd[4] : dd[5];
      a.ddq_res[bb * 6 + r16] = dv;
    }
    if (r16 == 0) { a.iters[bb] = it; a.status[bb] = stt; }
    if (a.c_cs) {
      // fused tail of the control iteration: Controller result + security_check (scripts/Controller.py:306-310,341-365; as
      // glue::result), one joint per lane, flags combined over the row
      double* cs = a.c_cs + bb;
      const size_t cB = (size_t)a.B;
      int err = (int)a.c_cs[bb + (size_t)glue::cERR * cB];
      const double c_qf = a.c_qfilt[bb * 19 + 7 + 3 * j + tc], c_vs = a.c_vsecu[bb * 12 + 3 * j + tc];
      const double qsec = (tc == 0) ? M_PI * 0.4 : (tc == 1) ? M_PI * 80 / 180 : M_PI;
      double e1 = 0.0, e2 = 0.0, e3 = 0.0, e4 = 0.0;
      if (wr) {
        if (fabs(c_qf) > qsec) e1 = 1.0;
        if (fabs(c_vs) > 50) e2 = 1.0;
        if (fabs(tff) > 8) e3 = 1.0;
        // not in the reference (its comparisons are blind to NaN): a non-finite command stops the robot, code 4 (glue::result)
        if (!(fabs(tff) <= 1.7976931348623157e308) || !(fabs(qd_leg) <= 1.7976931348623157e308) ||
            !(fabs(vd_leg) <= 1.7976931348623157e308)) e4 = 1.0;
      }
      e1 = row_max(e1); e2 = row_max(e2); e3 = row_max(e3); e4 = row_max(e4);
      if (err == 0) {  // the WBC counts this iteration: keep its references for the next one
        if (wr) {
          cs[(size_t)(glue::cQDES + 3 * j + t) * cB] = qd_leg;
          cs[(size_t)(glue::cVDES + 3 * j + t) * cB] = vd_leg;
        }
        if (e4 != 0.0) err = 4;
        if (e1 != 0.0) err = 1;
        if (e2 != 0.0) err = 2;
        if (e3 != 0.0) err = 3;
        if (r16 == 0) cs[(size_t)glue::cERR * cB] = (double)err;
      }
      if (wr) {
        double* r = a.c_result + (size_t)bb * 60;  // P | D | q_des | v_des | tau_ff
        const int i = 3 * j + t;
        if (err == 0) {
          r[i] = 3.0; r[12 + i] = 0.2; r[24 + i] = qd_leg; r[36 + i] = vd_leg; r[48 + i] = 0.8 * tff;
        } else {
          r[i] = 0.0; r[12 + i] = 0.1; r[24 + i] = 0.0; r[36 + i] = 0.0; r[48 + i] = 0.0;
        }
      }
      if (r16 == 0 && a.c_err) a.c_err[bb] = err;
    }
  }
  WPH(7);
}

// pseudoInverse<> of the reference (include/qrw/InvKin.hpp:60-66: JacobiSVD, V diag(1/s_i if s_i > eps max(r,c) s_0 else 0) U^H)
// for the 6 x 6 block M[:6,:6] of the stand-alone QPWBC::run (src/QPWBC.cpp:486-493), any matrix: one-sided (Hestenes) Jacobi
// SVD, one thread per instance -- rotate pairs of columns of A = Y (and of V) until all columns are orthogonal; then s_i = |a_i|,
// U = A diag(1/s), and V Sigma^+ U' = V diag(1/s_i^2) A'.  The pseudo-inverse is unique, so this agrees with any other accurate
// SVD to rounding (tests: numpy.linalg.pinv with the reference's threshold, and the oracle on symmetric blocks).  Not on the hot
// path: the reference's only caller masks the block to its diagonal (scripts/QP_WBC.py:93), which wbc_kernel handles directly.
__global__ __launch_bounds__(64) void pinv6_kernel(const double* M18, double* Yinv, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const double* M = M18 + (size_t)b * 324;
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int c = 0; c < 6; c++) { A[i][c] = M[i * 18 + c]; V[i][c] = (i == c) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int k = 0; k < 6; k++) { al += A[k][p] * A[k][p]; be += A[k][q] * A[k][q]; ga += A[k][p] * A[k][q]; }
        if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < 6; k++) {
          const double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - sn * aq;
          A[k][q] = sn * ap + c * aq;
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - sn * vq;
          V[k][q] = sn * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  double s2[6], smax = 0.0;
  for (int c = 0; c < 6; c++) {
    double n2 = 0.0;
    for (int k = 0; k < 6; k++) n2 += A[k][c] * A[k][c];
    s2[c] = n2;
    smax = fmax(smax, sqrt(n2));
  }
  const double tol = 2.220446049250313e-16 * 6.0 * smax;
  double* Yo = Yinv + (size_t)b * 36;
  for (int i = 0; i < 6; i++)
    for (int m = 0; m < 6; m++) {
      double acc = 0.0;
      for (int c = 0; c < 6; c++) {
        const double inv = (sqrt(s2[c]) > tol) ? 1.0 / s2[c] : 0.0;
        acc += V[i][c] * inv * A[m][c];
      }
      Yo[i * 6 + m] = acc;
    }
}

int pinv6_launch(const double* d_M18, double* d_Yinv, int B, hipStream_t stream) {
  hipLaunchKernelGGL(pinv6_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_M18, d_Yinv, B);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int wbc_launch(const WbcArgs& a, hipStream_t stream) {
  // full compute: sixteen lanes per instance (wbc16_kernel: B / 4 wavefronts, every SIMD of the chip at batch 4096) or one quad
  // per instance (wbc_kernel: B / 16 wavefronts of 1.55 x the length -- the better choice on a stream that owns few compute
  // units, e.g. the control loop's 32 of the asynchronous MPC mode); the handle decides (qrw_wbc_set_lanes, QRW_WBC16).  The
  // stand-alone modes 1-3 stay on the quad kernel.
  if (a.mode == 0 && a.lanes16) {
    hipLaunchKernelGGL(wbc16_kernel, dim3((a.B + 3) / 4), dim3(64), 0, stream, a);
  } else {
    hipLaunchKernelGGL(wbc_kernel, dim3((a.B + 15) / 16), dim3(64), 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace qrw
