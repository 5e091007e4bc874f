/*
 * oracle/rbd_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see qrw_oracle.h).
 *
 * Restatement of the Pinocchio calls on the hot path (third-party, absent from
 * /root/reference and from this image; PARITY UNPINNED), specialised to nothing: a
 * generic spatial-algebra RNEA / forward kinematics over a joint table built from
 * include/qrw_solo12_model.h.  Reference call sites:
 *   scripts/solo12InvKin.py:47-59   computeJointJacobians, forwardKinematics(q,dq,0),
 *                                   updateFramePlacements, oMf.translation,
 *                                   getFrameVelocity / getFrameAcceleration /
 *                                   getFrameJacobian (LOCAL_WORLD_ALIGNED)
 *   scripts/QP_WBC.py:89-116        crba (neutral q), computeJointJacobians,
 *                                   getFrameJacobian(LWA)[:3], rnea x2
 * Pinocchio semantics restated: motions/forces are
 * (linear, angular); free-flyer velocity is expressed in the base frame; rnea
 * includes gravity (0,0,-9.81); getFrameAcceleration returns the SPATIAL
 * acceleration; fixed joints' bodies are lumped into the parent joint's body.
 */
#include <math.h>
#include <string.h>

#include "../include/qrw_solo12_model.h"
#include "qrw_oracle.h"

typedef struct { double R[9], p[3]; } se3;     /* x_parent = R x_child + p */
typedef struct { double lin[3], ang[3]; } sv6; /* spatial motion or force */

static void cross3(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matvec3(const double R[9], const double v[3], double o[3]) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matTvec3(const double R[9], const double v[3], double o[3]) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void matmul3(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
/* parent-frame composition: out = a * b */
static void se3_mul(const se3 *a, const se3 *b, se3 *out) {
  se3 t;
  matmul3(a->R, b->R, t.R);
  matvec3(a->R, b->p, t.p);
  for (int i = 0; i < 3; i++) t.p[i] += a->p[i];
  *out = t;
}
/* SE3::actInv on a motion: express a parent-frame motion in the child frame */
static void se3_actinv_motion(const se3 *M, const sv6 *v, sv6 *o) {
  double t[3], l[3];
  cross3(M->p, v->ang, t);
  for (int i = 0; i < 3; i++) l[i] = v->lin[i] - t[i];
  matTvec3(M->R, l, o->lin);
  matTvec3(M->R, v->ang, o->ang);
}
/* SE3::act on a force: express a child-frame force in the parent frame */
static void se3_act_force(const se3 *M, const sv6 *f, sv6 *o) {
  double l[3], a[3], t[3];
  matvec3(M->R, f->lin, l);
  matvec3(M->R, f->ang, a);
  cross3(M->p, l, t);
  for (int i = 0; i < 3; i++) { o->lin[i] = l[i]; o->ang[i] = a[i] + t[i]; }
}
/* v x m (motion cross motion) */
static void motion_cross(const sv6 *v, const sv6 *m, sv6 *o) {
  double a[3], b[3], c[3];
  cross3(v->ang, m->ang, a);
  cross3(v->ang, m->lin, b);
  cross3(v->lin, m->ang, c);
  for (int i = 0; i < 3; i++) { o->ang[i] = a[i]; o->lin[i] = b[i] + c[i]; }
}
/* v x* f (motion cross force) */
static void force_cross(const sv6 *v, const sv6 *f, sv6 *o) {
  double a[3], b[3], c[3];
  cross3(v->ang, f->lin, a);
  cross3(v->ang, f->ang, b);
  cross3(v->lin, f->lin, c);
  for (int i = 0; i < 3; i++) { o->lin[i] = a[i]; o->ang[i] = b[i] + c[i]; }
}
/* spatial inertia (mass, com offset `c` in the joint frame, inertia about the com) times motion, accumulated */
static void inertia_apply_add(const qrw_link_inertial *L, const double off[3], const sv6 *v, sv6 *h) {
  double c[3] = {L->com[0] + off[0], L->com[1] + off[1], L->com[2] + off[2]};
  const double *I = L->inertia;
  double Ic[9] = {I[0], I[1], I[2], I[1], I[3], I[4], I[2], I[4], I[5]};
  double t[3], hl[3], ha[3], u[3];
  cross3(c, v->ang, t);
  for (int i = 0; i < 3; i++) hl[i] = L->mass * (v->lin[i] - t[i]);
  matvec3(Ic, v->ang, ha);
  cross3(c, hl, u);
  for (int i = 0; i < 3; i++) { h->lin[i] += hl[i]; h->ang[i] += ha[i] + u[i]; }
}

/* joint table: 0 = base (free-flyer or fixed), 1+3*leg+{0,1,2} = HAA, HFE, KFE */
#define NJ 13
static int parent_of(int j) { return (j == 0) ? -1 : (((j - 1) % 3 == 0) ? 0 : j - 1); }
static int axis_of(int j) { return ((j - 1) % 3 == 0) ? 0 : 1; } /* HAA about x, HFE/KFE about y */

static void joint_placement(int j, double qj, se3 *M) {
  const qrw_leg_model *L = &QRW_SOLO12_MODEL.leg[(j - 1) / 3];
  const double *t = ((j - 1) % 3 == 0) ? L->haa_xyz : (((j - 1) % 3 == 1) ? L->hfe_xyz : L->kfe_xyz);
  double c = cos(qj), s = sin(qj);
  if (axis_of(j) == 0) {
    double R[9] = {1, 0, 0, 0, c, -s, 0, s, c};
    memcpy(M->R, R, sizeof(R));
  } else {
    double R[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    memcpy(M->R, R, sizeof(R));
  }
  memcpy(M->p, t, 3 * sizeof(double));
}

static void body_inertia_apply(int j, const sv6 *v, sv6 *h) {
  static const double zero[3] = {0, 0, 0};
  memset(h, 0, sizeof(*h));
  if (j == 0) {
    inertia_apply_add(&QRW_SOLO12_MODEL.base, zero, v, h);
    return;
  }
  const qrw_leg_model *L = &QRW_SOLO12_MODEL.leg[(j - 1) / 3];
  switch ((j - 1) % 3) {
    case 0: inertia_apply_add(&L->shoulder, zero, v, h); break;
    case 1: inertia_apply_add(&L->upper, zero, v, h); break;
    default: /* KFE body = lower leg + foot behind the fixed ankle */
      inertia_apply_add(&L->lower, zero, v, h);
      inertia_apply_add(&L->foot, L->foot_xyz, v, h);
  }
}

static void quat_to_R(const double *q, double R[9]) { /* q = (x, y, z, w), Eigen::Quaternion::toRotationMatrix */
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
         tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* RNEA core. free_flyer=1: q19/v18/a18 with gravity; free_flyer=0: fixed base, q12/v12/a12,
 * gravity switch `with_gravity`.  Outputs per-joint spatial v, a (joint frames), liMi, oMi, tau. */
typedef struct {
  se3 liMi[NJ], oMi[NJ];
  sv6 v[NJ], a[NJ], f[NJ];
} rbd_data;

static void kinematics_pass(int free_flyer, int with_gravity, const double *q, const double *v, const double *a,
                            rbd_data *d) {
  const double *qj = free_flyer ? q + 7 : q;
  const double *vj = free_flyer ? v + 6 : v;
  const double *aj = a ? (free_flyer ? a + 6 : a) : 0;
  sv6 a_gf0;
  memset(&a_gf0, 0, sizeof(a_gf0));
  if (with_gravity) a_gf0.lin[2] = QRW_SOLO12_MODEL.gravity; /* a_gf[0] = -gravity */
  memset(&d->liMi[0], 0, sizeof(se3));
  if (free_flyer) {
    quat_to_R(q + 3, d->liMi[0].R);
    memcpy(d->liMi[0].p, q, 3 * sizeof(double));
    for (int i = 0; i < 3; i++) { d->v[0].lin[i] = v[i]; d->v[0].ang[i] = v[3 + i]; }
    se3_actinv_motion(&d->liMi[0], &a_gf0, &d->a[0]); /* v x v = 0 for the free-flyer joint */
    if (a) for (int i = 0; i < 3; i++) { d->a[0].lin[i] += a[i]; d->a[0].ang[i] += a[3 + i]; }
  } else {
    d->liMi[0].R[0] = d->liMi[0].R[4] = d->liMi[0].R[8] = 1.0;
    memset(&d->v[0], 0, sizeof(sv6));
    d->a[0] = a_gf0;
  }
  d->oMi[0] = d->liMi[0];
  for (int j = 1; j < NJ; j++) {
    int p = parent_of(j), ax = axis_of(j);
    joint_placement(j, qj[j - 1], &d->liMi[j]);
    se3_mul(&d->oMi[p], &d->liMi[j], &d->oMi[j]);
    sv6 vJ;
    memset(&vJ, 0, sizeof(vJ));
    vJ.ang[ax] = vj[j - 1];
    se3_actinv_motion(&d->liMi[j], &d->v[p], &d->v[j]);
    d->v[j].ang[ax] += vj[j - 1];
    se3_actinv_motion(&d->liMi[j], &d->a[p], &d->a[j]);
    sv6 c;
    motion_cross(&d->v[j], &vJ, &c);
    for (int i = 0; i < 3; i++) { d->a[j].lin[i] += c.lin[i]; d->a[j].ang[i] += c.ang[i]; }
    if (aj) d->a[j].ang[ax] += aj[j - 1];
  }
}

void rbd_oracle_rnea(const double *q19, const double *v18, const double *a18, double *tau18) {
  rbd_data d;
  kinematics_pass(1, 1, q19, v18, a18, &d);
  for (int j = 0; j < NJ; j++) {
    sv6 h, Ia, vxh;
    body_inertia_apply(j, &d.a[j], &Ia);
    body_inertia_apply(j, &d.v[j], &h);
    force_cross(&d.v[j], &h, &vxh);
    for (int i = 0; i < 3; i++) { d.f[j].lin[i] = Ia.lin[i] + vxh.lin[i]; d.f[j].ang[i] = Ia.ang[i] + vxh.ang[i]; }
  }
  for (int j = NJ - 1; j >= 1; j--) {
    tau18[6 + j - 1] = d.f[j].ang[axis_of(j)];
    sv6 fp;
    se3_act_force(&d.liMi[j], &d.f[j], &fp);
    int p = parent_of(j);
    for (int i = 0; i < 3; i++) { d.f[p].lin[i] += fp.lin[i]; d.f[p].ang[i] += fp.ang[i]; }
  }
  for (int i = 0; i < 3; i++) { tau18[i] = d.f[0].lin[i]; tau18[3 + i] = d.f[0].ang[i]; }
}

/* M(:,j) = rnea(q, 0, e_j) - rnea(q, 0, 0): definition of the joint-space inertia matrix */
void rbd_oracle_crba(const double *q19, double *M) {
  double v0[18] = {0}, a[18], b0[18], col[18];
  memset(a, 0, sizeof(a));
  rbd_oracle_rnea(q19, v0, a, b0);
  for (int j = 0; j < 18; j++) {
    memset(a, 0, sizeof(a));
    a[j] = 1.0;
    rbd_oracle_rnea(q19, v0, a, col);
    for (int i = 0; i < 18; i++) M[i * 18 + j] = col[i] - b0[i];
  }
}

/* Composite rigid-body inertia of the whole robot in the base frame = crba(q)[:6,:6] */
void rbd_oracle_crba_base_block(const double *q19, double *M6) {
  rbd_data d;
  double v0[18] = {0};
  kinematics_pass(1, 0, q19, v0, 0, &d);
  memset(M6, 0, 36 * sizeof(double));
  for (int k = 0; k < 6; k++) { /* column k: apply each body's inertia to the unit base motion, bring the force back */
    sv6 e, tot;
    memset(&e, 0, sizeof(e));
    memset(&tot, 0, sizeof(tot));
    if (k < 3) e.lin[k] = 1.0; else e.ang[k - 3] = 1.0;
    for (int j = 0; j < NJ; j++) {
      /* bMj: placement of joint j in the base frame */
      se3 binv = d.oMi[0], bMj;
      /* inverse of base placement */
      double Rt[9] = {binv.R[0], binv.R[3], binv.R[6], binv.R[1], binv.R[4], binv.R[7], binv.R[2], binv.R[5], binv.R[8]};
      se3 inv;
      memcpy(inv.R, Rt, sizeof(Rt));
      matvec3(Rt, binv.p, inv.p);
      for (int i = 0; i < 3; i++) inv.p[i] = -inv.p[i];
      se3_mul(&inv, &d.oMi[j], &bMj);
      sv6 vj, hj, hb;
      se3_actinv_motion(&bMj, &e, &vj);
      body_inertia_apply(j, &vj, &hj);
      se3_act_force(&bMj, &hj, &hb);
      for (int i = 0; i < 3; i++) { tot.lin[i] += hb.lin[i]; tot.ang[i] += hb.ang[i]; }
    }
    for (int i = 0; i < 3; i++) { M6[i * 6 + k] = tot.lin[i]; M6[(3 + i) * 6 + k] = tot.ang[i]; }
  }
}

static void foot_world(const rbd_data *d, int leg, double pf[3]) {
  int j = 1 + 3 * leg + 2;
  matvec3(d->oMi[j].R, QRW_SOLO12_MODEL.leg[leg].foot_xyz, pf);
  for (int i = 0; i < 3; i++) pf[i] += d->oMi[j].p[i];
}

void rbd_oracle_fixed_feet(const double *q12, const double *dq12, double *posf, double *vf, double *wf, double *af,
                           double *Jf) {
  rbd_data d;
  kinematics_pass(0, 0, q12, dq12, 0, &d); /* forwardKinematics(q, dq, 0): no gravity term */
  memset(Jf, 0, 144 * sizeof(double));
  for (int leg = 0; leg < 4; leg++) {
    int jk = 1 + 3 * leg + 2;
    se3 iMf;
    memset(&iMf, 0, sizeof(iMf));
    iMf.R[0] = iMf.R[4] = iMf.R[8] = 1.0;
    memcpy(iMf.p, QRW_SOLO12_MODEL.leg[leg].foot_xyz, 3 * sizeof(double));
    se3 oMf;
    se3_mul(&d.oMi[jk], &iMf, &oMf);
    sv6 vl, al;
    se3_actinv_motion(&iMf, &d.v[jk], &vl);
    se3_actinv_motion(&iMf, &d.a[jk], &al);
    for (int i = 0; i < 3; i++) posf[leg * 3 + i] = oMf.p[i];
    matvec3(oMf.R, vl.lin, &vf[leg * 3]);
    matvec3(oMf.R, vl.ang, &wf[leg * 3]);
    matvec3(oMf.R, al.lin, &af[leg * 3]);
    for (int k = 0; k < 3; k++) {
      int j = 1 + 3 * leg + k;
      double ax[3] = {0, 0, 0}, axw[3], r[3], col[3];
      ax[axis_of(j)] = 1.0;
      matvec3(d.oMi[j].R, ax, axw);
      for (int i = 0; i < 3; i++) r[i] = oMf.p[i] - d.oMi[j].p[i];
      cross3(axw, r, col);
      for (int i = 0; i < 3; i++) Jf[(3 * leg + i) * 12 + (3 * leg + k)] = col[i];
    }
  }
}

void rbd_oracle_feet_jacobians(const double *q19, double *J) {
  rbd_data d;
  double v0[18] = {0};
  kinematics_pass(1, 0, q19, v0, 0, &d);
  memset(J, 0, 12 * 18 * sizeof(double));
  const double *Rb = d.oMi[0].R, *pb = d.oMi[0].p;
  for (int leg = 0; leg < 4; leg++) {
    double pf[3], rw[3];
    foot_world(&d, leg, pf);
    for (int i = 0; i < 3; i++) rw[i] = pf[i] - pb[i];
    for (int k = 0; k < 3; k++) { /* base linear and angular columns (velocity in the base frame) */
      double e[3] = {Rb[k], Rb[3 + k], Rb[6 + k]}, col[3];
      for (int i = 0; i < 3; i++) J[(3 * leg + i) * 18 + k] = e[i];
      cross3(e, rw, col);
      for (int i = 0; i < 3; i++) J[(3 * leg + i) * 18 + 3 + k] = col[i];
    }
    for (int k = 0; k < 3; k++) {
      int j = 1 + 3 * leg + k;
      double ax[3] = {0, 0, 0}, axw[3], r[3], col[3];
      ax[axis_of(j)] = 1.0;
      matvec3(d.oMi[j].R, ax, axw);
      for (int i = 0; i < 3; i++) r[i] = pf[i] - d.oMi[j].p[i];
      cross3(axw, r, col);
      for (int i = 0; i < 3; i++) J[(3 * leg + i) * 18 + 6 + 3 * leg + k] = col[i];
    }
  }
}
