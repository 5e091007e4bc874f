/*
 * Solo12 rigid-body model constants — the ONE data file for the robot model.
 *
 * This is DATA, not algorithm: it is read by the CPU oracle (oracle/) and by the
 * HIP product (quadruped-reactive-walking_amd/csrc/) alike, so that swapping in
 * the real example-robot-data `solo12.urdf` numbers (absent from this image and
 * from /root/reference) is a one-file change.
 *
 * Provenance: restated from memory of the public example-robot-data solo12.urdf
 * (un-pinned by the reference: README.md:15, scripts/PyBulletSimulator.py:225).
 * Cross-checks against numbers embedded in the reference itself:
 *   - total mass 2.50000279            (src/MPC.cpp:17, scripts/test_mpc.py:31)
 *   - foot y offset 0.0875+0.014+0.03745+0.008 = 0.14695, x offset 0.1946
 *                                      (scripts/Controller.py:132-133, src/InvKin.cpp:9)
 *   - leg segment length 0.16 (0.32*cos(0.7) standing height, scripts/test_mpc.py:39)
 *   - base inertia diag (0.00578574, 0.01938108, 0.02476124)
 *                                      (scripts/crocoddyl_class/MPC_crocoddyl.py:32, commented)
 * The mirror-sign conventions of the off-diagonal link inertias are the least
 * certain items.
 *
 * Tree (Pinocchio joint order, scripts/QP_WBC.py:50 / scripts/utils_mpc.py:144):
 *   legs FL, FR, HL, HR; per leg HAA (axis x), HFE (axis y), KFE (axis y),
 *   then a FIXED ankle carrying the foot frame (its inertia is lumped into the
 *   lower leg, as Pinocchio does for fixed joints).
 */
#ifndef QRW_SOLO12_MODEL_H_
#define QRW_SOLO12_MODEL_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double mass;
  double com[3];     /* centre of mass in the link (joint) frame            */
  double inertia[6]; /* ixx ixy ixz iyy iyz izz about the CoM, link axes    */
} qrw_link_inertial;

typedef struct {
  double haa_xyz[3];  /* HAA joint origin in base_link                      */
  double hfe_xyz[3];  /* HFE joint origin in the shoulder link              */
  double kfe_xyz[3];  /* KFE joint origin in the upper-leg link             */
  double foot_xyz[3]; /* fixed ankle/foot frame origin in the lower-leg link*/
  qrw_link_inertial shoulder, upper, lower, foot;
} qrw_leg_model;

typedef struct {
  qrw_link_inertial base;
  qrw_leg_model leg[4]; /* FL, FR, HL, HR */
  double gravity;       /* 9.81, along -z of the world (Pinocchio default)  */
} qrw_solo12_model;

/* sx = +1 front / -1 hind, sy = +1 left / -1 right */
#define QRW_LEG(sx, sy)                                                                         \
  {                                                                                             \
    {(sx) * 0.1946, (sy) * 0.0875, 0.0}, {0.0, (sy) * 0.014, 0.0}, {0.0, (sy) * 0.03745, -0.16}, \
        {0.0, (sy) * 0.008, -0.16},                                                             \
        {0.14853845,                                                                            \
         {(sx) * -0.078707, (sy) * 0.01, 0.0},                                                  \
         {0.00003024, (sx) * (sy) * 0.00004671, 0.0, 0.00041193, 0.0, 0.00041107}},             \
        {0.14853845,                                                                            \
         {0.00001377, (sy) * 0.01935853, -0.07870700},                                          \
         {0.00041107, 0.0, 0.00000009, 0.00041193, (sy) * 0.00004671, 0.00003024}},             \
        {0.03070001,                                                                            \
         {0.0, (sy) * 0.00787644, -0.08928215},                                                 \
         {0.00012024, 0.0, 0.0, 0.00012029, (sy) * 0.00000305, 0.00000216}},                    \
    {                                                                                           \
      0.00693606, {0.0, 0.0, 0.00035767}, { 0.00000057, 0.0, 0.0, 0.00000084, 0.0, 0.00000053 } \
    }                                                                                           \
  }

/* The initialiser as a macro so that device code can make its own constant copy of the SAME numbers
 * (csrc/wbc_kernel.hip: `constexpr qrw_solo12_model ... = QRW_SOLO12_MODEL_INIT`). */
#define QRW_SOLO12_MODEL_INIT                                                                \
  {                                                                                          \
    {1.16115091, {0.0, 0.0, 0.0}, {0.00578574, 0.0, 0.0, 0.01938108, 0.0, 0.02476124}},      \
        {QRW_LEG(1.0, 1.0), QRW_LEG(1.0, -1.0), QRW_LEG(-1.0, 1.0), QRW_LEG(-1.0, -1.0)}, 9.81 \
  }

static const qrw_solo12_model QRW_SOLO12_MODEL = QRW_SOLO12_MODEL_INIT;

#ifdef __cplusplus
}
#endif
#endif /* QRW_SOLO12_MODEL_H_ */
