"""numpy restatement of the Controller glue around the hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Follows /root/reference/scripts/Controller.py line by line for the parts between the planners, the MPC and the
WBC: updateState :381-426, the WBC target / command assembly :258-296, the result
:306-310 and security_check :341-365.  PARITY UNPINNED (no reference vectors; the reference's Controller cannot be
imported here: pinocchio, pybullet, libquadruped_reactive_walking absent).
"""
import math

import numpy as np


def euler_to_quaternion(rpy):  # scripts/Estimator.py:672-684
    roll, pitch, yaw = rpy
    sr, cr = np.sin(roll / 2.), np.cos(roll / 2.)
    sp, cp = np.sin(pitch / 2.), np.cos(pitch / 2.)
    sy, cy = np.sin(yaw / 2.), np.cos(yaw / 2.)
    return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy]


def euler_to_rotation(roll, pitch, yaw):  # scripts/utils_mpc.py:87-107
    c_roll, s_roll = math.cos(roll), math.sin(roll)
    c_pitch, s_pitch = math.cos(pitch), math.sin(pitch)
    c_yaw, s_yaw = math.cos(yaw), math.sin(yaw)
    Rz = np.array([[c_yaw, -s_yaw, 0], [s_yaw, c_yaw, 0], [0, 0, 1]])
    Ry = np.array([[c_pitch, 0, s_pitch], [0, 1, 0], [-s_pitch, 0, c_pitch]])
    Rx = np.array([[1, 0, 0], [0, c_roll, -s_roll], [0, s_roll, c_roll]])
    return np.dot(Rz, np.dot(Ry, Rx))


class ControllerGlue:
    """State carried by scripts/Controller.py between iterations that the glue needs."""

    def __init__(self, q_init12, h_ref, dt_wbc):
        self.dt = dt_wbc
        self.h_ref = h_ref
        self.q = np.zeros((19, 1))
        self.q[0:7, 0] = np.array([0.0, 0.0, h_ref, 0.0, 0.0, 0.0, 1.0])  # Controller.py:119-121
        self.q[7:, 0] = q_init12
        self.v = np.zeros((18, 1))
        self.h_v = np.zeros((18, 1))
        self.v_ref = np.zeros((18, 1))
        self.yaw_estim = 0.0
        self.feet_a_cmd = np.zeros((3, 4))
        self.feet_v_cmd = np.zeros((3, 4))
        self.feet_p_cmd = np.zeros((3, 4))
        self.qdes = np.zeros(19)
        self.qdes[7:] = q_init12  # Controller.py:154
        self.vdes = np.zeros((18, 1))
        self.error = False
        self.error_flag = 0
        self.q_security = np.array([np.pi * 0.4, np.pi * 80 / 180, np.pi] * 4)  # :181

    def update_state(self, joy_v_ref, q_filt, v_filt, rpy):
        """Controller.updateState (:381-426), non-static gait branch."""
        self.v_ref[0:3, 0] = joy_v_ref[0:3]
        self.v_ref[3:6, 0] = joy_v_ref[3:6]
        self.v_ref[6:, 0] = 0.0
        Ryaw = np.array([[math.cos(self.yaw_estim), -math.sin(self.yaw_estim)],
                         [math.sin(self.yaw_estim), math.cos(self.yaw_estim)]])
        self.q[0:2, 0:1] = self.q[0:2, 0:1] + Ryaw @ self.v_ref[0:2, 0:1] * self.dt
        self.q[2, 0] = q_filt[2]
        self.yaw_estim += self.v_ref[5, 0] * self.dt
        self.q[3:7, 0] = euler_to_quaternion([rpy[0], rpy[1], self.yaw_estim])
        self.q[7:, 0] = q_filt[7:]
        self.v = np.asarray(v_filt, dtype=np.float64).reshape(18, 1).copy()
        hRb = euler_to_rotation(rpy[0], rpy[1], 0.0)
        self.h_v[0:3, 0:1] = hRb @ self.v[0:3, 0:1]
        self.h_v[3:6, 0:1] = hRb @ self.v[3:6, 0:1]
        oRh = np.eye(3)
        c, s = math.cos(self.yaw_estim), math.sin(self.yaw_estim)
        oRh[0:2, 0:2] = np.array([[c, -s], [s, c]])
        oTh = np.array([[self.q[0, 0]], [self.q[1, 0]], [0.0]])
        return oRh, oTh

    def wbc_inputs(self, x_f_mpc, xref, oRh, oTh, foot_pos, foot_vel, foot_acc):
        """Controller.compute :258-296 (non-static branch): returns q_wbc, b_v, f_cmd and the feet commands."""
        x_f_wbc = (x_f_mpc[:, 0]).copy()
        x_f_wbc[0] = self.dt * xref[6, 1]
        x_f_wbc[1] = self.dt * xref[7, 1]
        x_f_wbc[2] = self.h_ref
        x_f_wbc[3] = 0.0
        x_f_wbc[4] = 0.0
        x_f_wbc[5] = self.dt * xref[11, 1]
        x_f_wbc[6:12] = xref[6:, 1]
        q_wbc = np.zeros((19, 1))
        q_wbc[2, 0] = self.h_ref
        q_wbc[6, 0] = 1.0
        q_wbc[7:, 0] = self.qdes[7:]
        b_v = self.v.copy()
        b_v[:6, 0] = self.v_ref[:6, 0]
        b_v[6:, 0] = self.vdes[6:, 0]
        w = np.tile(self.v_ref[3:6, 0:1], (1, 4))
        self.feet_a_cmd = oRh.transpose() @ foot_acc \
            - np.cross(w, np.cross(w, self.feet_p_cmd, axis=0), axis=0) - 2 * np.cross(w, self.feet_v_cmd, axis=0)
        self.feet_v_cmd = oRh.transpose() @ foot_vel
        self.feet_v_cmd = self.feet_v_cmd - self.v_ref[0:3, 0:1] - np.cross(w, self.feet_p_cmd, axis=0)
        self.feet_p_cmd = oRh.transpose() @ (foot_pos - np.array([[0.0], [0.0], [self.h_ref]]) - oTh)
        return x_f_wbc, q_wbc, b_v

    def result(self, tau_ff, qdes, vdes, q_filt, v_secu):
        """Controller.compute :306-310 + security_check :341-365. Returns P, D, q_des, v_des, tau_ff."""
        if not self.error:  # the WBC block (:255-303) is skipped once an error has been raised
            self.qdes = np.asarray(qdes, dtype=np.float64).copy()
            self.vdes = np.asarray(vdes, dtype=np.float64).reshape(18, 1).copy()
        P, D = 3.0 * np.ones(12), 0.2 * np.ones(12)
        q_des, v_des, tau = self.qdes[7:].copy(), self.vdes[6:, 0].copy(), 0.8 * np.asarray(tau_ff, dtype=np.float64)
        if (self.error_flag == 0) and (not self.error):
            if np.any(np.abs(q_filt[7:]) > self.q_security):
                self.error, self.error_flag = True, 1
            if np.any(np.abs(v_secu) > 50):
                self.error, self.error_flag = True, 2
            if np.any(np.abs(tau_ff) > 8):
                self.error, self.error_flag = True, 3
        if self.error:
            P, D = np.zeros(12), 0.1 * np.ones(12)
            q_des, v_des, tau = np.zeros(12), np.zeros(12), np.zeros(12)
        return P, D, q_des, v_des, tau
