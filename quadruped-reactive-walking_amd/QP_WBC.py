"""Drop-in for scripts/QP_WBC.py (class wbc_controller) — same constructor, compute() signature
and public attributes (scripts/Controller.py:153,260-310,345-379; scripts/LoggerControl.py:159-169),
one fused gfx950 kernel launch per compute() instead of Pinocchio + two bound C++ objects.

wbc_controller_batch is the batched variant (leading dimension B, device tensors).
"""
from time import time

import numpy as np

import qrw_hip


class _InvKinView:
    """What callers read from wbc.invKin (scripts/Controller.py:260,273,278; LoggerControl.py)."""

    def __init__(self):
        self.cpp_posf = np.zeros((4, 3))
        self.cpp_vf = np.zeros((4, 3))
        self.dq_cmd = np.zeros((18,))
        self.q_cmd = np.zeros((19,))
        self.foot_ids = np.array([10, 18, 26, 34])


class wbc_controller():
    """Whole body controller which contains an Inverse Kinematics step and a BoxQP step

    Args:
        dt (float): time step of the whole body control
    """

    def __init__(self, dt, N_SIMULATION):
        self.dt = dt
        self._b = qrw_hip.shared_batch1("wbc", dt_wbc=float(dt))  # the process-wide batch-1 handle (its WBC state)
        self.invKin = _InvKinView()

        self.M = np.zeros((18, 18))
        self.M[:6, :6] = np.diag(self._b.base_inertia_diag())  # the only part the QP reads (QP_WBC.py:93)
        self.Jc = np.zeros((12, 18))

        self.error = False
        self.k_since_contact = np.zeros((1, 4))

        # Logging (scripts/QP_WBC.py:36-42)
        N_SIMULATION = int(N_SIMULATION)
        self.k_log = 0
        self.log_feet_pos = np.zeros((3, 4, N_SIMULATION))
        self.log_feet_err = np.zeros((3, 4, N_SIMULATION))
        self.log_feet_vel = np.zeros((3, 4, N_SIMULATION))
        self.log_feet_pos_target = np.zeros((3, 4, N_SIMULATION))
        self.log_feet_vel_target = np.zeros((3, 4, N_SIMULATION))
        self.log_feet_acc_target = np.zeros((3, 4, N_SIMULATION))

        self.qdes = np.zeros((19, ))
        self.vdes = np.zeros((18, 1))
        self.tau_ff = np.zeros(12)
        self.f_with_delta = np.zeros((12, 1))
        self.indexes = [10, 18, 26, 34]
        self.tic = self.tac = self.toc = 0.0

    def compute(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals):
        """q (19x1), dq (18x1), f_cmd (12,), contacts (4,), pgoals/vgoals/agoals (3x4)."""
        contacts = np.asarray(contacts, dtype=np.float64).reshape(4)
        self.k_since_contact += contacts
        self.k_since_contact *= contacts

        self.tic = time()
        o = self._b.wbc_compute_host(np.asarray(q, dtype=np.float64).reshape(1, 19),
                                     np.asarray(dq, dtype=np.float64).reshape(1, 18),
                                     np.asarray(f_cmd, dtype=np.float64).reshape(1, 12), contacts[None],
                                     np.asarray(pgoals, dtype=np.float64)[None],
                                     np.asarray(vgoals, dtype=np.float64)[None],
                                     np.asarray(agoals, dtype=np.float64)[None])
        self.tac = time()  # the IK / QP split of the reference no longer exists: one fused launch

        k = self.k_log
        if k < self.log_feet_pos.shape[2]:
            self.log_feet_pos[:, :, k] = o["feet"][0, 0]
            self.log_feet_err[:, :, k] = o["feet"][0, 1]
            self.log_feet_vel[:, :, k] = o["feet"][0, 2]
            self.log_feet_pos_target[:, :, k] = pgoals[:, :]
            self.log_feet_vel_target[:, :, k] = vgoals[:, :]
            self.log_feet_acc_target[:, :, k] = agoals[:, :]
        self.feet_pos = o["feet"][0, 0].copy()
        self.feet_err = o["feet"][0, 1].copy()
        self.feet_vel = o["feet"][0, 2].copy()
        self.invKin.cpp_posf[:] = self.feet_pos.T
        self.invKin.cpp_vf[:] = self.feet_vel.T

        self.f_with_delta = o["f_with_delta"][0].reshape((-1, 1))
        self.tau_ff[:] = o["tau_ff"][0]
        self.vdes[:, 0] = o["vdes"][0]
        self.qdes[:] = o["qdes"][0]
        self.invKin.dq_cmd[:] = o["vdes"][0]
        self.invKin.q_cmd[:] = o["qdes"][0]
        self.ddq_res = o["ddq_res"][0].copy()

        self.toc = time()
        self.k_log += 1
        return 0


class wbc_controller_batch:
    """B instances, device-resident: compute_batch(q (B,19), dq (B,18), f_cmd (B,12), contacts (B,4),
    pgoals/vgoals/agoals (B,3,4)) -> dict of (B,...) CUDA tensors (tau_ff, qdes, vdes, f_with_delta,
    ddq_res, feet). Torch is used for device memory and the stream only."""

    def __init__(self, dt, batch, device=0):
        self.dt = dt
        self.B = int(batch)
        self._b = qrw_hip.Batch(self.B, dt_wbc=float(dt), device=device)
        self._out = None

    def compute_batch(self, q, dq, f_cmd, contacts, pgoals, vgoals, agoals):
        self._out = self._b.wbc_compute(q, dq, f_cmd, contacts, pgoals, vgoals, agoals, out=self._out)
        return self._out

    def stats(self):
        return self._b.wbc_stats()
