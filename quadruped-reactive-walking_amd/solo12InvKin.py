"""Drop-in for scripts/solo12InvKin.py (class Solo12InvKin) without Pinocchio / example_robot_data.

The fixed-base Solo12 kinematics the reference gets from Pinocchio (scripts/solo12InvKin.py:47-59)
are evaluated by the gfx950 WBC kernel (mode: fixed-base feet kinematics) and the C++ InvKin
(src/InvKin.cpp:23-73) by its InvKin mode, both through libqrw_hip.so.
"""
import numpy as np

import libquadruped_reactive_walking as lrw
import qrw_hip


class Solo12InvKin:
    def __init__(self, dt):
        self.dt = dt
        self.InvKinCpp = lrw.InvKin(dt)
        self._kin = qrw_hip.shared_batch1("stateless", dt_wbc=float(dt))

        # Memory assignation for variables (scripts/solo12InvKin.py:19-28)
        self.cpp_posf = np.zeros((4, 3))
        self.cpp_vf = np.zeros((4, 3))
        self.cpp_wf = np.zeros((4, 3))
        self.cpp_af = np.zeros((4, 3))
        self.cpp_Jf = np.zeros((12, 12))

        self.ddq_cmd = np.zeros((18,))
        self.dq_cmd = np.zeros((18,))
        self.q_cmd = np.zeros((19,))

        # frame ids of FL_FOOT, FR_FOOT, HL_FOOT, HR_FOOT in the reference's free-flyer model
        # (scripts/QP_WBC.py:50); kept for callers that index with them
        self.foot_ids = np.array([10, 18, 26, 34])
        self.BASE_ID = self.foot_ids[0]

    def refreshAndCompute(self, q, dq, contacts, pgoals, vgoals, agoals):
        q12 = np.asarray(q, dtype=np.float64).reshape(12)
        dq12 = np.asarray(dq, dtype=np.float64).reshape(12)
        posf, vf, wf, af, Jf = self._kin.fixed_feet_host(q12[None], dq12[None])
        self.cpp_posf[:], self.cpp_vf[:], self.cpp_wf[:], self.cpp_af[:] = posf[0], vf[0], wf[0], af[0]
        self.cpp_Jf[:] = Jf[0]

        self.ddq_cmd[6:] = self.InvKinCpp.refreshAndCompute(np.array([contacts]), pgoals, vgoals, agoals,
                                                            self.cpp_posf, self.cpp_vf, self.cpp_wf,
                                                            self.cpp_af, self.cpp_Jf)
        self.dq_cmd[6:] = self.InvKinCpp.get_dq_cmd()
        self.q_cmd[7:] = q12 + self.InvKinCpp.get_q_step()
        return self.ddq_cmd
