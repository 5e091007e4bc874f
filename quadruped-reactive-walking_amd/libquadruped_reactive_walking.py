"""Drop-in for the reference's Boost.Python module `libquadruped_reactive_walking`
(python/gepadd.cpp) — hot-path classes only: MPC, QPWBC, InvKin.

Same class names, constructors, method names, argument meaning, return shapes and error
behaviour (every method returns 0 / never raises on solver status, src/MPC.cpp:648,
src/QPWBC.cpp:389), backed by the gfx950 kernels through libqrw_hip.so with a batch of one.
Return shapes follow eigenpy's conversion of the reference's return types: a dynamic
Eigen matrix with a single column comes back as a 1-D array (that is what lets
scripts/QP_WBC.py:114 do `ddq_with_delta[:6, 0] += deltaddq`), other matrices as 2-D.

The planner classes of the same module (StatePlanner, Gait, FootstepPlanner,
FootTrajectoryGenerator, Params; python/gepadd.cpp:44-181,230-281) are outside the hot path
 and are not provided here.
"""
import numpy as np

import qrw_hip


class MPC:
    """MPC(dt_in, n_steps_in, T_gait_in, N_gait) — python/gepadd.cpp:22-31, src/MPC.cpp."""

    def __init__(self, dt_in=None, n_steps_in=None, T_gait_in=None, N_gait=None):
        if dt_in is None:
            raise NotImplementedError("default-constructed MPC (src/MPC.cpp:34) has no parameters to run with")
        self._b = qrw_hip.Batch(1, n_steps=int(n_steps_in), N_gait=int(N_gait), dt_mpc=float(dt_in),
                                T_gait=float(T_gait_in))
        self._res = np.zeros((24, int(n_steps_in)))  # x_f_applied starts at zero (src/MPC.cpp:12)

    def run(self, num_iter, xref_in, fsteps_in):
        self._res = self._b.mpc_solve_host(np.asarray(xref_in, dtype=np.float64)[None],
                                           np.asarray(fsteps_in, dtype=np.float64)[None], int(num_iter))[0]
        return 0

    def get_latest_result(self):
        return self._res.copy()

    def get_gait(self):
        return self._b.mpc_gait(0)[0]

    def get_Sgait(self):
        return self._b.mpc_gait(0)[1].ravel()

    # not part of the reference binding: OSQP status / iteration count the reference ignores
    def solver_stats(self):
        s = self._b.mpc_stats()
        return {k: v[0] for k, v in s.items()}


class InvKin:
    """InvKin(dt_in) — python/gepadd.cpp:186-195, src/InvKin.cpp."""

    def __init__(self, dt_in=0.0):
        self._b = qrw_hip.Batch(1, dt_wbc=float(dt_in) if dt_in else 0.002)
        self._q_step = np.zeros(12)
        self._dq_cmd = np.zeros(12)

    def refreshAndCompute(self, contacts, goals, vgoals, agoals, posf, vf, wf, af, Jf):
        ddq, dq_cmd, q_step = self._b.invkin_host(np.asarray(contacts, dtype=np.float64).reshape(1, 4),
                                                  np.asarray(goals)[None], np.asarray(vgoals)[None],
                                                  np.asarray(agoals)[None], np.asarray(posf)[None],
                                                  np.asarray(vf)[None], np.asarray(wf)[None], np.asarray(af)[None],
                                                  np.asarray(Jf)[None])
        self._q_step, self._dq_cmd = q_step[0], dq_cmd[0]
        return ddq[0].copy()

    def get_q_step(self):
        return self._q_step.copy()

    def get_dq_cmd(self):
        return self._dq_cmd.copy()


class QPWBC:
    """QPWBC() — python/gepadd.cpp:217-224, src/QPWBC.cpp."""

    def __init__(self):
        self._b = qrw_hip.Batch(1)
        self._f_res = np.zeros(12)
        self._ddq_res = np.zeros(12)  # src/QPWBC.hpp:47 (12x1 until the first run resizes it to 6x1)
        self._H = np.zeros((12, 12))

    def run(self, M, Jc, f_cmd, RNEA, k_contact):
        f, d, H = self._b.qpwbc_host(np.asarray(M, dtype=np.float64)[None], np.asarray(Jc, dtype=np.float64)[None],
                                     np.asarray(f_cmd, dtype=np.float64).reshape(1, 12),
                                     np.asarray(RNEA, dtype=np.float64).reshape(1, 6))
        self._f_res, self._ddq_res, self._H = f[0], d[0], H[0]
        return 0

    def get_f_res(self):
        return self._f_res.copy()

    def get_ddq_res(self):
        return self._ddq_res.copy()

    def get_H(self):
        return self._H.copy()

    def solver_stats(self):
        s = self._b.wbc_stats()
        return {k: v[0] for k, v in s.items()}


def _out_of_scope(name):
    def ctor(*a, **k):
        raise NotImplementedError(
            "%s is a planner class outside the accelerated hot path (SURVEY.md §8(f)); use the reference's own "
            "libquadruped_reactive_walking for it" % name)
    return ctor


StatePlanner = _out_of_scope("StatePlanner")
Gait = _out_of_scope("Gait")
FootstepPlanner = _out_of_scope("FootstepPlanner")
FootTrajectoryGenerator = _out_of_scope("FootTrajectoryGenerator")
Params = _out_of_scope("Params")
