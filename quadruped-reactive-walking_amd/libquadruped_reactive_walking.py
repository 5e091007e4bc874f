"""Drop-in for the reference's Boost.Python module `libquadruped_reactive_walking`
(python/gepadd.cpp) — hot-path classes only: MPC, QPWBC, InvKin.

Same class names, constructors, method names, argument meaning, return shapes and error
behaviour (every method returns 0 / never raises on solver status, src/MPC.cpp:648,
src/QPWBC.cpp:389), backed by the gfx950 kernels through libqrw_hip.so with a batch of one.
Return shapes follow eigenpy's conversion of the reference's return types: a dynamic
Eigen matrix with a single column comes back as a 1-D array (that is what lets
scripts/QP_WBC.py:114 do `ddq_with_delta[:6, 0] += deltaddq`), other matrices as 2-D.

The planner classes of the same module (Gait, StatePlanner, FootstepPlanner, FootTrajectoryGenerator;
python/gepadd.cpp:44-181) are provided too, backed by planner_kernel.hip;
Params (YAML plumbing, python/gepadd.cpp:230-281) is not.
"""
import numpy as np

import qrw_hip


class MPC:
    """MPC(dt_in, n_steps_in, T_gait_in, N_gait) — python/gepadd.cpp:22-31, src/MPC.cpp."""

    def __init__(self, dt_in=None, n_steps_in=None, T_gait_in=None, N_gait=None):
        if dt_in is None:
            raise NotImplementedError("default-constructed MPC (src/MPC.cpp:34) has no parameters to run with")
        # the process-wide batch-1 handle of this configuration (its MPC state, if no other MPC object has it yet)
        self._b = qrw_hip.shared_batch1("mpc", n_steps=int(n_steps_in), N_gait=int(N_gait), dt_mpc=float(dt_in),
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
        self._b = qrw_hip.shared_batch1("stateless", dt_wbc=float(dt_in) if dt_in else 0.002)
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
        self._b = qrw_hip.shared_batch1("wbc")
        self._f_res = np.zeros(12)
        self._ddq_res = np.zeros(12)  # src/QPWBC.hpp:47 (12x1 until the first run resizes it to 6x1)
        self._H = np.zeros((12, 12))

    def run(self, M, Jc, f_cmd, RNEA, k_contact):
        # M[:6,:6] is pseudo-inverted as src/QPWBC.cpp:486 does (include/qrw/InvKin.hpp:60-66): directly inside the kernel for
        # the diagonal block the reference's caller always passes (scripts/QP_WBC.py:93), by a Jacobi SVD on the device otherwise
        M = np.asarray(M, dtype=np.float64)
        f, d, H = self._b.qpwbc_host(M[None], np.asarray(Jc, dtype=np.float64)[None],
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


# ------------------------------------------------------------------------------------------------------
# Planner classes (python/gepadd.cpp:44-181), SURVEY.md §8(f) ranks 1-2.  The reference wires them together by
# passing the Gait object to the other initialisers (scripts/Controller.py:119-137); here the Gait object owns the
# device-side planner state (a qrw_hip.Batch of one instance) and the other objects attach to it.
class _PlannerCore:
    def __init__(self, dt, T_gait, T_mpc, N_gait):
        self.dt, self.T_gait, self.T_mpc, self.N_gait = float(dt), float(T_gait), float(T_mpc), int(N_gait)
        self.n_steps = int(round(T_mpc / dt))
        if self.n_steps > self.N_gait or int(round(T_gait / dt)) > self.N_gait or self.n_steps + 1 > self.N_gait:
            # std::invalid_argument of Gait::initialize (src/Gait.cpp:30-31)
            raise ValueError("Sizes of matrices are too small for considered durations. Increase N_gait in config file.")
        self.cfg = dict(k_mpc=10, h_ref=0.2229, shoulders=qrw_hip.SHOULDERS.copy(), max_height=0.05, lock_time=0.07,
                        init_target=None, init_foot_pos=None, dt_wbc=0.002)
        self.batch = None
        self._build()

    def _build(self):
        # (re)create the handle whenever an initialise() call changes a parameter; state restarts like the reference
        n_steps = max(1, min(self.n_steps, 32))
        self.batch = qrw_hip.Batch(1, n_steps=n_steps, N_gait=self.N_gait, dt_mpc=self.dt, T_gait=self.T_gait,
                                   dt_wbc=self.cfg["dt_wbc"])
        c = self.cfg
        self.batch.planner_init(c["k_mpc"], c["h_ref"], c["shoulders"], c["max_height"], c["lock_time"],
                                c["init_target"], c["init_foot_pos"])


class Gait:
    """Gait() + initialize(dt_in, T_gait_in, T_mpc_in, N_gait) — src/Gait.cpp, python/gepadd.cpp:93-128."""

    def __init__(self):
        self._core = None

    def initialize(self, dt_in, T_gait_in, T_mpc_in, N_gait):
        self._core = _PlannerCore(dt_in, T_gait_in, T_mpc_in, N_gait)

    def _mat(self, which):
        return self._core.batch.planner_get(which, self._core.N_gait * 4).reshape(self._core.N_gait, 4)

    def getPastGait(self):
        return self._mat(0)

    def getCurrentGait(self):
        return self._mat(1)

    def getDesiredGait(self):
        return self._mat(2)

    def getIsStatic(self):
        return bool(self._core.batch.planner_get(4, 1)[0])

    def isNewPhase(self):
        return bool(self._core.batch.planner_get(3, 1)[0])

    def getQStatic(self):
        q = np.zeros(19)
        q[:7] = self._core.batch.planner_get(16, 7)
        return q

    def getRemainingTime(self):
        return float(self._core.batch.planner_get(5, 1)[0])

    def updateGait(self, k, k_mpc, q, joystickCode):
        if int(k_mpc) != self._core.cfg["k_mpc"]:
            raise ValueError("k_mpc differs from the value the planners were initialised with")
        self._core.batch.planner_call_host(qrw_hip.PLAN_GAIT, k=int(k), q7=np.asarray(q, dtype=np.float64).ravel()[:7][None],
                                           code=int(joystickCode), want=())

    def setGait(self, gaitMatrix):
        """Gait::setGait (src/Gait.cpp:262-269, python/gepadd.cpp:98): the reference prints the matrix it receives and returns
        false without touching the gait -- its body ends in a "Todo"."""
        print("Gait matrix received by setGait:")
        print(np.asarray(gaitMatrix))
        return False

    def changeGait(self, code, q):
        # Gait::changeGait alone (src/Gait.cpp:194-219): an updateGait with k not a multiple of k_mpc never rolls
        self._core.batch.planner_call_host(qrw_hip.PLAN_GAIT, k=1 if self._core.cfg["k_mpc"] > 1 else 0,
                                           q7=np.asarray(q, dtype=np.float64).ravel()[:7][None], code=int(code), want=())
        return self.getIsStatic()


class StatePlanner:
    """StatePlanner() + initialize(dt_in, T_mpc_in, h_ref_in) — src/StatePlanner.cpp, python/gepadd.cpp:44-62."""

    def __init__(self):
        self._b = None

    def initialize(self, dt_in, T_mpc_in, h_ref_in):
        n = int(round(T_mpc_in / dt_in))
        self._n = n
        self._b = qrw_hip.Batch(1, n_steps=n, N_gait=n + 4, dt_mpc=float(dt_in), T_gait=float(T_mpc_in))
        self._b.planner_init(h_ref=float(h_ref_in))
        self._xref = np.zeros((12, n + 1))

    def computeReferenceStates(self, q, v, vref, z_average):
        o = self._b.planner_call_host(qrw_hip.PLAN_STATE, q7=np.asarray(q, dtype=np.float64).ravel()[:7][None],
                                      v6=np.asarray(v, dtype=np.float64).ravel()[:6][None],
                                      vref6=np.asarray(vref, dtype=np.float64).ravel()[:6][None],
                                      z_average=float(z_average), want=("xref",))
        self._xref = o["xref"][0]

    def getReferenceStates(self):
        return self._xref.copy()

    def getNSteps(self):
        return self._n


class FootstepPlanner:
    """FootstepPlanner() + initialize(dt_in, dt_wbc_in, T_mpc_in, h_ref_in, shouldersIn, gaitIn, N_gait) —
    src/FootstepPlanner.cpp, python/gepadd.cpp:130-160."""

    def __init__(self):
        self._core = None

    def initialize(self, dt_in, dt_wbc_in, T_mpc_in, h_ref_in, shouldersIn, gaitIn, N_gait):
        self._core = gaitIn._core
        c = self._core.cfg
        c["dt_wbc"], c["h_ref"], c["shoulders"] = float(dt_wbc_in), float(h_ref_in), np.asarray(shouldersIn, dtype=np.float64)[:3, :4]
        self._core._build()

    def updateFootsteps(self, refresh, k, q, b_v, b_vref):
        o = self._core.batch.planner_call_host(qrw_hip.PLAN_FOOTSTEPS, k_footsteps=int(k), refresh=bool(refresh),
                                               q7=np.asarray(q, dtype=np.float64).ravel()[:7][None],
                                               v6=np.asarray(b_v, dtype=np.float64).ravel()[:6][None],
                                               vref6=np.asarray(b_vref, dtype=np.float64).ravel()[:6][None],
                                               want=("target",))
        return o["target"][0]

    def getFootsteps(self):
        return self._core.batch.planner_call_host(qrw_hip.PLAN_OUTPUTS, want=("fsteps",))["fsteps"][0]

    def getTargetFootsteps(self):
        return self._core.batch.planner_get(7, 12).reshape(3, 4)

    def getRz(self):
        """FootstepPlanner::getRz (src/FootstepPlanner.cpp:236, python/gepadd.cpp:123): the member as the last updateFootsteps left
        it -- the rotation by the base yaw that takes the target footsteps to the world frame (:214); zero but (2,2) = 1 before
        the first call (:10,:48).  The kernel keeps the yaw's cosine and sine in the planner state."""
        c, s = self._core.batch.planner_get(18, 2)
        return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


class FootTrajectoryGenerator:
    """FootTrajectoryGenerator() + initialize(maxHeightIn, lockTimeIn, targetFootstepIn, initialFootPosition,
    dt_tsid_in, k_mpc_in, gaitIn) — src/FootTrajectoryGenerator.cpp, python/gepadd.cpp:162-181."""

    def __init__(self):
        self._core = None

    def initialize(self, maxHeightIn, lockTimeIn, targetFootstepIn, initialFootPosition, dt_tsid_in, k_mpc_in, gaitIn):
        self._core = gaitIn._core
        c = self._core.cfg
        c["max_height"], c["lock_time"], c["k_mpc"], c["dt_wbc"] = float(maxHeightIn), float(lockTimeIn), int(k_mpc_in), float(dt_tsid_in)
        c["init_target"] = np.asarray(targetFootstepIn, dtype=np.float64)[:3, :4]
        c["init_foot_pos"] = np.asarray(initialFootPosition, dtype=np.float64)[:3, :4]
        self._core._build()

    def update(self, k, targetFootstep):
        self._core.batch.planner_call_host(qrw_hip.PLAN_TRAJ, k=int(k),
                                           target_in=np.asarray(targetFootstep, dtype=np.float64)[None], want=())

    def getTargetPosition(self):
        return self._core.batch.planner_get(17, 12).reshape(3, 4)

    def getFootPosition(self):
        return self._core.batch.planner_get(9, 12).reshape(3, 4)

    def getFootVelocity(self):
        return self._core.batch.planner_get(10, 12).reshape(3, 4)

    def getFootAcceleration(self):
        return self._core.batch.planner_get(11, 12).reshape(3, 4)


def _out_of_scope(name):
    def ctor(*a, **k):
        raise NotImplementedError(
            "%s is outside the accelerated path (SURVEY.md §8); use the reference's own module for it" % name)
    return ctor


Params = _out_of_scope("Params")
