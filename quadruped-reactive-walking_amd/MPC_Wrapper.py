"""Drop-in for scripts/MPC_Wrapper.py (class MPC_Wrapper): same constructor, solve(),
get_latest_result(), stop_parallel_loop() and first-iteration default result, with the OSQP
MPC running as one gfx950 kernel launch (libqrw_hip.so) instead of the bound C++ class.

  multiprocessing=False  the solve is synchronous, as scripts/MPC_Wrapper.py:128-148.
  multiprocessing=True   the reference runs the MPC in a child process polled through shared
                         flags (:150-225); here the solve is enqueued on its own HIP stream and
                         get_latest_result() returns the new result once its event has completed,
                         the previous one otherwise — the same observable protocol without a
                         second process.
The Crocoddyl variant (mpc_type=False) is outside the accelerated hot path.

MPC_Wrapper_batch is the batched variant: solve_batch / get_latest_result_batch over B
instances with device tensors.
"""
import numpy as np

import libquadruped_reactive_walking as MPC
import qrw_hip


def quaternionToRPY(quat):
    """Roll-pitch-yaw (3 x 1, ZYX convention) of a quaternion given as (x, y, z, w).

    Same contract as the helper the reference's wrapper uses for its first-iteration default result
    (scripts/utils_mpc.py:37-71, called at scripts/MPC_Wrapper.py:68), including its one quirk: roll and yaw are
    left at 0 whenever either argument of their arctan2 is exactly 0."""
    x, y, z, w = (float(v) for v in np.asarray(quat, dtype=np.float64).ravel()[:4])
    # the five entries of the (unnormalised) rotation matrix R(q) that the ZYX angles are read from
    r11 = w * w + x * x - y * y - z * z
    r21 = 2.0 * (x * y + w * z)
    r31 = 2.0 * (x * z - w * y)
    r32 = 2.0 * (y * z + w * x)
    r33 = w * w - x * x - y * y + z * z

    def angle(num, den):
        return float(np.arctan2(num, den)) if (num != 0.0 and den != 0.0) else 0.0

    pitch = float(np.arcsin(min(1.0, max(-1.0, -r31))))  # saturates at +-pi/2 at the gimbal lock
    return np.array([[angle(r32, r33)], [pitch], [angle(r21, r11)]])


class _EventFlag:
    """`.value` of the reference's multiprocessing.Value('b') flag `newResult`, answered from the solve's HIP event: True
    from the moment the asynchronous solve has finished until get_latest_result() has taken its result."""

    def __init__(self, wrapper):
        self._w = wrapper

    @property
    def value(self):
        ev = self._w._event
        return bool(ev is not None and ev.query())


class MPC_Wrapper:
    """Wrapper of the OSQP MPC (scripts/MPC_Wrapper.py:20-71).

    Args:
        mpc_type (bool): True to have PA's MPC (the only one accelerated), False for Crocoddyl
        dt (float): Time step of the MPC
        n_steps (int): Number of time steps in one gait cycle
        k_mpc (int): Number of inv dyn time step for one iteration of the MPC
        T_gait (float): Duration of one period of gait
        N_gait (int): number of rows of the gait / fsteps matrices
        q_init (array): the default position of the robot
        multiprocessing (bool): asynchronous solve (own HIP stream instead of a child process)
    """

    def __init__(self, mpc_type, dt, n_steps, k_mpc, T_gait, N_gait, q_init, multiprocessing=False):
        if not mpc_type:
            raise NotImplementedError("the Crocoddyl MPC (scripts/crocoddyl_class) is outside the accelerated path")
        self.f_applied = np.zeros((12,))
        self.not_first_iter = False
        self.k_mpc = k_mpc
        self.dt = dt
        self.n_steps = int(n_steps)
        self.T_gait = T_gait
        self.N_gait = int(N_gait)
        self.gait_memory = np.zeros(4)
        self.mpc_type = mpc_type
        self.multiprocessing = multiprocessing
        if multiprocessing:
            import torch

            self._torch = torch
            self._b = qrw_hip.shared_batch1("mpc", n_steps=self.n_steps, N_gait=self.N_gait, dt_mpc=float(dt),
                                            T_gait=float(T_gait))
            self._stream = torch.cuda.Stream()
            self._event = None
            self._pending = None
            # scripts/MPC_Wrapper.py:50-52: the shared flag the child process raises when a result is ready and
            # get_latest_result lowers when it takes it; callers poll `newResult.value` (scripts/test_mpc.py:65)
            self.newResult = _EventFlag(self)
        else:
            # Create the new version of the MPC solver object (scripts/MPC_Wrapper.py:58-61)
            self.mpc = MPC.MPC(dt, n_steps, T_gait, self.N_gait)

        # Setup initial result for the first iteration of the main control loop (:64-71)
        q_init = np.asarray(q_init, dtype=np.float64)
        x_init = np.zeros(12)
        x_init[0:3] = q_init[0:3, 0]
        x_init[3:6] = quaternionToRPY(q_init[3:7, 0]).ravel()
        self.last_available_result = np.zeros((24, self.n_steps))
        self.last_available_result[:, 0] = np.hstack((x_init, np.array([0.0, 0.0, 8.0] * 4)))

    def solve(self, k, xref, fsteps, gait):
        """scripts/MPC_Wrapper.py:73-104."""
        if self.multiprocessing:
            self.run_MPC_asynchronous(k, xref, fsteps)
        else:
            self.run_MPC_synchronous(k, xref, fsteps)

        # Bookkeeping of the default result (:89-102): past the first iterations the force rows are shifted one horizon
        # step to the left (the slice 12:12+n_steps of a 24-row array is rows 12..23), and when the last gait row is in
        # another phase than the first one the freed last column gets the static share m g / n_contacts on its stance feet.
        gait = np.asarray(gait)
        if k > 2:
            forces = self.last_available_result[12:(12 + self.n_steps), :]
            forces[:] = np.roll(forces, -1, axis=1)
            n_rows = 0  # rows until the first all-zero one
            while np.any(gait[n_rows, :]):
                n_rows += 1
            last_row = gait[n_rows - 1, :]
            if not np.array_equal(gait[0, :], last_row):
                share = 9.81 * 2.5 / np.sum(last_row)
                col = np.zeros(12)
                col[2::3] = np.where(last_row == 1, share, 0.0)
                self.last_available_result[12:, self.n_steps - 1] = col
        else:
            n_rows = 0
            while np.any(gait[n_rows, :]):  # the reference scans the gait on every call (IndexError if it has no zero row)
                n_rows += 1
        return 0

    def get_latest_result(self):
        """scripts/MPC_Wrapper.py:106-126."""
        if (self.not_first_iter):
            if self.multiprocessing:
                if self._event is not None and self._event.query():
                    self.last_available_result = self._pending.cpu().numpy()[0].copy()
                    self._event = None
                return self.last_available_result
            else:
                return self.f_applied
        else:
            self.not_first_iter = True
            return self.last_available_result

    def run_MPC_synchronous(self, k, xref, fsteps):
        """scripts/MPC_Wrapper.py:128-148."""
        self.mpc.run(int(k), xref.copy(), fsteps.copy())

        # Output of the MPC
        self.f_applied = self.mpc.get_latest_result()

    def run_MPC_asynchronous(self, k, xref, fsteps):
        """scripts/MPC_Wrapper.py:150-168, :186-223 — num_iter is k / k_mpc there (:235)."""
        torch = self._torch
        fsteps = np.asarray(fsteps, dtype=np.float64).copy()
        fsteps[np.isnan(fsteps)] = 0.0
        if self._event is not None:
            self._event.synchronize()  # one solve in flight, like the single child process
        with torch.cuda.stream(self._stream):
            dx = torch.from_numpy(np.ascontiguousarray(np.asarray(xref, dtype=np.float64))[None]).cuda()
            df = torch.from_numpy(np.ascontiguousarray(fsteps)[None]).cuda()
            self._pending = self._b.mpc_solve(dx, df, int(k / self.k_mpc))
            self._event = torch.cuda.Event()
            self._event.record(self._stream)
        return 0

    def stop_parallel_loop(self):
        """scripts/MPC_Wrapper.py:300-306 — nothing to stop: there is no child process."""
        if self.multiprocessing and self._event is not None:
            self._event.synchronize()
        return 0


class MPC_Wrapper_batch:
    """B instances, device-resident. solve_batch(k, xref (B,12,N+1), fsteps (B,N_gait,12)) runs one MPC
    iteration for every instance on the caller's stream; get_latest_result_batch() -> (B,24,N).
    Before the first solve it returns the reference's default result (scripts/MPC_Wrapper.py:64-71).

    groups (default 1: one handle, everything stream-ordered on the caller's stream; G > 1 opt-in): the fleet as independent
    stream groups, each with its own handle and stream (qrw_hip.StreamGroups' stream pool): a launch ends with its longest
    solve while most of the chip is idle, and with two groups in flight one group's stragglers run beside the other group's
    next solve (+10 % control steps/s at batch 4096).  Same results, bit for bit.  solve_batch then returns with the solves in
    flight on the groups' streams; get_latest_result_batch() makes the caller's stream wait for them.

    Input lifetime (both forms): xref / fsteps (and a tensor k) may be refilled in place or dropped as soon as solve_batch
    has returned -- with groups the call first copies them, on the caller's stream, into buffers the wrapper owns (two sets,
    alternating, so that the copy of call n + 1 does not wait for the solves of call n; 14 MB at 4096 robots), and the groups'
    streams read those.  (Rounds 4-5 read the caller's tensors from the groups' streams: a caller written for the
    stream-ordered single handle raced with the running solve -- ADVICE r5.)"""

    def __init__(self, dt, n_steps, T_gait, N_gait, batch, q_init=None, device=0, groups=1):
        import torch

        self._torch = torch
        self.B, self.n_steps, self.N_gait, self.device = int(batch), int(n_steps), int(N_gait), int(device)
        G = int(groups) if groups is not None else 1
        if G < 1 or self.B % G:
            raise qrw_hip.QrwError("batch %d does not split into %d equal groups" % (self.B, G))
        self.G, self.Bs = G, self.B // G
        dev = torch.device("cuda:%d" % device)
        self._bs = [qrw_hip.Batch(self.Bs, n_steps=self.n_steps, N_gait=self.N_gait, dt_mpc=float(dt), T_gait=float(T_gait),
                                  device=device) for _ in range(G)]
        self._b = self._bs[0]
        self._sl = [slice(g * self.Bs, (g + 1) * self.Bs) for g in range(G)]
        if G > 1:
            for g in range(G):
                if (device, g) not in qrw_hip.StreamGroups._streams:
                    qrw_hip.StreamGroups._streams[(device, g)] = torch.cuda.Stream(dev)
            self._streams = [qrw_hip.StreamGroups._streams[(device, g)] for g in range(G)]
            mk = lambda dt_, *shape: torch.empty(shape, dtype=dt_, device=dev)
            self._in = [(mk(torch.float64, self.B, 12, self.n_steps + 1), mk(torch.float64, self.B, self.N_gait, 12),
                         mk(torch.int32, self.B)) for _ in range(2)]
            self._in_done = [None, None]  # per input set: the events behind the groups' solves that read it
            self._n_calls = 0
        first = np.zeros((self.B, 24, self.n_steps))
        if q_init is not None:
            q_init = np.asarray(q_init, dtype=np.float64).reshape(self.B, 19)
            first[:, 0:3, 0] = q_init[:, 0:3]
            for b in range(self.B):
                first[b, 3:6, 0] = quaternionToRPY(q_init[b, 3:7]).ravel()
        first[:, 12:, 0] = np.array([0.0, 0.0, 8.0] * 4)
        self._result = torch.from_numpy(first).to(dev)
        self._out = None
        self._in_flight = False
        self.not_first_iter = False

    def solve_batch(self, k, xref, fsteps):
        torch = self._torch
        if self.G == 1:
            self._out = self._b.mpc_solve(xref, fsteps, k, out=self._out)
            return 0
        if self._out is None:
            self._out = torch.empty((self.B, 24, self.n_steps), dtype=torch.float64, device=xref.device)
        cur = torch.cuda.current_stream(self.device)
        i = self._n_calls & 1
        self._n_calls += 1
        xs, fs, ks = self._in[i]
        if self._in_done[i] is not None:  # the solves of two calls ago read this set: finished long ago in a loop, but say so
            for ev in self._in_done[i]:
                cur.wait_event(ev)
        xs.copy_(xref)
        fs.copy_(fsteps)
        k_is_tensor = isinstance(k, torch.Tensor)
        if k_is_tensor:
            ks.copy_(k)
        done = []
        for eng, st, sl in zip(self._bs, self._streams, self._sl):
            st.wait_stream(cur)  # the copies above are ready (and the previous result has been consumed)
            with torch.cuda.stream(st):
                eng.mpc_solve(xs[sl], fs[sl], ks[sl] if k_is_tensor else k, out=self._out[sl])
                ev = torch.cuda.Event()
                ev.record(st)
                done.append(ev)
        self._in_done[i] = done
        self._in_flight = True
        return 0

    def _join(self):
        if self._in_flight:
            cur = self._torch.cuda.current_stream(self.device)
            for st in self._streams:
                cur.wait_stream(st)
            self._in_flight = False

    def replay_batch(self, k0, xref_log, fsteps_log, out=None):
        """Recompute `mpc_x_f` for logged planner outputs: xref_log (K,B,12,N+1), fsteps_log (K,B,N_gait,12) as
        `LoggerControl` records them (`planner_xref`, `planner_fsteps`, scripts/LoggerControl.py:61-65,142-143) ->
        (K,B,24,N) (`mpc_x_f`, :76,:152).  Same results as K calls of solve_batch(k0 + s, ...), but one launch ordered per
        instance only (qrw_mpc_solve_sequence); for inputs known beforehand, not for a closed loop.  The last call's
        result becomes the latest result.  (A diagnostic for log replay; with groups the slices are replayed one after the
        other.)"""
        if self.G == 1:
            res = self._b.mpc_solve_sequence(xref_log, fsteps_log, k0, out=out)
            bad = self._b.mpc_sequence_timed_out()
        else:
            self._join()
            torch = self._torch
            K = int(xref_log.shape[0])
            res = out if out is not None else torch.empty((K, self.B, 24, self.n_steps), dtype=torch.float64, device=xref_log.device)
            bad = False
            for eng, sl in zip(self._bs, self._sl):
                res[:, sl] = eng.mpc_solve_sequence(xref_log[:, sl].contiguous(), fsteps_log[:, sl].contiguous(), k0)
                bad = eng.mpc_sequence_timed_out() or bad
        if bad:  # (synchronises) calls that never ran are NaN in res
            raise qrw_hip.QrwError("replay_batch: the sequence kernel's task queue gave up waiting (2 s without progress); "
                                   "calls that did not run are NaN in the result")
        self._out = res[-1]
        return res

    def get_latest_result_batch(self):
        """scripts/MPC_Wrapper.py:106-126, synchronous branch: the first call returns the default forces."""
        if self.G > 1:
            self._join()
        if self.not_first_iter:
            return self._out if self._out is not None else self._result
        self.not_first_iter = True
        return self._result

    def stats(self):
        if self.G > 1:
            self._join()
        st = [b.mpc_stats() for b in self._bs]
        return {k: np.concatenate([s_[k] for s_ in st]) for k in st[0]}
