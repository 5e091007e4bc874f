"""Batched, device-resident mirror of the reference's main control loop for B Solo12 instances.

Mirrors `Controller.compute` of /root/reference/scripts/Controller.py:200-326 between the estimator output and the
quantities sent to the control board: updateState -> Gait / FootstepPlanner / FootTrajectoryGenerator / StatePlanner
-> MPC every k_mpc iterations -> WBC target assembly -> InvKin + QPWBC -> result + security_check.  Every stage is a
HIP kernel of libqrw_hip.so working on the same handle; nothing leaves HBM.  The joystick, the state estimator, the
PyBullet / masterboard devices and the loggers are outside the accelerated path: their outputs (reference velocity,
filtered q / v, roll-pitch, joint velocities for the security check) are inputs here.
"""
import os
import numpy as np

import qrw_hip


class Result:
    """scripts/Controller.py:14-23 for B instances: views into one (B,5,12) device tensor."""

    def __init__(self, t):
        self.P, self.D, self.q_des, self.v_des, self.tau_ff = (t[:, i] for i in range(5))


# Largest fleet whose WORST paced control iteration (the one that carries the MPC solve) stays inside the reference's 2 ms slot
# (dt_wbc, /root/reference/src/config_solo12.yaml:6), per mode -- measured on one MI355X, horizon 16, k_mpc 10 (bench.py
# `realtime_slot`, BENCH_r05 / profiles/r6_bench_line.json): synchronous single handle 5.6 ms at 4096 robots, two staggered
# stream groups 3.2 ms, asynchronous MPC 0.24 ms.
REALTIME_SLOT_FITS = (("sync", 1024), ("staggered_groups", 2048), ("async", 4096))
REALTIME_SLOT_MEASURED_AT = 0.002


def recommended_mode(batch, deadline=REALTIME_SLOT_MEASURED_AT):
    """Constructor arguments of the cheapest mode whose worst iteration is expected to fit `deadline` seconds at this fleet size:
    {} (synchronous, one handle: every robot steps on every tick, the reference loop), dict(groups=2, stagger=True) (half the
    fleet runs k_mpc / 2 ticks behind the other half) or dict(multiprocessing=True) (the reference's asynchronous MPC: the
    solve is off the tick and its result is adopted when it is there).  From REALTIME_SLOT_FITS, scaled linearly in the deadline
    (the worst iteration is a whole MPC launch, proportional to the fleet from ~1000 robots on); None if no mode is expected to
    fit (shard the fleet over more GPUs, sharding.py)."""
    scale = float(deadline) / REALTIME_SLOT_MEASURED_AT
    for mode, fits in REALTIME_SLOT_FITS:
        if int(batch) <= fits * scale and not (mode == "staggered_groups" and int(batch) % 2):
            return {"sync": {}, "staggered_groups": dict(groups=2, stagger=True), "async": dict(multiprocessing=True)}[mode]
    return None


def auto_groups(batch, groups=None, multiprocessing=False):
    """Number of stream groups a fleet is stepped as: what the caller says, ONE handle otherwise -- every robot steps on every
    tick and self.k is every robot's clock, as in the reference loop (scripts/Controller.py:200-326).  Stream groups are opt-in:
    unstaggered they are bit-identical to one handle but slower in the 1:10 loop (twice the launches: 10.0 M against 12.4 M
    iterations/s at 4096 robots), staggered they halve the fleet's worst iteration (3.2 against 5.6 ms) at the price of half the
    fleet running k_mpc / 2 ticks behind -- a change of behaviour a caller has to ask for (recommended_mode / for_deadline say
    when).  (Rounds 4-5 chose two staggered groups by themselves from 2048 robots on; ADVICE r5: discontinuous in the fleet
    size and not the reference's loop.)"""
    return int(groups) if groups is not None else 1


class Controller_batch:
    def __new__(cls, batch, *args, groups=None, **kwargs):
        # more than one group: the fleet as independent stream groups, see Controller_groups
        mp = kwargs.get("multiprocessing", args[9] if len(args) > 9 else False)  # (position of `multiprocessing` in __init__)
        if cls is Controller_batch and auto_groups(batch, groups, mp) > 1:
            return super().__new__(Controller_groups)
        return super().__new__(cls)

    def __init__(self, batch, q_init, dt_wbc=0.002, dt_mpc=0.02, k_mpc=10, T_gait=0.32, T_mpc=0.32, N_gait=20,
                 h_ref=0.2229, device=0, multiprocessing=False, loop_cus=None, mpc_lag=None, fused=True, groups=None,
                 stagger=None, deadline=None, _out_views=None):
        """q_init: (12,) or (B,12) initial joint angles (Controller.__init__ q_init, scripts/Controller.py:60).

        groups: None or 1 = one handle (the default at every fleet size); G > 1 = the fleet as G stream groups
        (Controller_groups).  Unstaggered, every robot's results are the single handle's, bit for bit.  Measured at batch 4096 in
        the 1:10 loop: two groups joined on the caller's stream every iteration and NOT staggered run at 10.0 M iterations/s
        against 12.4 M for the single handle (twice the launches) with the same worst iteration (5.6 ms, the one that carries
        the solve); staggered, the worst iteration of the fleet drops to 3.1 ms, and 2048 robots fit the reference's 2 ms slot
        where one handle fits 1024 (bench.py, realtime_slot).

        stagger (groups > 1 only, default False): group g starts g * k_mpc / groups fleet ticks late, so the groups' MPC solves
        fall on different ticks and one group's solve runs beside the other's plain iterations (Controller_groups).  THIS CHANGES
        WHAT THE ROBOTS DO: a robot of group g runs the single-handle controller started lag_of(g) ticks late -- at fleet tick t
        it returns its iteration t - lag_of(g) --, and until then it is commanded to hold q_init (Result: P 3, D 0.2, q_des =
        q_init, zero v_des and tau_ff); self.k is group 0's clock.

        deadline (seconds, e.g. dt_wbc; None = no monitor): every compute() is bracketed by two events on the caller's stream and
        the device time of finished iterations is read back without synchronising; the first time one exceeds the deadline a
        RuntimeWarning names the measured worst case and the mode recommended_mode() would pick (the first 2 k_mpc iterations --
        QP set-up and cold start of the first two solves -- are start-up and not accounted).  `overrun` holds
        (worst seconds, iteration) from then on.  Controller_batch.for_deadline(...) builds that mode in the first place.

        multiprocessing=True mirrors the reference's asynchronous MPC (scripts/MPC_Wrapper.py:150-298, a child process
        on its own core polled through a shared flag) with HIP streams: the MPC solves on a stream restricted to all
        compute units but `loop_cus`, the control loop (planners, glue, WBC) on a stream restricted to those
        `loop_cus` units, so an iteration never queues behind a running solve (default: batch / 64 -- one SIMD for every
        wavefront of the loop's kernels, the lowest iteration latency, which is what the mode is for; 32 gives the highest
        free-running rate at batch 4096: +14 % at 0.20 instead of 0.13 ms median, profiles/r4_async_loop_cus.txt).  A finished solve is adopted by the
        first iteration that finds its event complete (mpc_lag=None, what the reference's flag polling does), or --
        deterministic, for tests and replay -- exactly `mpc_lag` iterations after it was issued.
        The masked streams are ordinary (blocking) HIP streams: they synchronise with the legacy default stream, so call
        compute() from a non-default stream (`with torch.cuda.stream(torch.cuda.Stream()): ...`) or the overlap is lost."""
        import torch

        self._torch = torch
        self.B = int(batch)
        self.dev = torch.device("cuda:%d" % device)
        self.k, self.k_mpc, self.h_ref, self.dt_wbc = 0, int(k_mpc), float(h_ref), float(dt_wbc)
        self.n_steps = int(round(T_mpc / dt_mpc))
        self._b = qrw_hip.Batch(self.B, n_steps=self.n_steps, N_gait=int(N_gait), dt_mpc=float(dt_mpc),
                                T_gait=float(T_gait), dt_wbc=float(dt_wbc), device=device)
        qi = np.broadcast_to(np.asarray(q_init, dtype=np.float64).reshape(-1, 12), (self.B, 12))
        self._b.planner_init(k_mpc=self.k_mpc, h_ref=self.h_ref)
        self._b.controller_init(torch.from_numpy(np.array(qi, dtype=np.float64, order="C")).to(self.dev), self.h_ref)
        # Default MPC result before the first solve is collected (scripts/MPC_Wrapper.py:64-71,123-126)
        first = np.zeros((self.B, 24, self.n_steps))
        first[:, 2, 0] = self.h_ref
        first[:, 12:, 0] = np.array([0.0, 0.0, 8.0] * 4)
        self._mpc_default = torch.from_numpy(first).to(self.dev)
        self._mpc_out = None
        self._not_first_iter = False
        self._st = self._plan = self._wi = self._wbc = self._res = None
        self.x_f_mpc = self._mpc_default
        self.fused = bool(fused)  # two launches per iteration (+ the solve) instead of five; same arithmetic
        self._pre = self._post = self._fast = None
        self.result = self.error_flag = None
        if _out_views is not None:
            # a stream group of a larger fleet: result / error flag are written straight into the fleet's tensors
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=self.dev)
            B = self.B
            if self.fused:
                self._post = dict(tau_ff=mk(B, 12), qdes=mk(B, 19), vdes=mk(B, 18), f_with_delta=mk(B, 12), ddq_res=mk(B, 6),
                                  feet=mk(B, 3, 3, 4), result=_out_views["result"], error_flag=_out_views["error_flag"])
            else:
                self._res = dict(result=_out_views["result"], error_flag=_out_views["error_flag"])
        self.multiprocessing = bool(multiprocessing)
        self.mpc_lag = mpc_lag
        self._init_deadline(deadline)
        self.wbc_lanes = 16  # lanes per robot of the full WBC step (qrw_wbc_set_lanes): wbc16_kernel unless chosen otherwise below
        if self.multiprocessing:
            n_cu = qrw_hip.device_cu_count(device)
            if loop_cus is None:
                # one SIMD for every wavefront of the loop's kernels (a quad per robot: batch / 16 wavefronts, four SIMDs per compute
                # unit), in steps of the 8 XCDs, at most a quarter of the chip: 64 compute units at batch 4096
                loop_cus = min(max(8, 8 * ((self.B + 511) // 512)), n_cu // 4)
            loop_cus = max(1, min(int(loop_cus), n_cu - 1))
            self._s_loop = qrw_hip.CuStream(device, 0, loop_cus)
            self._s_mpc = qrw_hip.CuStream(device, loop_cus, n_cu - loop_cus)
            lanes_exp = os.environ.get("QRW_EXP_ASYNC_LANES")  # experiments only (scripts/gpu_async_cus_exp.py)
            if (lanes_exp == "4") if lanes_exp else (4 * loop_cus < self.B // 4):
                # the loop's stream owns too few SIMDs for one wavefront per four robots in a single round: the quad WBC kernel
                # (a quarter of the wavefronts, 1.55 x the length) is the faster one there (measured at batch 4096 on 32 compute
                # units: 0.21 against 0.31 ms median paced latency)
                self._b.wbc_set_lanes(4)
                self.wbc_lanes = 4
            Ng = int(N_gait)
            mk = lambda *shape: torch.empty(shape, dtype=torch.float64, device=self.dev)
            self._snap = [(mk(self.B, 12, self.n_steps + 1), mk(self.B, Ng, 12)) for _ in range(3)]
            self._outs = [mk(self.B, 24, self.n_steps) for _ in range(3)]
            self._ev_in = [torch.cuda.Event() for _ in range(3)]
            self._ev_done = [torch.cuda.Event() for _ in range(3)]
            self._n_issued = 0          # number of solves issued so far
            self._pending = []          # [(solve index, iteration it was issued at)] not adopted yet
            self._adopted = None        # index (mod 3) of the buffer the loop currently reads

    @classmethod
    def for_deadline(cls, batch, q_init, deadline=None, dt_wbc=0.002, **kw):
        """The fleet in the cheapest mode whose worst iteration is expected inside `deadline` seconds (default: dt_wbc, the
        slot the reference loop runs in, src/config_solo12.yaml:6) -- recommended_mode(): synchronous up to 1024 robots, two
        staggered stream groups up to 2048, asynchronous MPC above (measured at 2 ms) --, with the deadline monitor armed."""
        deadline = float(dt_wbc if deadline is None else deadline)
        mode = recommended_mode(batch, deadline)
        if mode is None:
            raise qrw_hip.QrwError("no mode of one GPU has been measured to step %d robots inside %.3g s: shard the fleet (sharding.py)"
                                   % (batch, deadline))
        return Controller_batch(batch, q_init, dt_wbc=dt_wbc, deadline=deadline, **dict(mode, **kw))

    # ---- deadline monitor (deadline=...): device time of every iteration, read back without synchronising
    def _init_deadline(self, deadline):
        self.deadline = None if deadline is None else float(deadline)
        self.overrun = None        # (worst seconds, iteration) once an iteration exceeded the deadline
        self.worst_iteration = 0.0  # worst measured so far (seconds; finished iterations only)
        self._dl_pairs = []
        # iterations before this one are not accounted: the loop's first two solves (QP set-up, cold start) are start-up, as the
        # reference's own first iterations are (OSQP set-up inside the first MPC call, scripts/MPC_Wrapper.py:128-148)
        self._dl_skip = 2 * self.k_mpc + max(getattr(self, "_delay", [0]))

    def _deadline_begin(self):
        torch = self._torch
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), self.k)
        ev[0].record(torch.cuda.current_stream(self.dev))
        return ev

    def _deadline_end(self, ev):
        ev[1].record(self._torch.cuda.current_stream(self.dev))
        self._dl_pairs.append(ev)
        self._deadline_account()

    def _deadline_account(self):
        while self._dl_pairs and self._dl_pairs[0][1].query():
            a, b, k = self._dl_pairs.pop(0)
            if k < self._dl_skip:
                continue  # start-up: the first two solves of every robot set the QP up and start cold (about twice the ADMM iterations)
            t = a.elapsed_time(b) * 1e-3
            if t > self.worst_iteration:
                self.worst_iteration = t
            if t > self.deadline and (self.overrun is None or t > self.overrun[0]):
                first = self.overrun is None
                self.overrun = (t, k)
                if first:
                    import warnings

                    rec = recommended_mode(self.B, self.deadline)
                    how = ("more GPUs (sharding.py)" if rec is None else
                           "Controller_batch(..., %s)" % ", ".join("%s=%r" % kv for kv in rec.items()) if rec else
                           "this mode (is the device shared?)")
                    warnings.warn("control iteration %d of %d robots took %.2f ms on the device, deadline %.2f ms (the iterations "
                                  "that carry the MPC solve are the long ones); recommended for this fleet and deadline: %s"
                                  % (k, self.B, t * 1e3, self.deadline * 1e3, how), RuntimeWarning, stacklevel=4)

    def deadline_flush(self):
        """Wait for the iterations issued so far and account them; returns the worst iteration's device time (seconds)."""
        if self.deadline is not None and self._dl_pairs:
            self._dl_pairs[-1][1].synchronize()
            self._deadline_account()
        return self.worst_iteration

    def compute(self, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code=0):
        """One control iteration for every instance. All arguments CUDA float64 with leading dimension B:
        joy_v_ref (B,6), q_filt (B,19), v_filt (B,18), rpy (B,3), v_secu (B,12). Returns the Result views."""
        if self.deadline is not None:
            ev = self._deadline_begin()
            r = self._compute_any(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code)
            self._deadline_end(ev)
            return r
        return self._compute_any(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code)

    def _compute_any(self, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code=0):
        if not self.multiprocessing:
            return self._compute(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code)
        torch = self._torch
        caller = torch.cuda.current_stream(self.dev)
        loop = self._s_loop.torch
        # a caller that already works on the loop's stream (`with torch.cuda.stream(ctl.loop_stream)`) needs no hand-over: the two
        # stream waits below are four runtime calls, ~30 us of host time per iteration -- more than the iteration's two launches
        on_loop = caller.cuda_stream == self._s_loop.ptr
        if not on_loop:
            self._b.stream_wait_stream(loop, caller)
        # an iteration that does not solve is one library call: given the loop's stream explicitly (no stream context to enter)
        if not (self.fused and (self.k % self.k_mpc) != 0 and self._nonsolve_fast(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code, loop)):
            with torch.cuda.stream(loop):
                self._compute(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code)
        if not on_loop:
            self._b.stream_wait_stream(caller, loop)
        return self.result

    @property
    def loop_stream(self):
        """The stream the control loop's kernels run on in the asynchronous mode (a torch ExternalStream on the loop's compute
        units), None otherwise.  A caller that produces the inputs and consumes the Result under `with torch.cuda.stream(
        ctl.loop_stream)` saves compute() the hand-over between its own stream and this one."""
        return self._s_loop.torch if self.multiprocessing else None

    def _solve_async(self, plan, k):
        """scripts/MPC_Wrapper.py:150-180 (run_MPC_asynchronous): hand the planner outputs to the MPC and return."""
        torch = self._torch
        n = self._n_issued
        i = n % 3
        if self._adopted == i and self._pending:  # the loop still reads the buffer this solve will write: catch up
            self._adopt(self._pending[-1][0], wait=True)
        if n >= 3:
            # solve n-3 read these snapshot buffers and wrote this output buffer: it must have finished before the loop
            # stream overwrites them (free-running start-up, before anything has been adopted; a no-op afterwards).
            # This also bounds the backlog on the MPC stream to three solves.
            self._s_loop.torch.wait_event(self._ev_done[i])
        xs, fs = self._snap[i]
        xs.copy_(plan["xref"])
        fs.copy_(plan["fsteps"])
        self._ev_in[i].record(self._s_loop.torch)
        self._s_mpc.torch.wait_event(self._ev_in[i])
        with torch.cuda.stream(self._s_mpc.torch):
            self._b.mpc_solve(xs, fs, k, out=self._outs[i])
            self._ev_done[i].record(self._s_mpc.torch)
        self._pending.append((n, k))
        self._n_issued = n + 1
        if k > 2:
            # scripts/MPC_Wrapper.py:89-102: the result the loop keeps reading until the new one arrives is shifted one
            # horizon step (its first column is the force the WBC applies); before any adoption that is the default result
            self._b.mpc_result_shift(plan["gait"], self._mpc_out if self._mpc_out is not None else self._mpc_default)

    def _adopt(self, n, wait):
        i = n % 3
        if wait:
            self._s_loop.torch.wait_event(self._ev_done[i])
        self._mpc_out = self._outs[i]
        self._adopted = i
        self._pending = [p for p in self._pending if p[0] > n]

    def _poll(self, k):
        """scripts/MPC_Wrapper.py:106-120: take the newest finished result, if any."""
        ready = None
        for n, k0 in self._pending:
            if (self.mpc_lag is None and self._ev_done[n % 3].query()) or (self.mpc_lag is not None and k >= k0 + self.mpc_lag):
                ready = n
        if ready is not None:
            self._adopt(ready, wait=self.mpc_lag is not None)

    def _compute(self, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code):
        if self.fused:
            return self._compute_fused(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code)
        b, k = self._b, self.k
        self._st = st = b.controller_update_state(joy_v_ref, q_filt, v_filt, rpy, out=self._st)
        self._plan = plan = b.planner_step(k, st["q"], st["h_v"], st["v_ref"], joystick_code, out=self._plan)
        self._mpc_step(plan, k)
        self._wi = wi = b.controller_wbc_inputs(self.x_f_mpc, plan["xref"], plan["feet_pva"], st["v"], out=self._wi)
        fc = wi["feet_cmd"]
        self._wbc = w = b.wbc_compute(wi["q_wbc"], wi["b_v"], wi["f_cmd"], plan["contacts"], fc[0], fc[1], fc[2],
                                      out=self._wbc)
        self._res = b.controller_result(w["tau_ff"], w["qdes"], w["vdes"], q_filt, v_secu, out=self._res)
        self.result = Result(self._res["result"])
        self.error_flag = self._res["error_flag"]
        self.k += 1
        return self.result

    def _mpc_step(self, plan, k):
        """Solve every k_mpc-th iteration (synchronously or handed to the MPC stream) and pick the result to use."""
        if (k % self.k_mpc) == 0:
            if self.multiprocessing:
                self._solve_async(plan, k)
            else:
                self._mpc_out = self._b.mpc_solve(plan["xref"], plan["fsteps"], k, out=self._mpc_out)
        if self.multiprocessing:
            self._poll(k)
        if self._not_first_iter and self._mpc_out is not None:
            self.x_f_mpc = self._mpc_out
        else:
            self._not_first_iter = True
            self.x_f_mpc = self._mpc_default

    def _compute_fused(self, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code):
        b, k = self._b, self.k
        if (k % self.k_mpc) == 0:
            # this iteration hands new data to the MPC: which result the WBC uses is only known after that, so the
            # WBC targets are assembled by the separate entry point (one extra launch every k_mpc-th iteration)
            self._pre = p = b.control_pre(k, joy_v_ref, q_filt, v_filt, rpy, joystick_code, x_f_mpc=None, out=self._pre)
            self._mpc_step(p, k)
            b.controller_wbc_inputs(self.x_f_mpc, p["xref"], p["feet_pva"], p["v"], out=p)
        else:
            if self._nonsolve_fast(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code, None):
                return self.result
            self._pick_mpc_result(k)
            # nobody reads the MPC's inputs on an iteration that does not solve: fsteps / gait / most of xref are not produced
            self._pre = p = b.control_pre(k, joy_v_ref, q_filt, v_filt, rpy, joystick_code, x_f_mpc=self.x_f_mpc,
                                          out=self._pre, mpc_inputs=False)
        fc = p["feet_cmd"]
        self._post = w = b.wbc_compute_result(p["q_wbc"], p["b_v"], p["f_cmd"], p["contacts"], fc[0], fc[1], fc[2], q_filt,
                                              v_secu, out=self._post)
        if self._res is not w or self.result is None:  # (the same buffers every iteration: the views are made once)
            self._res = w
            self.result = Result(w["result"])
            self.error_flag = w["error_flag"]
        self.k += 1
        return self.result

    def _pick_mpc_result(self, k):
        """scripts/Controller.py:246-253 on an iteration that does not solve: the newest finished result, or the default one."""
        if self.multiprocessing:
            self._poll(k)
        if self._not_first_iter and self._mpc_out is not None:
            self.x_f_mpc = self._mpc_out
        else:
            self._not_first_iter = True
            self.x_f_mpc = self._mpc_default

    def _nonsolve_fast(self, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code, stream):
        """An iteration that does not solve, through the buffers bound in the library (qrw_hip.Batch.bind_iteration: the same two
        launches as control_pre + wbc_compute_result, one foreign call).  False until both calls have run once the ordinary way.
        Bound again whenever the caller comes with other tensor objects -- or the same objects on other storage -- than last time
        (a loop passes the same ones every tick; the check is one data_ptr() per input tensor)."""
        if self._pre is None or self._post is None or self.result is None:
            return False
        fast = self._fast
        ins = (joy_v_ref, q_filt, v_filt, rpy, v_secu)
        if (fast is None or fast[1] is not self._pre or fast[2] is not self._post or fast[3] is not stream
                or fast[4] is not joy_v_ref or fast[5] is not q_filt or fast[6] is not v_filt or fast[7] is not rpy
                or fast[8] is not v_secu
                or not (fast[9] is joystick_code or (isinstance(joystick_code, int) and fast[9] == joystick_code))
                # the same tensor OBJECTS on other storage (t.data = ..., set_(), resize_()): the bound pointers would be stale
                or fast[10] != tuple(t.data_ptr() for t in ins)
                or (fast[11] is not None and fast[11] != joystick_code.data_ptr())):
            step = self._b.bind_iteration(self._pre, self._post, (joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code), stream)
            self._fast = fast = (step, self._pre, self._post, stream, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code,
                                 tuple(t.data_ptr() for t in ins),
                                 joystick_code.data_ptr() if self._torch.is_tensor(joystick_code) else None)
        k = self.k
        self._pick_mpc_result(k)
        fast[0](k, self.x_f_mpc)
        self.k = k + 1
        return True

    def stop_parallel_loop(self):
        """scripts/MPC_Wrapper.py:300-306: drain the MPC stream and release both streams."""
        if self.multiprocessing:
            self._s_mpc.torch.synchronize()
            self._s_loop.torch.synchronize()
            self._s_mpc.close()
            self._s_loop.close()
            self.multiprocessing = False

    def stats(self):
        return dict(mpc=self._b.mpc_stats(), wbc=self._b.wbc_stats())


class Controller_groups(Controller_batch):
    """`Controller_batch(batch, ..., groups=G)`: the fleet as G independent stream groups, each a single-handle
    Controller_batch of batch / G robots on its own stream (qrw_hip.StreamGroups' stream pool).  Opt-in.

    Why: one handle steps the whole fleet with one launch per kernel, and the MPC launch ends with its longest solve while
    most of the chip is already idle; with two groups in flight one group's stragglers run beside the
    other group's work.  The robots are independent, so UNSTAGGERED every robot's results are those of the single handle,
    bit for bit; STAGGERED (stagger=True) every robot of group g gets the results of a single handle started lag_of(g) fleet
    ticks later, bit for bit, and holds q_init until then (tests/test_gpu_controller.py) -- the fleet's robots are then NOT
    at the same iteration on a given tick, and self.k follows group 0.

    compute() keeps the single-handle contract: whole-fleet inputs in, whole-fleet Result out, valid on the caller's stream
    (which is made to wait for every group: one join per iteration; measured slower than the single handle in the 1:10 loop
    unless staggered, see Controller_batch.__init__).  A caller whose own feedback is per robot as well can
    skip the join and let the groups run free: `compute_group(g, ...)` with that group's slices `slice_of(g)` on the stream
    `stream_of(g)` (what bench.py's 1:10 figure does)."""

    def __init__(self, batch, q_init, dt_wbc=0.002, dt_mpc=0.02, k_mpc=10, T_gait=0.32, T_mpc=0.32, N_gait=20,
                 h_ref=0.2229, device=0, multiprocessing=False, loop_cus=None, mpc_lag=None, fused=True, groups=None,
                 stagger=None, deadline=None, _out_views=None):
        import torch

        G = auto_groups(batch, groups, multiprocessing)
        stagger = bool(stagger)  # opt-in: it changes what the robots of the late groups do (see Controller_batch.__init__)
        if G < 2 or int(batch) % G:
            raise qrw_hip.QrwError("batch %d does not split into %d equal groups" % (batch, G))
        self._torch = torch
        self.B, self.G, self.Bs = int(batch), G, int(batch) // G
        self.dev = torch.device("cuda:%d" % device)
        self.k, self.k_mpc, self.h_ref, self.dt_wbc = 0, int(k_mpc), float(h_ref), float(dt_wbc)
        self.n_steps = int(round(T_mpc / dt_mpc))
        self.multiprocessing, self.fused = bool(multiprocessing), bool(fused)
        self._fleet_result = torch.zeros((self.B, 5, 12), dtype=torch.float64, device=self.dev)
        # stagger: group g starts g * k_mpc / G fleet ticks late, so that the groups solve on different ticks (the reference solves
        # on k % k_mpc == 0 of the robot's OWN clock, scripts/Controller.py:246-253: every robot of group g still sees exactly
        # the single-handle controller, started that many ticks later).  On the device the late group's first iteration waits
        # for group 0 to have reached the same tick, which is what takes the two groups' solves apart.
        self.stagger = bool(stagger)
        self._delay = [(g * int(k_mpc)) // G if self.stagger else 0 for g in range(G)]
        self.lag = tuple(self._delay)  # fleet ticks group g runs behind group 0 (and holds q_init for at the start)
        self._init_deadline(deadline)
        self._calls = [0] * G
        self._views = None
        self._tick_ev = {}
        self.error_flag = torch.zeros((self.B,), dtype=torch.int32, device=self.dev)
        self.result = Result(self._fleet_result)
        qi = np.broadcast_to(np.asarray(q_init, dtype=np.float64).reshape(-1, 12), (self.B, 12))
        self._sl = [slice(g * self.Bs, (g + 1) * self.Bs) for g in range(G)]
        # a group that starts late holds its initial posture until then: the gains the controller itself commands
        # (scripts/Controller.py:306-307) around q_init, which is what the reference's main loop does with a robot whose
        # controller has not started (scripts/main_solo12_control.py: PD on q_init before the loop)
        for g in range(G):
            if self._delay[g] > 0:
                hold = np.zeros((self.Bs, 5, 12))
                hold[:, 0], hold[:, 1], hold[:, 2] = 3.0, 0.2, qi[self._sl[g]]
                self._fleet_result[self._sl[g]] = torch.from_numpy(hold).to(self.dev)
        for g in range(G):
            if (device, g) not in qrw_hip.StreamGroups._streams:
                qrw_hip.StreamGroups._streams[(device, g)] = torch.cuda.Stream(self.dev)
        self.streams = [qrw_hip.StreamGroups._streams[(device, g)] for g in range(G)]
        self.groups = []
        for g, sl in enumerate(self._sl):
            with torch.cuda.stream(self.streams[g]):
                self.groups.append(Controller_batch(
                    self.Bs, qi[sl], dt_wbc=dt_wbc, dt_mpc=dt_mpc, k_mpc=k_mpc, T_gait=T_gait, T_mpc=T_mpc, N_gait=N_gait,
                    h_ref=h_ref, device=device, multiprocessing=multiprocessing, loop_cus=loop_cus, mpc_lag=mpc_lag, fused=fused,
                    groups=1, _out_views=dict(result=self._fleet_result[sl], error_flag=self.error_flag[sl])))

    @property
    def loop_stream(self):
        """None: every group brings its own streams (Controller_batch.loop_stream is a single handle's)."""
        return None

    def slice_of(self, g):
        return self._sl[g]

    def lag_of(self, g):
        """Fleet ticks the robots of group g run behind group 0 (0 unless staggered): at fleet tick t they return their own
        iteration t - lag_of(g), and hold q_init during the first lag_of(g) ticks."""
        return self._delay[g]

    def group_started(self, g):
        """False while a staggered group has not run its first iteration yet (its slice of the result still holds q_init)."""
        return self._calls[g] > self._delay[g]

    def stream_of(self, g):
        return self.streams[g]

    def compute_group(self, g, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code=0):
        """One control iteration (fleet tick) of group g alone: EVERY argument is that group's slice (contiguous, leading
        dimension batch / groups; a tensor joystick_code too), enqueued on the CURRENT stream -- call it under
        `torch.cuda.stream(ctl.stream_of(g))`, once per fleet tick and group.  With stagger=True the first
        g * k_mpc / groups calls of group g do nothing (its robots have not started yet: its slice of the result holds q_init
        with the controller's PD gains, its error flags are 0) and return that slice."""
        torch = self._torch
        t = self._calls[g]
        self._calls[g] = t + 1
        if g == 0:
            self.k = t + 1
        if t < self._delay[g]:
            return Result(self._fleet_result[self._sl[g]])
        if self.stagger and g > 0 and t == self._delay[g] and t in self._tick_ev:
            torch.cuda.current_stream(self.dev).wait_event(self._tick_ev[t])  # group 0 has finished its first t ticks
        r = self.groups[g].compute(joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code)
        if self.stagger and g == 0 and (t + 1) in self._delay[1:]:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._tick_ev[t + 1] = ev
        return r

    def _group_views(self, args):
        """The groups' slices of the fleet-wide argument tensors, made once per set of tensor OBJECTS: a loop passes the same ones on
        every tick, and the groups' bound iterations (Controller_batch._nonsolve_fast) recognise their inputs by identity."""
        c = self._views
        if c is None or len(c[0]) != len(args) or any(a is not b for a, b in zip(c[0], args)):
            torch = self._torch
            views = [tuple(a[sl] if torch.is_tensor(a) else a for a in args) for sl in self._sl]
            self._views = c = (args, views)  # (holds the arguments: an id cannot be reused while it is cached)
        return c[1]

    def _solves_now(self, g):
        """Will group g's next compute_group call carry an MPC solve (its robots' own clock at a multiple of k_mpc)?"""
        return self._calls[g] >= self._delay[g] and (self.groups[g].k % self.k_mpc) == 0

    def _compute_any(self, joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code=0):
        """compute(): one fleet tick, whole-fleet inputs in, whole-fleet Result out, valid on the caller's stream.  A group whose
        iteration does not solve runs on the caller's stream itself (nothing to overlap: two launches of ~25 us); a group whose
        iteration carries its MPC solve runs on the group's stream, forked from and joined to the caller's stream -- with
        staggered groups that is one group on two ticks out of k_mpc, and the fleet's worst tick takes a half-fleet solve instead
        of the whole fleet's."""
        torch = self._torch
        caller = torch.cuda.current_stream(self.dev)
        views = self._group_views((joy_v_ref, q_filt, v_filt, rpy, v_secu, joystick_code))
        if self.multiprocessing:
            forked = list(range(self.G))  # (asynchronous groups bring their own compute-unit-masked streams)
        else:
            forked = [g for g in range(self.G) if self._solves_now(g)]
            for g in range(self.G):
                if g not in forked:
                    self.compute_group(g, *views[g])
        b0 = self.groups[0]._b
        for g in forked:
            st = self.streams[g]
            b0.stream_wait_stream(st, caller)  # the inputs produced on the caller's stream (and this tick's other groups) are ready
            with torch.cuda.stream(st):
                self.compute_group(g, *views[g])
        for g in forked:
            b0.stream_wait_stream(caller, self.streams[g])
        return self.result

    def stop_parallel_loop(self):
        for c in self.groups:
            c.stop_parallel_loop()
        self.multiprocessing = False

    def stats(self):
        st = [c.stats() for c in self.groups]
        return {k: {kk: np.concatenate([s_[k][kk] for s_ in st]) for kk in st[0][k]} for k in st[0]}
