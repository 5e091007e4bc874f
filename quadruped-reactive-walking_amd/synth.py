"""Synthetic Solo12 control-loop inputs for tests and bench (host side, numpy).

Recipe: SURVEY.md §8(d).  Per instance b the RNG seed is 20260000 + b.  The shapes and
formulas follow the reference components that produce the hot path's inputs:
  - gait rows            src/Gait.cpp:38-108 (walk / trot / pacing / bounding / static)
  - reference trajectory src/StatePlanner.cpp:21-61 (computeReferenceStates)
  - footstep matrix      src/FootstepPlanner.cpp:76-186 (computeFootsteps / computeNextFootstep)
  - velocity ranges      scripts/Joystick.py:201-284
  - WBC call assembly    scripts/Controller.py:275-303
This is an input generator, not a re-implementation of the planners: it keeps no
persistent footstep state (feet in stance at row 0 are placed analytically).
"""
import numpy as np

SHOULDERS = np.array([[0.1946, 0.1946, -0.1946, -0.1946],
                      [0.14695, -0.14695, 0.14695, -0.14695],
                      [0.0, 0.0, 0.0, 0.0]])  # scripts/Controller.py:131-133
Q_NOMINAL = np.array([0.0, 0.7, -1.4, 0.0, 0.7, -1.4, 0.0, -0.7, 1.4, 0.0, -0.7, 1.4])  # scripts/test_mpc.py:40
H_REF = 0.2229  # 0.32*cos(0.8), scripts/Estimator.py:245
GAIT_KINDS = ("trot", "walk", "bounding", "pacing", "static")


def gait_pattern(kind, n_period):
    """One period of contact rows (n_period x 4), src/Gait.cpp:38-108."""
    P = np.zeros((n_period, 4))
    if kind == "trot":
        h = n_period // 2
        P[:h] = [1, 0, 0, 1]
        P[h:] = [0, 1, 1, 0]
    elif kind == "pacing":
        h = n_period // 2
        P[:h] = [1, 0, 1, 0]
        P[h:] = [0, 1, 0, 1]
    elif kind == "bounding":
        h = n_period // 2
        P[:h] = [1, 1, 0, 0]
        P[h:] = [0, 0, 1, 1]
    elif kind == "walk":
        qn = n_period // 4
        seqs = ([0, 1, 1, 1], [1, 0, 1, 1], [1, 1, 0, 1], [1, 1, 1, 0])
        for i, s in enumerate(seqs):
            P[i * qn:(i + 1) * qn] = s
        P[4 * qn:] = seqs[3]
    elif kind == "static":
        P[:] = 1
    else:
        raise ValueError(kind)
    return P


def reference_states(x0, vref, n_steps, dt, h_ref=H_REF):
    """xref (B,12,N+1): column 0 = current state, columns 1.. per src/StatePlanner.cpp:35-60."""
    x0 = np.atleast_2d(x0)
    vref = np.atleast_2d(vref)
    B = x0.shape[0]
    xref = np.zeros((B, 12, n_steps + 1))
    xref[:, :, 0] = x0
    t = dt * np.arange(1, n_steps + 1)[None, :]
    vx, vy, wz = vref[:, 0:1], vref[:, 1:2], vref[:, 5:6]
    nz = (wz != 0)
    wsafe = np.where(nz, wz, 1.0)
    xs = np.where(nz, (vx * np.sin(wz * t) + vy * (np.cos(wz * t) - 1.0)) / wsafe, vx * t)
    ys = np.where(nz, (vy * np.sin(wz * t) - vx * (np.cos(wz * t) - 1.0)) / wsafe, vy * t)
    xref[:, 0, 1:] = xs + x0[:, 0:1]
    xref[:, 1, 1:] = ys + x0[:, 1:2]
    xref[:, 2, 1:] = h_ref
    yaw = wz * t
    xref[:, 5, 1:] = yaw
    xref[:, 6, 1:] = vx * np.cos(yaw) - vy * np.sin(yaw)
    xref[:, 7, 1:] = vx * np.sin(yaw) + vy * np.cos(yaw)
    xref[:, 11, 1:] = wz
    return xref


def footsteps(gait, v, vref, n_steps, dt, progress0, h_ref=H_REF, k_feedback=0.03, L=0.155, g=9.81):
    """fsteps (B,N_gait,12) from gait (B,N_gait,4): zero for swing feet and beyond the horizon.

    Future touch-downs follow src/FootstepPlanner.cpp:113-186; a foot already in stance at row 0
    is placed where a foot `progress0` (B,4) through its stance would be (in the base frame).
    """
    B, N_gait, _ = gait.shape
    fs = np.zeros((B, N_gait, 12))
    # stance duration per foot: count of 1s in a period (src/Gait.cpp getPhaseDuration semantics)
    t_stance = dt * gait[:, :n_steps, :].sum(axis=1)  # (B,4)
    t_stance = np.where(t_stance <= 0, dt * n_steps, t_stance)
    t_stance = np.minimum(t_stance, dt * n_steps)
    cross = np.stack([v[:, 1] * vref[:, 5] - v[:, 2] * vref[:, 4], v[:, 2] * vref[:, 3] - v[:, 0] * vref[:, 5]], 1)
    dt_cum = dt * np.arange(N_gait)[None, :] * np.ones((B, 1))
    wz = vref[:, 5:6]
    nz = (wz != 0)
    wsafe = np.where(nz, wz, 1.0)
    dx = np.where(nz, (v[:, 0:1] * np.sin(wz * dt_cum) + v[:, 1:2] * (np.cos(wz * dt_cum) - 1.0)) / wsafe,
                  v[:, 0:1] * dt_cum)
    dy = np.where(nz, (v[:, 1:2] * np.sin(wz * dt_cum) - v[:, 0:1] * (np.cos(wz * dt_cum) - 1.0)) / wsafe,
                  v[:, 1:2] * dt_cum)
    yaws = wz * dt_cum
    for j in range(4):
        nxt = np.zeros((B, 2))
        for c in range(2):
            off = t_stance[:, j] * 0.5 * v[:, c] + k_feedback * (v[:, c] - vref[:, c]) \
                + 0.5 * np.sqrt(h_ref / g) * cross[:, c]
            nxt[:, c] = np.clip(off, -L, L) + SHOULDERS[c, j]
        cur = np.zeros((B, 2))
        for c in range(2):
            cur[:, c] = SHOULDERS[c, j] + (0.5 - progress0[:, j]) * t_stance[:, j] * v[:, c]
        pos = np.zeros((B, 2))
        st0 = gait[:, 0, j] > 0
        pos[st0] = cur[st0]
        fs[:, 0, 3 * j:3 * j + 2] = np.where(st0[:, None], pos, 0.0)
        for i in range(1, n_steps):
            st = gait[:, i, j] > 0
            was = gait[:, i - 1, j] > 0
            new = st & ~was
            c_, s_ = np.cos(yaws[:, i - 1]), np.sin(yaws[:, i - 1])
            px = c_ * nxt[:, 0] - s_ * nxt[:, 1] + dx[:, i - 1]
            py = s_ * nxt[:, 0] + c_ * nxt[:, 1] + dy[:, i - 1]
            pos = np.where(new[:, None], np.stack([px, py], 1), pos)
            fs[:, i, 3 * j:3 * j + 2] = np.where(st[:, None], pos, 0.0)
    # a stance foot whose x happens to be exactly 0 would read as swing (src/MPC.cpp:691)
    x = fs[:, :, 0::3]
    fs[:, :, 0::3] = np.where((gait > 0) & (x == 0.0), 1e-9, x)
    return fs


def leg_fk(qj):
    """Foot positions (B,4,3) of the fixed-base Solo12 (constants of include/qrw_solo12_model.h)."""
    qj = np.atleast_2d(qj)
    B = qj.shape[0]
    out = np.zeros((B, 4, 3))
    sx = [1, 1, -1, -1]
    sy = [1, -1, 1, -1]
    for leg in range(4):
        q0, q1, q2 = qj[:, 3 * leg], qj[:, 3 * leg + 1], qj[:, 3 * leg + 2]
        foot = np.array([0.0, sy[leg] * 0.008, -0.16])
        kfe = np.array([0.0, sy[leg] * 0.03745, -0.16])
        hfe = np.array([0.0, sy[leg] * 0.014, 0.0])
        haa = np.array([sx[leg] * 0.1946, sy[leg] * 0.0875, 0.0])

        def ry(a, p):
            return np.stack([np.cos(a) * p[..., 0] + np.sin(a) * p[..., 2], p[..., 1] + 0 * a,
                             -np.sin(a) * p[..., 0] + np.cos(a) * p[..., 2]], -1)

        def rx(a, p):
            return np.stack([p[..., 0] + 0 * a, np.cos(a) * p[..., 1] - np.sin(a) * p[..., 2],
                             np.sin(a) * p[..., 1] + np.cos(a) * p[..., 2]], -1)

        p = ry(q2, np.broadcast_to(foot, (B, 3))) + kfe
        p = ry(q1, p) + hfe
        p = rx(q0, p) + haa
        out[:, leg] = p
    return out


class SyntheticBatch:
    """Deterministic batched input sequences.

    step(s) returns a dict with the MPC inputs of receding-horizon call s
    (xref (B,12,N+1), fsteps (B,N_gait,12), gait (B,N_gait,4)) and the WBC inputs of the
    same control step (q (B,19), dq (B,18), contacts (B,4), pgoals/vgoals/agoals (B,3,4)).
    x0 may be supplied to close the loop with the MPC's predicted next state
    (scripts/test_mpc.py:78); otherwise the state follows the reference plus noise.
    """

    def __init__(self, B, n_steps=16, N_gait=20, dt=0.02, gaits=("trot",), n_seq=64, seed0=20260000, b0=0):
        self.B, self.N, self.N_gait, self.dt = B, n_steps, N_gait, dt
        self.n_seq = n_seq
        self.vref = np.zeros((B, 6))
        self.phase = np.zeros(B, dtype=np.int64)
        self.kind = np.zeros(B, dtype=np.int64)
        self.noise_x0 = np.zeros((n_seq, B, 12))
        self.noise_q = np.zeros((n_seq, B, 12))
        self.noise_dq = np.zeros((n_seq, B, 12))
        self.noise_goal = np.zeros((n_seq, B, 3, 4, 3))
        self.gaits = tuple(gaits)
        for b in range(B):
            r = np.random.default_rng(seed0 + b0 + b)
            self.vref[b, 0] = r.uniform(-0.5, 1.5)
            self.vref[b, 1] = r.uniform(-0.5, 0.5)
            self.vref[b, 5] = r.uniform(-0.7, 0.7)
            self.phase[b] = r.integers(0, n_steps)
            self.kind[b] = r.integers(0, len(self.gaits))
            nx = r.uniform(-1, 1, (n_seq, 12))
            nx *= np.array([0, 0, 0.01, 0.05, 0.05, 0, 0.1, 0.1, 0.1, 0.2, 0.2, 0.2])
            self.noise_x0[:, b] = nx
            self.noise_q[:, b] = r.uniform(-0.1, 0.1, (n_seq, 12))
            self.noise_dq[:, b] = r.uniform(-1.0, 1.0, (n_seq, 12))
            self.noise_goal[:, b] = r.uniform(-1.0, 1.0, (n_seq, 3, 4, 3))
        self.patterns = [gait_pattern(k, n_steps) for k in self.gaits]

    def gait_at(self, s):
        B, N = self.B, self.N
        gait = np.zeros((B, self.N_gait, 4))
        rows = (self.phase[:, None] + s + np.arange(N)[None, :]) % N
        for ki, P in enumerate(self.patterns):
            m = self.kind == ki
            if m.any():
                gait[m, :N] = P[rows[m]]
        return gait

    def progress_at(self, s, gait):
        """fraction of its stance each row-0 stance foot has completed (0 for swing feet)."""
        B, N = self.B, self.N
        prog = np.zeros((B, 4))
        for ki, P in enumerate(self.patterns):
            m = np.where(self.kind == ki)[0]
            if m.size == 0:
                continue
            for j in range(4):
                col = P[:, j]
                # for each row of the period: (#consecutive stance rows before it, stance length)
                before = np.zeros(N)
                length = np.zeros(N)
                for r in range(N):
                    if col[r] > 0:
                        k = 0
                        while k < N and col[(r - k - 1) % N] > 0:
                            k += 1
                        a = 0
                        while a < N and col[(r + a) % N] > 0:
                            a += 1
                        before[r], length[r] = min(k, N), min(k + a, N)
                r0 = (self.phase[m] + s) % N
                prog[m, j] = np.where(length[r0] > 0, before[r0] / np.maximum(length[r0], 1), 0.0)
        return prog

    def step(self, s, x0=None):
        B, N, dt = self.B, self.N, self.dt
        i = s % self.n_seq
        base = np.zeros((B, 12))
        base[:, 2] = H_REF
        base[:, 6:9] = self.vref[:, 0:3]
        base[:, 9:12] = self.vref[:, 3:6]
        if x0 is None:
            x0 = base + self.noise_x0[i]
        else:
            x0 = np.array(x0, dtype=np.float64).reshape(B, 12).copy()
            x0[:, 0:2] = 0.0  # horizontal frame, src/StatePlanner.cpp:27-31
            x0[:, 5] = 0.0
        xref = reference_states(x0, self.vref, N, dt)
        gait = self.gait_at(s)
        fsteps = footsteps(gait, x0[:, 6:12], self.vref, N, dt, self.progress_at(s, gait))
        # WBC inputs (scripts/Controller.py:275-303 shapes)
        q = np.zeros((B, 19))
        q[:, 2] = H_REF
        q[:, 6] = 1.0
        q[:, 7:] = Q_NOMINAL + self.noise_q[i]
        dq = np.zeros((B, 18))
        dq[:, :6] = self.vref
        dq[:, 6:] = self.noise_dq[i]
        contacts = gait[:, 0, :].copy()
        feet = leg_fk(q[:, 7:])  # (B,4,3)
        swing = (contacts == 0)[:, None, :]  # (B,1,4)
        ng = self.noise_goal[i]
        pgoals = np.transpose(feet, (0, 2, 1)) + 0.01 * ng[:, 0].transpose(0, 2, 1)
        vgoals = np.where(swing, 0.1 * ng[:, 1].transpose(0, 2, 1), 0.0)
        agoals = np.where(swing, 1.0 * ng[:, 2].transpose(0, 2, 1), 0.0)
        return dict(xref=xref, fsteps=fsteps, gait=gait, q=q, dq=dq, contacts=contacts, pgoals=pgoals,
                    vgoals=vgoals, agoals=agoals, x0=x0)


STANCE_SETS = np.array([[(m >> i) & 1 for i in range(4)] for m in range(1, 16)], dtype=np.float64)  # the 15 non-empty ones


class RandomContactTables:
    """Seeded MPC inputs with ARBITRARY contact tables, for parity tests and soaks.

    MPC::construct_gait / update_ML accept any 0/1 table the footstep matrix encodes (src/MPC.cpp:418-464,686-701), not
    only the five periodic gaits of src/Gait.cpp that SyntheticBatch produces.  Per instance (seed = seed0 + b) and call:
      - table length anywhere in 1..n_steps (about a third of the tables full length; N_gait == n_steps gives the table
        without a terminating zero row), the rows after it all zero;
      - rows drawn in runs of 1..6 from the 15 non-empty stance sets in arbitrary order (single-stance rows included);
      - footholds up to +-`reach` m from the shoulders in x and y (held over a foot's stance run, re-drawn in a fifth
        of the rows), z = 0 mostly, a few centimetres otherwise; a stance foot's x is never exactly 0 (it would read
        as swing, src/MPC.cpp:691);
      - current state with roll / pitch up to +-0.3 rad, yaw over +-pi, velocities up to +-1.5 m/s and +-1.5 rad/s; the
        reference trajectory turns and advances at a random constant rate from there (src/StatePlanner.cpp:35-60 shape);
      - between two calls the table either recedes by one row (a new row appended, as the planner does) or, with
        probability `p_change`, is drawn afresh: every B block and S flag changes under a warm-started solver.
    step(c) -> dict(xref (B,12,N+1), fsteps (B,N_gait,12), gait (B,N_gait,4)); deterministic in (seed0, b, c) only if
    calls are made in order c = 0, 1, 2, ... (the receding tables carry state)."""

    def __init__(self, B, n_steps=16, N_gait=None, dt=0.02, seed0=20600000, reach=0.3, p_change=0.4, b0=0):
        self.B, self.N, self.dt = B, n_steps, dt
        self.N_gait = N_gait or max(20, n_steps + 4)
        assert self.N_gait >= n_steps
        self.reach, self.p_change = reach, p_change
        self.rng = [np.random.default_rng(seed0 + b0 + b) for b in range(B)]
        self.rows = [None] * B   # (L,4) stance sets of the current table
        self.feet = [None] * B   # (L,4,3) footholds
        self.next_call = 0

    def _draw_row_run(self, r, n, prev_row=None, prev_feet=None):
        """n more rows: runs of equal stance sets; a foot keeps its foothold while it stays in stance."""
        rows, feet = [], []
        row, ft = prev_row, prev_feet
        left = 0
        for _ in range(n):
            if left == 0 or row is None:
                new = STANCE_SETS[r.integers(0, 15)]
                left = int(r.integers(1, 7))
                nf = np.zeros((4, 3)) if ft is None else ft.copy()
                for j in range(4):
                    if new[j] > 0 and (row is None or row[j] == 0):
                        nf[j] = self._foothold(r, j)
                row, ft = new, nf
            elif r.random() < 0.2:
                ft = ft.copy()
                j = int(r.integers(0, 4))
                if row[j] > 0:
                    ft[j] = self._foothold(r, j)
            left -= 1
            rows.append(row)
            feet.append(ft)
        return rows, feet

    def _foothold(self, r, j):
        p = np.array([SHOULDERS[0, j] + r.uniform(-self.reach, self.reach), SHOULDERS[1, j] + r.uniform(-self.reach, self.reach),
                      0.0 if r.random() < 0.7 else r.uniform(-0.03, 0.03)])
        if p[0] == 0.0:
            p[0] = 1e-9
        return p

    def _fresh_table(self, r):
        N = self.N
        L = N if (self.N_gait == N or r.random() < 0.35) else int(r.integers(1, N + 1))
        rows, feet = self._draw_row_run(r, L)
        return rows, feet

    def step(self, c):
        assert c == self.next_call, "RandomContactTables.step must be called with c = 0, 1, 2, ..."
        self.next_call += 1
        B, N, dt = self.B, self.N, self.dt
        xref = np.zeros((B, 12, N + 1))
        fsteps = np.zeros((B, self.N_gait, 12))
        gait = np.zeros((B, self.N_gait, 4))
        t = dt * np.arange(1, N + 1)
        for b in range(B):
            r = self.rng[b]
            if self.rows[b] is None or r.random() < self.p_change:
                self.rows[b], self.feet[b] = self._fresh_table(r)
            else:  # recede by one row; the table keeps its length (or grows / shrinks by one now and then)
                rows, feet = self.rows[b][1:], self.feet[b][1:]
                want = len(self.rows[b]) + int(r.integers(-1, 2))
                want = N if self.N_gait == N else min(max(want, 1), N)
                if len(rows) < want:
                    more_r, more_f = self._draw_row_run(r, want - len(rows), rows[-1] if rows else None,
                                                        feet[-1] if feet else None)
                    rows, feet = rows + more_r, feet + more_f
                self.rows[b], self.feet[b] = rows[:want], feet[:want]
            L = len(self.rows[b])
            R = np.array(self.rows[b])
            F = np.array(self.feet[b])
            gait[b, :L] = R
            fsteps[b, :L] = (F * R[:, :, None]).reshape(L, 12)
            # current state and reference trajectory
            x0 = np.zeros(12)
            x0[0:2] = r.uniform(-0.05, 0.05, 2)
            x0[2] = H_REF + r.uniform(-0.04, 0.04)
            x0[3:5] = r.uniform(-0.3, 0.3, 2)
            x0[5] = r.uniform(-np.pi, np.pi)
            x0[6:9] = r.uniform(-1.5, 1.5, 3) * np.array([1.0, 1.0, 0.3])
            x0[9:12] = r.uniform(-1.5, 1.5, 3)
            v = r.uniform(-1.0, 1.5, 2) * np.array([1.0, 0.4])
            wz = r.uniform(-1.0, 1.0) if r.random() < 0.8 else 0.0
            yaw = x0[5] + wz * t
            xref[b, :, 0] = x0
            xref[b, 5, 1:] = yaw
            xref[b, 6, 1:] = v[0] * np.cos(yaw) - v[1] * np.sin(yaw)
            xref[b, 7, 1:] = v[0] * np.sin(yaw) + v[1] * np.cos(yaw)
            xref[b, 0, 1:] = x0[0] + np.cumsum(xref[b, 6, 1:]) * dt
            xref[b, 1, 1:] = x0[1] + np.cumsum(xref[b, 7, 1:]) * dt
            xref[b, 2, 1:] = H_REF
            xref[b, 11, 1:] = wz
        return dict(xref=xref, fsteps=fsteps, gait=gait)


class RandomWbcInputs:
    """Seeded WBC inputs far outside what the periodic-gait controller produces, for parity tests and soaks of the whole-body
    step (scripts/QP_WBC.py:52-131 takes ANY q / dq / f_cmd / contacts / goals): arbitrary base orientation (a random unit
    quaternion in a fifth of the calls, up to +-0.6 rad about a random axis otherwise), base offsets, joint angles +-0.6 rad
    around the nominal pose (knees kept >= 0.25 rad from straight: the InvKin's 3x3 foot Jacobian is singular there, as in the
    reference), joint velocities up to 6 rad/s, a base twist up to 1.5; contact sets that persist and flip at random (all 16
    occur, so k_since_contact and its 1/16-ramp of the foot tracking gains move); commanded forces that VIOLATE the QP's bounds
    in a third of the stance feet (f_z up to 32 N > 25, tangential up to 1.3 f_z > mu f_z), occasionally non-zero on swing feet;
    foot goals up to 5 cm / 1 m/s / 10 m/s^2 from the feet.  step(c) -> dict(q, dq, f_cmd, contacts, pgoals, vgoals, agoals)."""

    def __init__(self, B, seed0=20700000, b0=0):
        self.B = B
        self.rng = [np.random.default_rng(seed0 + b0 + b) for b in range(B)]
        self.contacts = np.ones((B, 4))

    def step(self, c):
        B = self.B
        q, dq = np.zeros((B, 19)), np.zeros((B, 18))
        f, pg, vg, ag = np.zeros((B, 12)), np.zeros((B, 3, 4)), np.zeros((B, 3, 4)), np.zeros((B, 3, 4))
        for b in range(B):
            r = self.rng[b]
            flip = r.random(4) < (1.0 if c == 0 else 0.3)
            new = (r.random(4) < 0.6).astype(np.float64)
            self.contacts[b] = np.where(flip, new, self.contacts[b])
            q[b, :3] = r.uniform(-0.3, 0.3, 3) + np.array([0, 0, H_REF])
            if r.random() < 0.2:
                quat = r.normal(size=4)
            else:
                axis = r.normal(size=3)
                ang = r.uniform(-0.6, 0.6)
                quat = np.concatenate([np.sin(ang / 2) * axis / np.linalg.norm(axis), [np.cos(ang / 2)]])
            q[b, 3:7] = quat / np.linalg.norm(quat)
            qj = Q_NOMINAL + r.uniform(-0.6, 0.6, 12)
            knee = qj[2::3]
            qj[2::3] = np.where(np.abs(knee) < 0.25, np.sign(Q_NOMINAL[2::3]) * 0.25, knee)
            q[b, 7:] = qj
            dq[b, :6] = r.uniform(-1.5, 1.5, 6)
            dq[b, 6:] = r.uniform(-6.0, 6.0, 12)
            ct = self.contacts[b]
            fz = r.uniform(0.0, 24.0, 4)
            viol = r.random(4) < 0.33
            fz = np.where(viol, r.uniform(20.0, 32.0, 4), fz)
            tang = np.where(viol, 1.3, 0.6)
            fx, fy = r.uniform(-1, 1, 4) * tang * fz, r.uniform(-1, 1, 4) * tang * fz
            on = ct if r.random() < 0.9 else np.ones(4)  # now and then forces on swing feet too
            f[b, 0::3], f[b, 1::3], f[b, 2::3] = fx * on, fy * on, fz * on
            feet = leg_fk(qj[None])[0]  # (4,3) in the base frame of the fixed-base model
            pg[b] = feet.T + r.uniform(-0.05, 0.05, (3, 4))
            vg[b] = r.uniform(-1.0, 1.0, (3, 4)) * (1 - ct)[None, :]
            ag[b] = r.uniform(-10.0, 10.0, (3, 4)) * (1 - ct)[None, :]
        return dict(q=q, dq=dq, f_cmd=f, contacts=self.contacts.copy(), pgoals=pg, vgoals=vg, agoals=ag)
