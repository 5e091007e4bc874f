"""Known-answer scenarios the reference itself holds for the MPC path, restated for the current API.

/root/reference/scripts/test_mpc.py is written against an older API: the planner carried a COMPRESSED gait
(`gait[r] = [duration, c_FL, c_FR, c_HL, c_HR]`, `fsteps[r] = [duration, 12 foot coordinates]`, :42-49) and the MPC
expanded it; today's `MPC::run` takes the expanded table (`fsteps` N_gait x 12, row i = horizon step i,
src/MPC.cpp:626,686-701).  The scenarios below are the reference's inputs (:136-147, :165-179), its receding-horizon
update (`roll`, :96-133, re-expressed on the compressed rows and expanded row by row) and its acceptance criterion
(state of the first predicted step within 1e-2 of the reference state, :160,:190).  They are DATA + a property,
not code of the reference: the helper is shared by the oracle test (CPU) and the HIP test (GPU).
"""
import numpy as np

N = 16
N_GAIT = 20
H_REF = 0.24474949993103629  # scripts/test_mpc.py:39,139
PAIR_1 = np.array([0.195, 0.147, 0., 0., 0., 0., 0., 0., 0., -0.195, -0.147, 0.])  # FL + HR in stance (:137)
PAIR_2 = np.array([0., 0., 0., 0.195, -0.147, 0., -0.195, 0.147, 0., 0., 0., 0.])  # FR + HL in stance (:138)
NOT_CENTERED = np.array([0.05, 0.05, 0.2, 0.1, 0.1, 0.1, 0.01, 0.01, 0.04, 0.4, 0.4, 0.4])  # :179


class CompressedTrot:
    """Three compressed rows [7 x pair_1, 8 x pair_2, 1 x pair_1] (:141-147) and the reference's roll (:96-133)."""

    def __init__(self):
        self.dur = [7, 8, 1]
        self.rows = [PAIR_1.copy(), PAIR_2.copy(), PAIR_1.copy()]

    def roll(self):
        d, r = self.dur, self.rows
        if d[0] == 1:  # first phase ends: rows move up, the third one is emptied
            d[2] += 1
            d[0], d[1], d[2] = d[1], d[2], 0
            r[0], r[1], r[2] = r[1], r[2], np.zeros(12)
        elif d[0] == 8:  # a full phase starts to be consumed: its continuation appears at the end of the horizon
            d[0] -= 1
            d[2] = 1
            r[2] = r[0].copy()
        else:
            d[0] -= 1
            d[2] += 1

    def fsteps(self):
        out = np.zeros((N_GAIT, 12))
        k = 0
        for dur, row in zip(self.dur, self.rows):
            out[k:k + dur] = row
            k += dur
        assert k == N
        return out


def run_twostance(solve, calls, centered):
    """Drive `solve(i, xref, fsteps) -> x_f (24 x N)` exactly as test_twostance_(not_)centered does.
    Returns (last result, xref) so the caller can apply the reference's assertion."""
    xref = np.zeros((12, N + 1))
    xref[2, :] = H_REF
    if not centered:
        xref[:, 0] = NOT_CENTERED
    plan = CompressedTrot()
    x_f = None
    for i in range(calls):
        x_f = solve(i, xref.copy(), plan.fsteps())
        plan.roll()
        if i > 0:
            xref[:, 0] = x_f[:12, 0]
    return x_f, xref
