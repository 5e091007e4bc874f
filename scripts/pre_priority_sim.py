"""Offline study behind the priority levels of the time-sliced MPC launch (mpc_kernel.hip, PRE; DESIGN.md 4.1 "Time slicing").

Data: profiles/r3_res_trace_n32_mixed.npz (scripts/gpu_res_trace.py on the diagnostic build `make trace`): for 7 consecutive
calls of BASELINE config 4's workload (batch 4096, N = 32, walk / trot / bounding) every solve's iteration count and, at each
adaptive-rho test (every 200 iterations), primal residual / its tolerance and dual residual / its tolerance.

ADMM converges linearly between rho updates, so with r = max of the two ratios the remaining iterations at test t are about
200 ln r(t) / ln(r(t - 200) / r(t)).  The script (1) measures that predictor against the recorded counts and (2) list-schedules
the recorded solves on the 512 resident slots for the policies that were considered: one FIFO (round robin), strict priority
levels by predicted remaining iterations (what the kernel does: the first park of a solve goes to a FIFO level of its own, later
ones to level 1 + (levels - 2 - remaining / bin)), the same with the true remaining count (the bound of any predictor), and the
"no queue change" variant that merely keeps predicted-long solves running.  Makespans are relative to work / slots."""
import collections
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLOTS, OV = 512, 12.0  # resident two-wavefront instances; set-up / resume cost of a slice in iteration-times


def load(path=None):
    d = np.load(path or os.path.join(ROOT, "profiles", "r3_res_trace_n32_mixed.npz"))
    its = d["iters"].astype(int)
    R = d["ratios"].astype(float).max(-1)  # [S][B][20]: r at iteration 200 (c + 1)
    return its, R


def predict(R, s, b, t, its=None, mode="residuals"):
    """remaining iterations of solve (s, b) parked at iteration t (a multiple of 200)"""
    if mode == "true":
        return its[s, b] - t
    c = t // 200 - 1
    r1 = R[s, b, c]
    r0 = R[s, b, c - 1] if c > 0 else 0.0
    rem = 4000.0 - t
    if r0 > r1 > 1.0:
        rem = min(rem, 200.0 * np.log(r1) / np.log(r0 / r1))
    return rem


def lower_bound(it):
    w = it + OV
    return max(w.max(), w.sum() / SLOTS)


def simulate(its, R, s, first_order, chunk, levels, lbin, mode="residuals", keep_running_above=None):
    """levels = 1: one FIFO.  levels > 1: level 0 = first parks (FIFO), levels 1.. by predicted remaining (most first).
    keep_running_above: single FIFO, but a solve whose predicted remaining exceeds this is not parked at all."""
    it = its[s]
    running, free, t = [], SLOTS, 0.0
    pending = list(first_order)[::-1]
    Q = [collections.deque() for _ in range(levels)]
    nrun = collections.Counter()
    finish, slices = 0.0, 0

    def start(b, att, now, length, overhead):
        nonlocal free, slices
        run = min(it[b] - att, length)
        slices += overhead > 0
        heapq.heappush(running, (now + overhead + run, b, att + run))
        free -= 1

    while True:
        while free > 0:
            if pending:
                b = pending.pop(); nrun[b] = 1; start(b, 0, t, chunk, OV); continue
            for q in Q:
                if q:
                    b, att = q.popleft(); nrun[b] += 1; start(b, att, t, chunk, OV); break
            else:
                break
        if not running:
            break
        end, b, att = heapq.heappop(running)
        t = end
        free += 1
        if att >= it[b]:
            finish = max(finish, t)
            continue
        if keep_running_above is not None:
            if att >= 2 * chunk and predict(R, s, b, att, its, mode) > keep_running_above:
                start(b, att, t, 200, 0.0)
            else:
                Q[0].append((b, att))
        elif levels == 1 or nrun[b] == 1:
            Q[0].append((b, att))
        else:
            nl = levels - 1
            rem = predict(R, s, b, att, its, mode)
            Q[1 + (nl - 1 - min(nl - 1, max(0, int(rem / lbin))))].append((b, att))
    return finish / lower_bound(it), slices / len(it)


def predictor_quality(its, R):
    rows = []
    for t in (400, 600, 800, 1200, 1800):
        e = []
        for s in range(its.shape[0]):
            alive = np.where(its[s] > t)[0]
            p = np.array([predict(R, s, b, t) for b in alive])
            e.append(np.log((p + 25.0) / (its[s, alive] - t + 25.0)))
        e = np.concatenate(e)
        rows.append((t, e.mean(), e.std(), np.percentile(e, 0.1), np.percentile(e, 1)))
    return rows


def policies(its, R):
    S, B = its.shape
    out = {}
    ema = its[0].astype(float)
    for s in range(1, S):
        order = np.argsort(-ema, kind="stable")  # the shipped first-pass order: moving average of the previous counts
        cases = {
            "plain launch (no slicing)": dict(chunk=4000, levels=1, lbin=400),
            "one FIFO, slices of 600 (round robin)": dict(chunk=600, levels=1, lbin=400),
            "9 levels of 400, slices of 600": dict(chunk=600, levels=9, lbin=400),
            "9 levels of 200, slices of 600 (shipped)": dict(chunk=600, levels=9, lbin=200),
            "5 levels of 800, slices of 600": dict(chunk=600, levels=5, lbin=800),
            "9 levels of 400, slices of 400": dict(chunk=400, levels=9, lbin=400),
            "9 levels of 400, TRUE remaining count": dict(chunk=600, levels=9, lbin=400, mode="true"),
            "one FIFO, predicted > 1600 keeps running": dict(chunk=600, levels=1, lbin=400, keep_running_above=1600),
        }
        for name, kw in cases.items():
            out.setdefault(name, []).append(simulate(its, R, s, order, **kw))
        ema += (its[s] - ema) * 0.125
    return out


if __name__ == "__main__":
    its, R = load(sys.argv[1] if len(sys.argv) > 1 else None)
    print("%d calls x %d solves: mean %.0f iterations, %.1f %% at max_iter" % (its.shape[0], its.shape[1], its.mean(), 100 * (its >= 4000).mean()))
    print("predictor (log of predicted / true remaining; +25 iterations on both):")
    for t, m, sd, q001, q01 in predictor_quality(its, R):
        print("  at iteration %4d: mean %+.3f  std %.3f  0.1 %% quantile %+.2f  1 %% quantile %+.2f" % (t, m, sd, q001, q01))
    print("launch time / (work / 512 slots), mean and worst of %d calls; slices per solve:" % (its.shape[0] - 1))
    for name, v in policies(its, R).items():
        print("  %-44s %.3f  %.3f  %.2f" % (name, np.mean([x[0] for x in v]), np.max([x[0] for x in v]), np.mean([x[1] for x in v])))
