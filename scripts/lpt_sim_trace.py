"""Offline study of the block order of one N = 32 launch (batch 4096, walk / trot / bounding, open-loop noisy states) on the
recorded iteration counts (profiles/r3_iter_trace_n32_mixed.npy, scripts/gpu_iter_trace.py): list scheduling of the
instances on the 512 resident slots, makespan relative to the lower bound max(longest, work / 512), for the shipped order
(moving average of the previous counts, longest first), other predictors, the true counts, and for solves cut into chunks
that are relaunched until every instance has finished (continuation overhead 10 iteration-times)."""
import heapq, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
its = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r3_iter_trace_n32_mixed.npy")).astype(float)
S, B = its.shape
SLOTS, OV_FIRST, OV_CONT = 512, 12.0, 10.0

def makespan(order, w):
    h = [0.0] * SLOTS
    heapq.heapify(h)
    for i in order:
        heapq.heappush(h, heapq.heappop(h) + w[i])
    return max(h)

def chunked(it, bounds, ema):
    total, prev = 0.0, 0
    for li, bd in enumerate(bounds):
        alive = np.where(it > prev)[0]
        if len(alive) == 0:
            break
        w = np.minimum(it[alive], bd) - prev + (OV_FIRST if li == 0 else OV_CONT)
        order = np.argsort(-ema[alive], kind="stable") if li == 0 else np.arange(len(alive))
        total += makespan(order, w)
        prev = bd
    return total

res, ema = {}, None
for s in range(S):
    it = its[s]
    if s >= 3:
        w = it + OV_FIRST
        lb = max(w.max(), w.sum() / SLOTS)
        r = {"moving average 1/8 (shipped)": makespan(np.argsort(-ema, kind="stable"), w) / lb,
             "previous call's counts": makespan(np.argsort(-its[s - 1], kind="stable"), w) / lb,
             "maximum of the history": makespan(np.argsort(-its[:s].max(0), kind="stable"), w) / lb,
             "index order (no prediction)": makespan(np.arange(B), w) / lb,
             "true counts (not available)": makespan(np.argsort(-it, kind="stable"), w) / lb,
             "chunks 1500 | rest": chunked(it, [1500, 4000], ema) / lb,
             "chunks 600 | 1200 | 2200 | rest": chunked(it, [600, 1200, 2200, 4000], ema) / lb,
             "chunks of 500": chunked(it, [500 * i for i in range(1, 9)], ema) / lb,
             "corr(previous, current)": np.corrcoef(its[s - 1], it)[0, 1]}
        for k, v in r.items():
            res.setdefault(k, []).append(v)
    ema = it.copy() if ema is None else ema + (it - ema) * 0.125
print("N = 32 mixed gaits, %d calls x %d instances: mean %.0f iterations, %.1f %% at max_iter, quantiles 50/90/99 %% %s" % (
    S, B, its[3:].mean(), 100 * (its[3:] >= 4000).mean(), np.percentile(its[3:], [50, 90, 99])))
print("instances that hit max_iter in some call: %d, in every call: %d" % ((its[3:] >= 4000).any(0).sum(), (its[3:] >= 4000).all(0).sum()))
for k, v in res.items():
    print("%-36s %.3f" % (k, np.mean(v)))
