"""How good is the longest-first block order? Collects per-instance ADMM iteration counts over consecutive solves of
the bench workload and simulates list scheduling on 1024 SIMD slots with (a) the previous solve's counts as the
order (what the order kernel does), (b) the actual counts (oracle order), against the lower bound."""
import heapq, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np
import qrw_hip, synth

def makespan(order, w, slots=1024):
    h = [0.0] * slots
    heapq.heapify(h)
    for i in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + w[i])
    return max(h)

B, N = 4096, 16
sb = synth.SyntheticBatch(B, N, n_seq=48)
g = qrw_hip.Batch(B, N)
prev = None
hist = []
ema = {}
tot = {}
for s in range(44):
    d = sb.step(s)
    gait_now = (d["fsteps"].reshape(B, -1, 4, 3)[:, :N, :, 0] != 0)
    sw0 = (gait_now[:, 0] != gait_prev[:, 0]).any(1).astype(float) if s > 0 else np.zeros(B)
    gait_prev = gait_now
    g.mpc_solve_host(d["xref"], d["fsteps"], s)
    it = g.mpc_stats()["iters"].astype(float) + 9.0  # + setup/factor overhead in iteration units
    if prev is not None and s >= 3:
        lb = max(it.max(), it.sum() / 1024)
        a = makespan(np.argsort(-prev, kind="stable"), it)
        b = makespan(np.argsort(-it, kind="stable"), it)
        c = makespan(np.arange(B), it)
        if len(hist) >= 3:
            preds = {"prev": prev, "max2": np.maximum(hist[-1], hist[-2]), "max3": np.maximum(np.maximum(hist[-1], hist[-2]), hist[-3]),
                     "mean3": (hist[-1] + hist[-2] + hist[-3]) / 3, "ema": 0.5 * hist[-1] + 0.3 * hist[-2] + 0.2 * hist[-3]}
            if len(hist) >= 16:
                preds["period16"] = hist[-16]
                preds["max(prev,period16)"] = np.maximum(hist[-1], hist[-16])
                preds["mean(prev,period16)"] = 0.5 * (hist[-1] + hist[-16])
            for al in (0.1, 0.2, 0.3, 0.5):
                if 8 in ema:
                    preds["ema8*(1+%.1f*sw0)" % al] = ema[8] * (1 + al * sw0)
            for a_ in (4, 8, 16):
                if a_ in ema:
                    preds["ema%d" % a_] = ema[a_]
                    preds["max(prev,ema%d)" % a_] = np.maximum(prev, ema[a_])
            if len(hist) >= 5:
                preds["max5"] = np.max(np.stack(hist[-5:]), 0)
            if len(hist) >= 8:
                preds["max8"] = np.max(np.stack(hist[-8:]), 0)
                preds["mean8"] = np.mean(np.stack(hist[-8:]), 0)
                preds["p90_8"] = np.percentile(np.stack(hist[-8:]), 90, axis=0)
            if len(hist) >= 16:
                preds["max16"] = np.max(np.stack(hist[-16:]), 0)
                preds["mean16"] = np.mean(np.stack(hist[-16:]), 0)
                preds["max(max3,p16)"] = np.maximum(preds["max3"], hist[-16])
            if len(hist) >= 32:
                preds["mean(p16,p32)"] = 0.5 * (hist[-16] + hist[-32])
            for kk, pv in preds.items():
                tot.setdefault(kk, []).append(makespan(np.argsort(-pv, kind="stable"), it) / lb)
            tot.setdefault("actual", []).append(b / lb)
        if s % 8 == 0: print("step %d: mean %.0f max %.0f | lower bound %.0f | order by previous counts %.0f (%.2fx) | by actual counts %.0f (%.2fx) | index order %.0f (%.2fx) | corr(prev,cur) %.3f"
              % (s, it.mean(), it.max(), lb, a, a / lb, b, b / lb, c, c / lb, np.corrcoef(prev, it)[0, 1]))
    for a_ in (4, 8, 16):
        ema[a_] = it.copy() if a_ not in ema else ema[a_] + (it - ema[a_]) / a_
    prev = it
    hist.append(it)
print({k: round(float(np.mean(v)), 3) for k, v in tot.items()})
