"""CPU study on gpurun_out/iters_b4096.npy (scripts/gpu_collect_iters.py): is a launch of the headline batch bounded by its longest
solve, where does that solve sit in the predicted (EMA 1/8) order, and what would compute units reserved for the K solves predicted
longest buy (one wavefront per compute unit runs the sweeps 14 % faster, profiles/r4_pmc_lds_contention.txt)?"""
import heapq, sys
import numpy as np
it_all = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/iters_b4096.npy").astype(float)
S, B = it_all.shape
TAU, SPEED = 2.31e-3, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0 / 0.90  # ms per iteration on a full compute unit; alone

def simulate(order, w, K, cus=256):
    """list scheduling: the first K tasks of `order` each get a compute unit alone (4 slots held, speed SPEED) -- the other tasks
    go to the slot that frees first; a reserved compute unit's 4 slots join the pool when its task ends."""
    w = w + 9.0
    slots = [0.0] * (4 * (cus - K))
    heapq.heapify(slots)
    end = 0.0
    for i in order[:K]:
        t = w[i] / SPEED
        end = max(end, t)
        for _ in range(4): heapq.heappush(slots, t)
    for i in order[K:]:
        t = heapq.heappop(slots) + w[i]
        end = max(end, t)
        heapq.heappush(slots, t)
    return end * TAU

ema = it_all[0].copy()
rows = []
for s in range(1, S):
    it = it_all[s]
    order = np.argsort(-ema, kind="stable")
    rank = np.empty(B, int); rank[order] = np.arange(B)
    top = np.argsort(-it)[:5]
    if s >= 5:
        rows.append([it.max(), it.sum() / 1024, rank[top[0]], rank[top[1]], rank[top[2]]] + [simulate(order, it, K) for K in (0, 4, 8, 16, 32, 64)]
                    + [simulate(np.argsort(-it), it, 0), simulate(np.argsort(-it), it, 8)])
    ema = ema + (it - ema) * 0.125
r = np.array(rows)
np.set_printoptions(linewidth=200, precision=2, suppress=True)
print("step: max it, work/1024, rank of the 3 longest in the predicted order | simulated launch ms for K = 0 4 8 16 32 64 | true order K=0, K=8")
print(r[:12])
print("mean:", r.mean(0))
print("median rank of the longest solve in the predicted order:", np.median(r[:, 2]), " share with rank < 8/16/32/64:",
      [(r[:, 2] < k).mean() for k in (8, 16, 32, 64)])
