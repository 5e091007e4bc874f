"""Does the number of torch pool streams USED earlier in the process change what two stream groups reach?  (bench.py measured 600 k
for the closed sequence as two groups inside the full run and 970 k alone.)  GPU_MAX_HW_QUEUES=4 python scripts/gpu_stream_pool_exp.py N
(importing bench raises the queue count to 8 unless the variable is already set: N = 2 gives 600 k with 4 queues, 955-990 k with 8 or 16)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import torch
import bench
n = int(sys.argv[1])
keep = [torch.cuda.Stream(torch.device("cuda", 0)) for _ in range(n)]
for st in keep:  # a stream gets its hardware queue when it is first used
    with torch.cuda.stream(st):
        torch.zeros(8, device="cuda").add_(1.0)
torch.cuda.synchronize()
r = bench.closed_loop_sequence(4096, 16, 20, ("trot",), torch.device("cuda", 0), 4, 20)
print("pool streams created before: %d -> single %.0f, two groups %.0f" % (n, r["value"], r["two_stream_groups_steps_per_s"]), flush=True)
