"""Asynchronous-mode 1:10 loop (bench.py's secondary_ratio_1_10_async) for several sizes of the loop's compute-unit mask and both
WBC kernels: rate and paced latency.  QRW_EXP_ASYNC_LANES=4|16 overrides the Controller's choice of WBC kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import torch
import bench, synth
B, N = 4096, 16
sb = synth.SyntheticBatch(B, N, N_gait=20, gaits=("trot",), n_seq=2)
for cus in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "32,48,64").split(",")]:
    for lanes in ("4", "16"):
        os.environ["QRW_EXP_ASYNC_LANES"] = lanes
        r = bench.device_resident_loop(sb, B, N, 20, torch.device("cuda", 0), iters=80, multiprocessing=True, loop_cus=cus)
        print(cus, lanes, "%.2f M/s" % (r["value"] / 1e6), r["paced_2ms_latency_ms"], flush=True)
