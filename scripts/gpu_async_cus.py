"""Free-running asynchronous 1:10 loop (bench.py's secondary_ratio_1_10_async) for several splits of the compute units
between the control loop's stream and the MPC's: python scripts/gpu_async_cus.py [loop_cus ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import torch
import bench, synth
B, N = 4096, 16
sb = synth.SyntheticBatch(B, N, N_gait=20, gaits=("trot",), n_seq=2)
for lc in [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48]:
    r = bench.device_resident_loop(sb, B, N, 20, torch.device("cuda", 0), multiprocessing=True, loop_cus=lc)
    print("loop_cus %3d: %.2f M iterations/s, %.3f ms per iteration, paced latency %s" % (lc, r["value"] / 1e6, r["ms_per_iteration"], json.dumps(r["paced_2ms_latency_ms"])), flush=True)
