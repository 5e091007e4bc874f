"""The device-resident 1:10 control loop alone (bench.py's secondary_ratio_1_10 leg), for profiling control_pre_kernel /
wbc_kernel: python scripts/gpu_loop_only.py [iterations]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import torch
import bench, synth
B, N = int(os.environ.get("QRW_EXP_B", "4096")), 16
sb = synth.SyntheticBatch(B, N, N_gait=20, gaits=("trot",), n_seq=2)
print(json.dumps(bench.device_resident_loop(sb, B, N, 20, torch.device("cuda", 0), iters=int(sys.argv[1]) if len(sys.argv) > 1 else 40)))
