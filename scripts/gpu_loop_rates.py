"""The 1:10 control loop's rates at batch 4096 (bench.py's device_resident_loop legs, more iterations than the bench line takes):
single handle, asynchronous MPC mode, two staggered groups.  QRW_WBC16=0 for the quad WBC kernel.  gpurun -- python3 scripts/gpu_loop_rates.py [iters]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "quadruped-reactive-walking_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench, synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B, N, Ng = 4096, 16, 20
dev = torch.device("cuda:0")
sb = synth.SyntheticBatch(B, N, N_gait=Ng, gaits=("trot",))
for name, kw in (("single", {}), ("async", dict(multiprocessing=True)), ("two_groups_staggered", dict(groups=2, free_running=True, stagger=True))):
    r = bench.device_resident_loop(sb, B, N, Ng, dev, iters=iters, **kw)
    print(name, {"M_iterations_per_s": round(r["value"] / 1e6, 3), "ms_per_iteration": round(r["ms_per_iteration"], 4), "paced": r["paced_2ms_latency_ms"],
                 "stopped": r["instances_in_security_stop"]}, flush=True)
