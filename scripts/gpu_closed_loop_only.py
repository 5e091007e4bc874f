"""bench.py's closed_loop_sequence leg alone (single handle, then the same replay as two stream groups), for tracing."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import torch
import bench
print(json.dumps(bench.closed_loop_sequence(4096, 16, 20, ("trot",), torch.device("cuda", 0), 4, int(sys.argv[1]) if len(sys.argv) > 1 else 20)))
