#!/bin/bash
# config 4 (batch 4096, N = 32, mixed gaits) under different time-slicing knobs, one box: scripts/gpu_n32_knobs.sh "ENV1" "ENV2" ... (each a quoted VAR=val list)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for round in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 300 python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['value']), round(d['kernels_ms']['mpc_solve_kernel'],3))" | tee -a gpurun_out/r5_n32_knobs.txt
  done
done
