#!/bin/bash
# A/B of two environment settings on the bench headline (same binary), one GPU box:
# scripts/gpu_ab_env.sh "VAR=a" "VAR=b" [rounds] [out file] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for i in $(seq 1 ${3:-3}); do
  for e in "$1" "$2"; do
    env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --no-configs $5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', round(d['value']), d['kernels_ms']['mpc_solve_kernel'], d['roofline']['frac'], d['roofline']['mean_admm_iters'])" | tee -a gpurun_out/${4:-r5_ab_env.txt}
  done
done
