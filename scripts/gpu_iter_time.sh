#!/bin/bash
# per-ADMM-iteration time of builds made by scripts/experiments/build_timing_experiment.sh with -DQRW_EXPERIMENT_NOTERM (every instance runs 4000 iterations):
# scripts/gpu_iter_time.sh build/lib_a.so build/lib_b.so ...   (batch 4096 = 4 rounds of 1024 resident instances)
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in "$@"; do
  QRW_ALLOW_WRONG_RESULTS=1 QRW_HIP_LIB=$R/$l python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; ms=r['launch_ms_mean']
print('$l', 'launch %.2f ms, mean iters %.0f -> %.3f us per ADMM iteration (4 rounds of 1024 resident instances)' % (ms, r['mean_admm_iters'], ms*1e3/4/r['mean_admm_iters']))"
done
