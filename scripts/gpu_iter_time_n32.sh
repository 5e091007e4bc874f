#!/bin/bash
# per-ADMM-iteration time of -DQRW_EXPERIMENT_NOTERM builds (scripts/experiments/build_timing_experiment.sh) at N = 16 (4 rounds of 1024 resident instances) and N = 32 (unsliced: 8 rounds
# of 512): scripts/gpu_iter_time_n32.sh build/lib_a.so build/lib_b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in "$@"; do
  QRW_ALLOW_WRONG_RESULTS=1 QRW_HIP_LIB=$R/$l python3 $R/bench.py --no-cpu-baseline --no-secondary --no-configs --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; ms=r['launch_ms_mean']
print('$l N=16', 'launch %.2f ms, mean iters %.0f -> %.3f us per ADMM iteration' % (ms, r['mean_admm_iters'], ms*1e3/4/r['mean_admm_iters']))"
  QRW_ALLOW_WRONG_RESULTS=1 QRW_PREEMPT_CHUNK=0 QRW_HIP_LIB=$R/$l python3 $R/bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; ms=r['launch_ms_mean']
print('$l N=32', 'launch %.2f ms, mean iters %.0f -> %.3f us per ADMM iteration' % (ms, r['mean_admm_iters'], ms*1e3/8/r['mean_admm_iters']))"
done
