#!/bin/bash
# A/B of two builds of libqrw_hip.so on the same GPU box: scripts/gpu_ab.sh build/lib_a.so build/lib_b.so [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${3:-3}); do
  for l in $1 $2; do
    QRW_HIP_LIB=$R/$l python3 $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), d['kernels_ms']['mpc_solve_kernel'])"
  done
done
