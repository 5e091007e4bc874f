#!/bin/bash
# A/B of builds of libqrw_hip.so with arbitrary bench.py arguments on one GPU box:
# scripts/gpu_ab_args.sh "build/lib_a.so build/lib_b.so" "<bench args>" [rounds] [key printed besides value]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${3:-2}); do
  for l in $1; do
    QRW_HIP_LIB=$R/$l timeout -k 10 600 python3 bench.py $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k='${4:-}'; print('$l', round(d['value']), d['kernels_ms']['mpc_solve_kernel'], (d.get(k) if k else ''))"
  done
done
