#!/bin/bash
# A/B of builds of libqrw_hip.so on config 4's bench leg (batch 4096, N = 32, walk / trot / bounding), one GPU box:
# scripts/gpu_ab_n32.sh "build/lib_a.so build/lib_b.so ..." [rounds] [out file]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for i in $(seq 1 ${2:-2}); do
  for l in $1; do
    QRW_HIP_LIB=$R/$l timeout -k 10 300 python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), d['kernels_ms']['mpc_solve_kernel'], d['roofline']['frac'])" | tee -a gpurun_out/${3:-r4_ab_n32.txt}
  done
done
