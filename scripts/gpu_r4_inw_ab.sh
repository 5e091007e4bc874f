#!/bin/bash
# round 4: N = 32 exchange variants (QRW_N32_INWAVE = 0 LDS + barriers / 1 ds_bpermute / 2 LDS inside the wavefront) -- parity of
# the variants named in $1 (default "2"), then an A/B of config 4's bench leg on one box: scripts/gpu_r4_inw_ab.sh "1 2" "0 1 2"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for v in ${1:-2}; do
  QRW_HIP_LIB=$R/build/lib_inw$v.so timeout -k 10 900 python3 -m pytest tests/test_gpu_mpc.py -x -q -m gpu -k "long_horizons or wide_oracle or time_sliced or sequence_launch or config4 or full_size" > gpurun_out/r4_inw${v}_tests.log 2>&1 || { tail -30 gpurun_out/r4_inw${v}_tests.log; exit 1; }
  tail -1 gpurun_out/r4_inw${v}_tests.log
done
for i in 1 2; do
  for v in ${2:-0 2}; do
    l=build/lib_inw$v.so
    QRW_HIP_LIB=$R/$l timeout -k 10 300 python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), d['kernels_ms']['mpc_solve_kernel'], d['roofline']['frac'])" | tee -a gpurun_out/r4_inw_ab.txt
  done
done
