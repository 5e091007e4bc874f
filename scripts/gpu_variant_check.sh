#!/bin/bash
# quick correctness verdict of variant builds on the GPU box: scripts/gpu_variant_check.sh build/lib_a.so build/lib_b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in "$@"; do
  echo "== $l: $(QRW_HIP_LIB=$R/$l timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_mpc.py -q -x -k 'trot_batch or sweeps_selftest' 2>&1 | tail -1)"
done
