#!/bin/bash
# loop kernels (wbc16_kernel, control_pre_quad_kernel) built with other instruction-scheduling strategies: rocprofv3 averages per library
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in "" build/lib_loop_max-ilp.so build/lib_loop_max-memory-clause.so build/lib_loop_iterative-ilp.so; do
  echo "== ${l:-shipped}"
  if [ -n "$l" ]; then export QRW_HIP_LIB=$R/$l; else unset QRW_HIP_LIB; fi
  bash $R/scripts/gpu_loop_profile.sh 2>&1 | grep -E "wbc16|control_pre"
done
