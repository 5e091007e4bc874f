#!/bin/bash
# On the GPU box: control_pre as one quad per instance (default) against one thread per instance (QRW_PRE_QUAD=0):
# parity tests first, then the kernels' durations from rocprofv3 kernel traces of 40 control iterations at batch 4096.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_controller.py tests/test_gpu_planner.py -x -q > gpurun_out/preq_tests.log 2>&1
rc=$?; tail -4 gpurun_out/preq_tests.log
[ $rc -ne 0 ] && exit 1
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $R/gpurun_out/preq_$v
  QRW_PRE_QUAD=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/preq_$v -o st -- python3 $R/scripts/gpu_loop_kernels.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/preq_$v -name "*kernel_stats.csv" | head -1)
  echo "QRW_PRE_QUAD=$v"; grep -E "control_pre|wbc_kernel|Name" $f | cut -d, -f1-6 | sed 's/void //; s/(.*)//'
  cp $f $R/gpurun_out/preq_kernel_stats_quad$v.csv
done
