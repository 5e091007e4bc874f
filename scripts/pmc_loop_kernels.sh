#!/bin/bash
# rocprofv3 counter passes of the 1:10 control loop (scripts/gpu_loop_only.py): instruction counts, wait / issue split and
# instruction-cache behaviour of control_pre_kernel and wbc_kernel.  Run through gpurun from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_loop
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT -d $OUT/sq -o sq --output-format csv -- python3 $R/scripts/gpu_loop_only.py 20 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/scripts/gpu_loop_only.py 20 > $OUT/sq2.log 2>&1 || echo "sq2 failed"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/ic -o ic --output-format csv -- python3 $R/scripts/gpu_loop_only.py 20 > $OUT/ic.log 2>&1 || echo "ic failed"
python3 $R/scripts/pmc_summarize.py $OUT > $OUT/summary.json
