#!/bin/bash
# rocprofv3 counter passes of BASELINE config 4's bench leg (batch 4096, N = 32, walk / trot / bounding, time-sliced launch) on the GPU
# box: where the two wavefronts' cycles go (SQ counters only; separate passes, no trace domains with --pmc).  gpurun_out/pmc_n32/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_n32
mkdir -p $OUT
ARGS="--n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs --steps 3 --warmup 2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq -o sq --output-format csv -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1
python3 $R/scripts/pmc_summarize.py $OUT > $OUT/summary.json
head -c 400 $OUT/summary.json
