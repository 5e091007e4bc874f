#!/bin/bash
# rocprofv3 counter passes of the default bench workload on the GPU box (run through gpurun from the repo root).
# Counters go in separate passes (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share one; no trace domains with --pmc).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r3}
OUT=$R/gpurun_out/pmc_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq -o sq --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/sq.log 2>&1
# second SQ pass: where the wave cycles go (parked at s_waitcnt / issue stalls / active), and the chip clock (GRBM_GUI_ACTIVE)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 > $OUT/sq2.log 2>&1 || echo "sq2 pass failed (counter names?)"
python3 $R/scripts/pmc_summarize.py $OUT > $OUT/summary.json
# stamp of the kernel sources these counters belong to (bench.py refuses a summary collected on other sources)
(cd $R && python3 -c "import bench, json; print(json.dumps({'mpc_source_sha256': bench.mpc_source_stamp()}))") > $OUT/stamp.json
head -c 600 $OUT/summary.json
