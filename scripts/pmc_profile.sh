#!/bin/bash
# rocprofv3 counter passes of one bench shape on the GPU box (run through gpurun from the repo root):
#   scripts/pmc_profile.sh TAG [bench.py arguments of the shape]      e.g.  bench_b4096
#                                                                           n32_mixed_time_sliced --n-steps 32 --gaits walk,trot,bounding
#                                                                           bench_b256 --batch 256      bench_b1 --batch 1
# -> gpurun_out/pmc_$ROUND_TAG/{summary.json, stamp.json}; copy them to profiles/$ROUND_pmc_summary_TAG.json / _pmc_stamp_TAG.json
# (bench.py's PMC_SHAPES maps shapes to tags).  Counters go in separate passes (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share
# one; no trace domains with --pmc); python3 itself after `--` (no env / shell hop under the profiler).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r6}
TAG=${1:-bench_b4096}
shift || true
ARGS="$* --no-cpu-baseline --no-secondary --no-configs --steps 3 --warmup 2"
OUT=$R/gpurun_out/pmc_${ROUND}_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq -o sq --output-format csv -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
# second SQ pass: where the wave cycles go (parked at s_waitcnt / issue stalls / active), and the chip clock (GRBM_GUI_ACTIVE)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1 || echo "sq2 pass failed (counter names?)"
python3 $R/scripts/pmc_summarize.py $OUT > $OUT/summary.json
# stamp of the kernel sources these counters belong to (bench.py refuses a summary collected on other sources)
(cd $R && python3 -c "import bench, json; print(json.dumps({'mpc_source_sha256': bench.mpc_source_stamp()}))") > $OUT/stamp.json
rm -rf $OUT/fetch $OUT/write $OUT/sq $OUT/sq2   # (the raw per-dispatch CSVs: tens of MB; the summary is what is kept)
head -c 400 $OUT/summary.json
