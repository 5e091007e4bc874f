#!/bin/bash
# On the GPU box (through gpurun, from the repo root): everything profiles/ is refreshed from, into gpurun_out/refresh/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/r1_bench_line.json 2> $OUT/bench.err
echo "bench done"
python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline > $OUT/r1_bench_line_n32_mixed.json 2> $OUT/bench32.err
echo "bench n32 done"
python3 bench.py --batch 1 --no-cpu-baseline --no-secondary > $OUT/bench_b1.json 2>> $OUT/bench.err
python3 bench.py --batch 256 --no-cpu-baseline --no-secondary > $OUT/bench_b256.json 2>> $OUT/bench.err
python3 bench.py --batch 16384 --no-cpu-baseline --no-secondary > $OUT/bench_b16384.json 2>> $OUT/bench.err
echo "batch sweep done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/r1_bench_line_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r1_kernel_stats_bench_b4096.csv
echo "rocprof stats done"
cd $R && bash scripts/pmc_profile.sh > $OUT/pmc.log 2>&1 && cp $R/gpurun_out/pmc_r1/summary.json $OUT/r1_pmc_summary_bench_b4096.json
echo "pmc done"
