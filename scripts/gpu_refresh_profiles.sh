#!/bin/bash
# On the GPU box (through gpurun, from the repo root): everything profiles/ is refreshed from, into gpurun_out/refresh/.
# ROUND tags the file names (profiles/ keeps one set per round).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r6}
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_line_n32_mixed.json 2> $OUT/bench32.err
QRW_PREEMPT_CHUNK=0 python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_line_n32_mixed_unsliced.json 2>> $OUT/bench32.err
echo "bench n32 done"
python3 bench.py --batch 1 --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_b1.json 2>> $OUT/bench.err
python3 bench.py --batch 256 --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_b256.json 2>> $OUT/bench.err
python3 bench.py --batch 16384 --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_b16384.json 2>> $OUT/bench.err
echo "batch sweep done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_line_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_kernel_stats_bench_b4096.csv
cp $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace_b4096.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats32 -o st -- python3 $R/bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary > $OUT/${ROUND}_bench_line_n32_under_rocprof.json 2> $OUT/stats32.err
cp $(find $OUT/stats32 -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_kernel_stats_bench_n32_mixed.csv
cp $(find $OUT/stats32 -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace_n32.csv
QRW_PREEMPT_CHUNK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats32u -o st -- python3 $R/bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/stats32u.err
cp $(find $OUT/stats32u -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_kernel_stats_bench_n32_mixed_unsliced.csv
# the 1:10 control loop's own kernels (control_pre_quad_kernel, wbc16_kernel): 40 iterations at batch 4096
bash $R/scripts/gpu_loop_profile.sh > $OUT/loopk.log 2>&1 && cp $(find $R/gpurun_out/loopk -name "*kernel_stats.csv" | head -1) $OUT/${ROUND}_kernel_stats_control_loop.csv
echo "rocprof stats done"
cd $R
# counter passes of every bench shape that carries a roofline block (bench.py PMC_SHAPES): headline, config 4, config 2, batch 1
pmc() { tag=$1; shift; ROUND=$ROUND bash scripts/pmc_profile.sh $tag "$@" > $OUT/pmc_$tag.log 2>&1 && cp $R/gpurun_out/pmc_${ROUND}_$tag/summary.json $OUT/${ROUND}_pmc_summary_$tag.json && cp $R/gpurun_out/pmc_${ROUND}_$tag/stamp.json $OUT/${ROUND}_pmc_stamp_$tag.json; echo "pmc $tag done"; }
pmc bench_b4096
pmc n32_mixed_time_sliced --n-steps 32 --gaits walk,trot,bounding
pmc bench_b256 --batch 256
pmc bench_b1 --batch 1
echo "pmc done"
python3 scripts/trace_timed_avg.py $OUT/kernel_trace_b4096.csv 20 > $OUT/${ROUND}_timed_launch_avg_b4096.txt
python3 scripts/trace_timed_avg.py $OUT/kernel_trace_n32.csv 20 > $OUT/${ROUND}_timed_launch_avg_n32.txt
for n in 16 32; do QRW_PHASES_N=$n python3 scripts/gpu_phases.py > $OUT/${ROUND}_mpc_phase_cycles_n$n.txt 2>/dev/null; done
echo "phases done"
# the ONE bench line last, with this call's counter summary + stamp in place (bench.py quotes roofline.traffic from profiles/ only when
# the stamp matches the kernel sources)
cp $OUT/${ROUND}_pmc_summary_*.json $OUT/${ROUND}_pmc_stamp_*.json $R/profiles/
python3 bench.py > $OUT/${ROUND}_bench_line.json 2> $OUT/bench.err
echo "bench done"
