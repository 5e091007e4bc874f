#!/bin/bash
# TIMING EXPERIMENTS ONLY -- builds that compute WRONG RESULTS on purpose (upper bounds of what a re-arrangement of the sweeps
# could gain; profiles/r2_iter_time_noterm_vs_nodep.txt, profiles/r4_n32_experiments.txt).  The code they need is NOT in the
# shipped sources: this script copies csrc/ to build/exp_src/pkg/csrc, applies timing_experiments.patch there (which also adds
# the self-test bypass such a build needs) and links build/WRONG_RESULTS_<NAME>.so -- a name qrw_hip.py refuses to load unless
# QRW_ALLOW_WRONG_RESULTS=1 is set next to QRW_HIP_LIB.
#   scripts/experiments/build_timing_experiment.sh NAME "-DQRW_EXPERIMENT_NOTERM [-DQRW_EXPERIMENT_NODEP] [-DQRW_EXPERIMENT_HALFREADS]"
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
S=$R/build/exp_src
rm -rf "$S" && mkdir -p "$S/pkg/csrc"
cp $R/quadruped-reactive-walking_amd/csrc/*.h $R/quadruped-reactive-walking_amd/csrc/*.hip "$S/pkg/csrc/"
ln -sfn $R/include "$S/include"   # the sources include ../../include/...
(cd "$S/pkg/csrc" && patch -p1 < $R/scripts/experiments/timing_experiments.patch)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -Wno-pass-failed"
for f in qrw_api mpc_kernel wbc_kernel planner_kernel controller_kernel; do
  extra=""; [ $f = mpc_kernel ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
  hipcc $FLAGS $extra $2 -c -o "$S/$f.o" "$S/pkg/csrc/$f.hip"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build/WRONG_RESULTS_$1.so" "$S"/*.o
echo "built build/WRONG_RESULTS_$1.so (load with QRW_HIP_LIB=... QRW_ALLOW_WRONG_RESULTS=1)"
