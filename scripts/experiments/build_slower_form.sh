#!/bin/bash
# MEASURED-SLOWER FORMS -- parity-green variants of the MPC kernel that lost their A/B and left the shipped translation units in
# round 6 (docs/HISTORY.md has the measurements):
#   -DQRW_N32_DISSECT=1        N = 32: the horizon dissected around step 16, both wavefronts sweep their own half (dissect.h);
#                              60 k against 75 k control steps/s at batch 4096 (round 3)
#   -DQRW_BANK_FREE_LAYOUT=1   chain matrices laid out without LDS bank conflicts between the two chains; +-0 (round 4)
#   -DQRW_CHAIN_READ2          sweep operands through plain loads (merged into ds_read2_b64); slower per byte (round 2)
#   -DQRW_NO_ACCD              Delta^-1 rows in compiler-allocated registers instead of pinned AGPRs; spills (round 2)
# The code they need is NOT in csrc/: this script copies csrc/ to build/slow_src/pkg/csrc, applies slower_forms.patch there
# (which also restores the dissected solve's self-test) and links build/libqrw_hip_<NAME>.so; QRW_HIP_LIB=... selects it.
#   scripts/experiments/build_slower_form.sh NAME "-DQRW_N32_DISSECT=1" [PROFILE]
# PROFILE (any third argument) adds -DQRW_PROFILE_PHASES and drops the max-ilp scheduling, as `make prof` does.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
S=$R/build/slow_src
rm -rf "$S" && mkdir -p "$S/pkg/csrc"
cp $R/quadruped-reactive-walking_amd/csrc/*.h $R/quadruped-reactive-walking_amd/csrc/*.hip "$S/pkg/csrc/"
ln -sfn $R/include "$S/include"   # the sources include ../../include/...
(cd "$S/pkg/csrc" && patch -p1 < $R/scripts/experiments/slower_forms.patch)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -Wno-pass-failed"
for f in qrw_api mpc_kernel wbc_kernel planner_kernel controller_kernel; do
  extra=""
  if [ -n "$3" ]; then extra="-DQRW_PROFILE_PHASES"; elif [ $f = mpc_kernel ]; then extra="-mllvm -amdgpu-sched-strategy=max-ilp"; fi
  hipcc $FLAGS $extra $2 -c -o "$S/$f.o" "$S/pkg/csrc/$f.hip"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build/libqrw_hip_$1.so" "$S"/*.o
echo "built build/libqrw_hip_$1.so (load with QRW_HIP_LIB=$R/build/libqrw_hip_$1.so)"
