#!/bin/bash
# scripts/build_wt_bad.sh SHA... : for each commit, a worktree copy under build/wt/SHA with the shipped build and the
# "bad" build of docs/HISTORY.md 6b (max-ilp + -DQRW_PROFILE_PHASES on mpc_kernel.hip) as libqrw_hip_bad.so
set -e
R=/root/repo
for sha in "$@"; do
  rm -rf $R/build/wt/$sha; mkdir -p $R/build/wt/$sha
  git -C $R archive $sha | tar -x -C $R/build/wt/$sha
  ( cd $R/build/wt/$sha/quadruped-reactive-walking_amd/csrc && make -s >/dev/null 2>&1
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -Wno-pass-failed \
      -mllvm -amdgpu-sched-strategy=max-ilp -DQRW_PROFILE_PHASES $QRW_EXTRA -c -o mpc_bad.o mpc_kernel.hip 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libqrw_hip_bad.so qrw_api.o mpc_bad.o wbc_kernel.o planner_kernel.o controller_kernel.o )
  ( cd $R/build/wt/$sha/oracle && make -s libqrw_oracle.so >/dev/null )
  echo built $sha
done
