#!/bin/bash
# Build a variant of libqrw_hip.so that differs only in mpc_kernel.hip's compile flags: scripts/build_variant.sh NAME "-DFLAG ..." ["extra hipcc flags"]
# -> build/lib_NAME.so (select it with QRW_HIP_LIB=build/lib_NAME.so; scripts/gpu_ab.sh compares two of them on one GPU box)
set -e
R=/root/repo
C=$R/quadruped-reactive-walking_amd/csrc
mkdir -p $R/build/var
SCHED=${QRW_SCHED:--mllvm -amdgpu-sched-strategy=max-ilp}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -Wno-pass-failed $SCHED $2 $3 \
  -c -o $R/build/var/mpc_$1.o $C/mpc_kernel.hip
make -s -C $C >/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/lib_$1.so $C/qrw_api.o $R/build/var/mpc_$1.o $C/wbc_kernel.o $C/planner_kernel.o $C/controller_kernel.o
echo built build/lib_$1.so
