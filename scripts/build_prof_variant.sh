#!/bin/bash
# Phase-counter build (-DQRW_PROFILE_PHASES, default scheduling as `make prof`) of a variant of mpc_kernel.hip:
# scripts/build_prof_variant.sh NAME "-DFLAG ..."  -> build/lib_prof_NAME.so (QRW_HIP_LIB=... python scripts/gpu_phases.py)
set -e
R=/root/repo
C=$R/quadruped-reactive-walking_amd/csrc
mkdir -p $R/build/var
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -Wno-pass-failed -DQRW_PROFILE_PHASES $2 \
  -c -o $R/build/var/mpc_prof_$1.o $C/mpc_kernel.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -DQRW_PROFILE_PHASES \
  -c -o $R/build/var/api_prof.o $C/qrw_api.hip
make -s -C $C >/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/lib_prof_$1.so $R/build/var/api_prof.o $R/build/var/mpc_prof_$1.o $C/wbc_kernel.o $C/planner_kernel.o $C/controller_kernel.o
echo built build/lib_prof_$1.so
