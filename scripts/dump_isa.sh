#!/bin/bash
# Dump the gfx950 ISA of mpc_kernel.hip into build/ (k11.s = mpc_solve_kernel<1,true>) and print resource usage.
set -e
R=/root/repo
mkdir -p $R/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -I$R/include -I$R/quadruped-reactive-walking_amd/csrc \
  -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value -Wno-unused-result -Wno-unused-function -Rpass-analysis=kernel-resource-usage \
  $R/quadruped-reactive-walking_amd/csrc/mpc_kernel.hip -o $R/build/mpc_kernel.s 2> $R/build/mpc_kernel.remarks || { cat $R/build/mpc_kernel.remarks | grep error; exit 1; }
awk '/^_ZN3qrw16mpc_solve_kernelILi1ELb1ELb0ELb0EEEvNS_7MpcArgsE:/,/s_endpgm/' $R/build/mpc_kernel.s > $R/build/k11.s
grep -A12 "mpc_solve_kernelILi1ELb1ELb0ELb0" $R/build/mpc_kernel.remarks | grep -E "VGPRs|AGPRs|Scratch|Spill|LDS" | sed 's/.*remark: *//'
echo "scratch ops in k11: $(grep -c scratch_ $R/build/k11.s)  lines: $(wc -l < $R/build/k11.s)"
