#!/bin/bash
# On the GPU box: the dissected N = 32 path (a measured-slower form, NOT in csrc/: build it first with
#   scripts/experiments/build_slower_form.sh dissect "-DQRW_N32_DISSECT=1"   and run this with QRW_HIP_LIB=build/libqrw_hip_dissect.so)
# -- linear-algebra self-test first, then the N = 32 parity tests, then timing.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 120 python3 - <<'PY' > gpurun_out/dis_selftest.log 2>&1
import sys
sys.path.insert(0, "quadruped-reactive-walking_amd")
import qrw_hip
rc, err = qrw_hip.selftest_sweeps()
msg = qrw_hip.load_library().qrw_last_error()
print("selftest rc", rc, "err", err, msg.decode() if rc else "")
sys.exit(1 if rc else 0)
PY
rc=$?; cat gpurun_out/dis_selftest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_mpc.py -x -q -k "long_horizons or wide_oracle or sequence_launch or mixed_gaits or config4" > gpurun_out/dis_tests.log 2>&1
rc=$?; tail -15 gpurun_out/dis_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --steps 6 --warmup 3 > gpurun_out/dis_bench_n32.json 2> gpurun_out/dis_bench_n32.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/dis_bench_n32.json
