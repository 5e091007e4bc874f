#!/bin/bash
# On the GPU box: parity tests, bench line, phase breakdown (expects build/libqrw_hip_prof.so).
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; tail -4 gpurun_out/t_gpu.log
python bench.py --no-cpu-baseline > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err
python -c "
import json; d=json.load(open('gpurun_out/bench_cur.json')); print(d['value'], d['kernels_ms'], d['roofline']['mean_admm_iters'], d['secondary_ratio_1_10']['value'])"
python scripts/gpu_phases.py > gpurun_out/phases.log 2>&1; tail -11 gpurun_out/phases.log
