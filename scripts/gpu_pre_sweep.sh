#!/bin/bash
# On the GPU box: config 4's bench leg (batch 4096, N = 32, walk / trot / bounding) over the time-sliced launch's knobs.
# Usage: scripts/gpu_pre_sweep.sh "LEVELS:BIN:CHUNK ..."   (results: gpurun_out/pre_sweep/*.json, one summary line each)
mkdir -p gpurun_out/pre_sweep
for cfg in ${1:-"9:400:600 1:400:600"}; do
  IFS=: read L B C <<< "$cfg"
  QRW_PREEMPT_LEVELS=$L QRW_PREEMPT_BIN=$B QRW_PREEMPT_CHUNK=$C python3 bench.py --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs \
    > gpurun_out/pre_sweep/L${L}_B${B}_C${C}.json 2> gpurun_out/pre_sweep/L${L}_B${B}_C${C}.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/pre_sweep/L${L}_B${B}_C${C}.json').read().strip().splitlines()[-1])
print('levels $L bin $B chunk $C: %.1f steps/s, %.3f ms/step, roofline %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
"
done
