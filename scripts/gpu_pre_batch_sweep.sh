#!/bin/bash
# On the GPU box: the N = 32 mixed-gait bench leg at smaller batches, unsliced / one FIFO / priority levels.
for B in 768 1024 2048; do for cfg in "0:9" "600:1" "600:9"; do IFS=: read C L <<< "$cfg"; QRW_PREEMPT_CHUNK=$C QRW_PREEMPT_LEVELS=$L python3 bench.py --batch $B --n-steps 32 --gaits walk,trot,bounding --no-cpu-baseline --no-secondary --no-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B chunk=$C levels=$L: %.0f steps/s %.2f ms'%(d['value'], d['ms_per_step']))"; done; done
