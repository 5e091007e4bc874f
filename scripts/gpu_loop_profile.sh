#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of a synchronous Controller_batch loop at batch 4096.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/loopk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lk -- python3 $R/scripts/gpu_loop_kernels.py "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
