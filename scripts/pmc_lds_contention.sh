#!/bin/bash
# Round 4: what the four wavefronts of a compute unit lose to the LDS pipe they share.  SQ counter passes of
# the headline kernel (N = 16) at batch 256 (one wavefront per compute unit), batch 512 (two) and batch 4096 (four resident, four
# rounds): LDS instructions, cycles an LDS instruction was active / waited for, bank conflicts.  Separate --pmc passes, no trace
# domains.  gpurun_out/pmc_lds/ -> scripts/pmc_summarize.py
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -o "SQ_[A-Z_]*LDS[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/lds_counters.txt
for B in ${BATCHES:-256 512 4096}; do
  ARGS="--batch $B --no-cpu-baseline --no-secondary --no-configs --steps 3 --warmup 2"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT/b$B/sq -o sq --output-format csv -- python3 $R/bench.py $ARGS > $OUT/b$B.sq.log 2>&1 || echo "pass sq b$B failed"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d $OUT/b$B/lds -o lds --output-format csv -- python3 $R/bench.py $ARGS > $OUT/b$B.lds.log 2>&1 || echo "pass lds b$B failed"
  python3 $R/scripts/pmc_summarize.py $OUT/b$B > $OUT/summary_b$B.json
  tail -1 $OUT/b$B.sq.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('batch $B', round(d['value']), 'steps/s', d['kernels_ms']['mpc_solve_kernel'], 'ms per launch, mean iterations', d.get('mpc_iters_mean'))
except Exception as e: print('no bench line', e)"
done
python3 - <<PY
import json
for B in (256, 512, 4096, 16384):
    try:
        s = json.load(open("$OUT/summary_b%d.json" % B))
    except Exception as e:
        print(B, "no summary", e); continue
    m = {r["counter"]: r["mean"] for r in s if r["kernel"].startswith("qrw::mpc_solve_kernel")}
    wc = m.get("SQ_WAVE_CYCLES", float("nan"))
    print("batch %d:" % B, {k: round(v / wc, 4) for k, v in m.items() if k != "SQ_WAVE_CYCLES"}, "wave cycles %.3g" % wc)
PY
