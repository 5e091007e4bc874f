"""Static instruction mix of the ADMM loop body (outside the factorisation and the every-25th-iteration check) of
mpc_solve_kernel<1,true>, from build/k11.s (scripts/dump_isa.sh)."""
import collections, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = open(os.path.join(R, "build", "k11.s")).read().splitlines()
def cls(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_mov_b32_dpp"): return "mov_dpp"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"): return "vmov"
    if op.startswith("v_") and "f64" in op: return "valu64"
    if op.startswith("v_"): return "valu32"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait/nop"
    if op.startswith("s_"): return "salu"
    return "other"
hdr = [i for i, l in enumerate(L) if "Loop Header: Depth=1" in l and "Child Loop" in L[i + 1]][0]
tgt = None
for i in range(hdr, hdr + 30):
    m = re.search(r"s_cbranch_execz (\.LBB\d+_\d+)", L[i])
    if m: tgt = m.group(1); break
start = [i for i, l in enumerate(L) if l.startswith(tgt + ":")][0]
end = [i for i, l in enumerate(L) if "s_cbranch_scc0" in l and i > start][0]
c = collections.Counter(); ops = collections.Counter()
for l in L[start:end]:
    t = l.strip()
    if not t or t.startswith((";", ".", "#")) or t.endswith(":"): continue
    op = t.split()[0]; c[cls(op)] += 1; ops[op] += 1
print("loop body lines %d..%d: %d instructions" % (start, end, sum(c.values())), dict(c))
print(ops.most_common(14))
