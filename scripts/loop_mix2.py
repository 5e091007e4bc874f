"""Static instruction mix of the ADMM loop body of one mpc_solve_kernel listing: python scripts/loop_mix2.py build/k2pre.s
(the body = from the first QRW rhs code after the factor block to the loop's back edge; found as the largest Depth=1 loop)."""
import collections, re, sys
L = open(sys.argv[1]).read().splitlines()
def cls(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_bpermute"): return "bpermute"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_mov_b32_dpp"): return "mov_dpp"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"): return "vmov"
    if op.startswith("v_") and "f64" in op: return "valu64"
    if op.startswith("v_"): return "valu32"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"
# loop body: lines between the label that follows the factor block's execz skip and the back edge
hdr = [i for i, l in enumerate(L) if "Loop Header: Depth=1" in l and "Child Loop" in L[i + 1]]
hdr = hdr[-1] if len(sys.argv) < 3 else hdr[int(sys.argv[2])]
tgt = None
for i in range(hdr, hdr + 40):
    m = re.search(r"s_cbranch_execz (\.LBB\d+_\d+)", L[i])
    if m: tgt = m.group(1); break
start = [i for i, l in enumerate(L) if l.startswith(tgt + ":")][0]
end = [i for i, l in enumerate(L) if ("s_cbranch_scc0" in l or "s_cbranch_vccnz" in l or "s_cbranch_scc1" in l) and i > start][0]
c = collections.Counter(); ops = collections.Counter()
for l in L[start:end]:
    t = l.strip()
    if not t or t.startswith((";", ".", "#")) or t.endswith(":"): continue
    op = t.split()[0]; c[cls(op)] += 1; ops[op] += 1
print("loop body lines %d..%d: %d instructions" % (start, end, sum(c.values())), dict(sorted(c.items())))
