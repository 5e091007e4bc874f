"""AGPR-pinning checks on a gfx950 ISA listing of mpc_kernel.hip (hipcc -S --cuda-device-only [-DQRW_MARK_PHASES]).

mpc_solve_kernel parks values in accumulation registers through inline asm (qrw_device.h `AccD`: v_accvgpr_write_b32 to
park, v_accvgpr_read_b32 to fetch: the 72 registers of the Delta^-1 rows, read twelve at a time in every ADMM iteration, and
48 of rarely read scalings).  The compiler allocates those AGPRs itself ("=a" / "a" constraints), may move a parked value
around a cold high-pressure block (it does, legally, around the factor and the termination-check blocks) and uses the rest
of the accumulation file as spill space for architectural registers.  The one build of this kernel on record that computed
wrong results (docs/HISTORY.md 6b) had run out of AGPRs and spilled to scratch while AccD values were live.  Checked per
mpc_solve_kernel instantiation:
  (a) no scratch_ instruction anywhere in the kernel;
  (b) every AGPR an asm block reads is written by some asm block of the kernel;
  (c) listings built with -DQRW_MARK_PHASES only: on the hot path of an ADMM iteration (phase markers 0 .. 6: right-hand
      side, force elimination, sweeps, Delta^-1 products, updates) no compiler-generated instruction writes an AGPR that an
      asm block of that region reads, i.e. the Delta^-1 rows sit untouched in their registers from one factorisation to the
      next, and no scratch access happens there.
What this does and does not show: every build that has passed the GPU parity tests satisfies (a)-(c); the wrong build of
commit 9ff2d51 (max-ilp + phase counters) violates (a) (96 scratch instructions, none on the hot path) but, rebuilt with
phase markers, not (c) -- its defect was never located, (c) is a necessary condition, not a proof of a correct allocation.
Exit status 1 and one line per finding if a check fails.  Usage: python scripts/isa_accd_scan.py file.s [kernel-substring]"""
import re
import sys


def kernels(text):
    for m in re.finditer(r"^(_ZN3qrw16mpc_solve_kernel\w+):[^\n]*\n", text, re.M):
        start = m.end()
        end = text.index(".Lfunc_end", start)
        yield m.group(1), text[start:end].split("\n")


def agprs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^a\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^a(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _walk(lines, lo, hi):
    """(asm-read AGPRs, asm-written AGPRs, compiler-generated AGPR writes [(line, set, text)], scratch instructions) of lines[lo:hi]"""
    in_asm = False
    reads, writes, cw, scratch = set(), set(), [], []
    for i in range(lo, hi):
        raw = lines[i]
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        parts = t.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        if "scratch_" in op:
            scratch.append((i, t))
        if in_asm:
            if op == "v_accvgpr_read_b32" and len(ops) > 1:
                reads |= agprs(ops[1])
            elif op == "v_accvgpr_write_b32" and ops:
                writes |= agprs(ops[0])
            continue
        dst = set()
        if op in ("v_accvgpr_write_b32", "v_accvgpr_mov_b32") and ops:
            dst = agprs(ops[0])
        elif op.startswith(("ds_read", "ds_load", "global_load", "scratch_load", "buffer_load", "flat_load", "v_mfma")) and ops:
            dst = agprs(ops[0])
        if dst:
            cw.append((i, dst, t))
    return reads, writes, cw, scratch


def scan(name, lines):
    findings = []
    reads, writes, cw, scratch = _walk(lines, 0, len(lines))
    for i, t in scratch[:4]:
        findings.append("%s: scratch instruction (listing line +%d): %s" % (name, i, t))
    if len(scratch) > 4:
        findings.append("%s: ... %d scratch instructions in all" % (name, len(scratch)))
    never = reads - writes
    if never:
        findings.append("%s: asm blocks read AGPRs no asm block writes: %s" % (name, sorted(never)[:8]))
    stats = dict(asm_read=len(reads), asm_written=len(writes), compiler_agpr_writes=len(cw), scratch=len(scratch), hot=None)
    marks = [(i, int(re.search(r"QRW_PHASE (\d+)", l).group(1))) for i, l in enumerate(lines) if "QRW_PHASE" in l]
    if marks:
        p0 = [i for i, p in marks if p == 0]
        lo = p0[0] if p0 else None
        hi = next((i for i, p in marks if p == 6 and lo is not None and i > lo), None)
        if lo is None or hi is None:
            findings.append("%s: phase markers 0 / 6 not found in order" % name)
        else:
            hr, _, hcw, hsc = _walk(lines, lo, hi)
            for i, d, t in hcw:
                if d & hr:
                    findings.append("%s: compiler-generated write to AGPR(s) %s that the hot path's asm blocks read (listing line +%d): %s"
                                    % (name, sorted(d & hr)[:4], i, t))
            for i, t in hsc:
                findings.append("%s: scratch access on the hot path (listing line +%d): %s" % (name, i, t))
            stats["hot"] = dict(lines=hi - lo, asm_read=len(hr), compiler_agpr_writes=len(hcw))
            if len(hr) < 72:
                findings.append("%s: the hot path reads only %d AGPRs through asm blocks (72 expected: the Delta^-1 rows)" % (name, len(hr)))
    return findings, stats


def scan_file(path, want=""):
    out = []
    text = open(path).read()
    for name, lines in kernels(text):
        if want and want not in name:
            continue
        f, st = scan(name, lines)
        out.append((name, f, st))
    return out


if __name__ == "__main__":
    bad = 0
    for name, f, st in scan_file(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        print("%s: %s, findings %d" % (name, st, len(f)))
        for x in f[:10]:
            print("   " + x)
        bad += len(f)
    sys.exit(1 if bad else 0)
