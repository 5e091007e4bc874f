"""Scan a gfx950 ISA listing for hazards the compiler cannot see inside inline asm: a DPP / permlane source register
written by a VALU instruction fewer than 2 wait states earlier, and EXEC written by a VALU instruction fewer than 5
wait states before a DPP instruction.  Usage: python scripts/isa_hazard_scan.py file.s"""
import re, sys
def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'^-?\|?([va])\[(\d+):(\d+)\]\|?$', tok)
    if m: return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'^-?\|?([va])(\d+)\|?$', tok)
    if m: return {(m.group(1), int(m.group(2)))}
    return set()
lines = open(sys.argv[1]).read().splitlines()
hist = []  # (wait_states_since, written regs, text, writes_exec)
found = 0
for ln, l in enumerate(lines, 1):
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':') or t.startswith('#'): continue
    parts = t.replace(',', ' ').split()
    op = parts[0]
    if op == 's_nop':
        n = int(parts[1]) + 1
        hist = [(w + n, r, x, e) for (w, r, x, e) in hist]
        continue
    is_valu = op.startswith('v_')
    ops = parts[1:]
    # source registers that are read across lanes
    cross = set()
    if 'dpp' in op or 'row_newbcast' in t or 'quad_perm' in t or 'row_shr' in t or 'row_ror' in t:
        if len(ops) >= 2: cross = regs(ops[1])
        for (w, r, x, e) in hist:
            if e and w < 5:
                print("%s:%d EXEC written %d wait states before DPP: %s  <- %s" % (sys.argv[1], ln, w, t, x)); found += 1
    if op.startswith('v_permlane'):
        cross = regs(ops[0]) | regs(ops[1])
    for (w, r, x, e) in hist:
        if w < 2 and (r & cross):
            print("%s:%d cross-lane read %d wait states after write: %s  <- %s" % (sys.argv[1], ln, w, t, x)); found += 1
    hist = [(w + 1, r, x, e) for (w, r, x, e) in hist if w + 1 < 6]
    if is_valu:
        wr = regs(ops[0]) if ops else set()
        if op.startswith('v_permlane') and len(ops) > 1: wr |= regs(ops[1])
        we = op.startswith('v_cmpx') or (len(ops) > 0 and ops[0] == 'exec')
        hist.append((0, wr, t, we))
    elif op.startswith(('s_', 'ds_', 'global_', 'scratch_', 'buffer_', 'flat_')):
        pass
print("hazards found:", found)
