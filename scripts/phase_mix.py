"""Static instruction mix per phase of the MPC kernel's ADMM loop (FULL N=16 instantiation), from an ISA listing
compiled with -DQRW_MARK_PHASES.  Usage: python scripts/phase_mix.py"""
import collections, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(R, "build", "mpc_mark.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-DQRW_MARK_PHASES", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Wno-pass-failed",
                       "-I" + R + "/include", "-I" + R + "/quadruped-reactive-walking_amd/csrc", "-Wno-unused-value",
                       "-Wno-unused-result", "-Wno-unused-function", R + "/quadruped-reactive-walking_amd/csrc/mpc_kernel.hip", "-o", out])
txt = open(out).read()
k = txt[txt.index("_ZN3qrw16mpc_solve_kernelILi1ELb1ELb0ELb0EEEvNS_7MpcArgsE:"):]
k = k[:k.index("s_endpgm")]
names = {9: "loop head (before factor)", 0: "factor", 1: "rhs", 2: "elim_g", 3: "fwd_chain", 4: "middle", 5: "bwd_chain",
         6: "backsub+A+upd", 7: "exit"}
cur, mix = None, collections.OrderedDict()
def cls(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("v_") and ("f64" in op or "_b64" in op and "mov" not in op): return "valu64"
    if op.startswith("v_"): return "valu32"
    if op.startswith("s_"): return "salu"
    return "other"
for line in k.splitlines():
    m = re.search(r"QRW_PHASE (\d+)", line)
    if m:
        cur = int(m.group(1)); mix.setdefault(cur, collections.Counter()); continue
    t = line.strip()
    if not t or t.startswith((";", ".", "#")) or t.endswith(":") or cur is None: continue
    mix[cur][cls(t.split()[0])] += 1
print("phase ends at marker -> instructions listed BEFORE the marker belong to the phase named by the NEXT marker's id")
for ph, c in mix.items():
    print("after marker %d (%s): total %d  %s" % (ph, names.get(ph, "?"), sum(c.values()), dict(c)))
