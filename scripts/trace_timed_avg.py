"""Average duration of the TIMED launches of each qrw kernel in a rocprofv3 kernel-trace CSV: the first `warmup`
launches are the bench's warm-up steps (the very first full-batch one sets the QPs up from a cold start; before it
comes qrw_create's one-instance self-test solve) and are left out: the LAST `K` launches of every kernel are the timed
ones, so the figure is comparable with bench.py's own HIP-event average.
Usage: python scripts/trace_timed_avg.py kernel_trace.csv K"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
by = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    name = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
    if not name.startswith("qrw::"):
        continue
    by.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for name, d in by.items():
    t = d[-K:]
    print("%-28s launches %3d (timed %3d)  all-launch avg %.4f ms  timed-launch avg %.4f ms  min %.4f  max %.4f" % (
        name, len(d), len(t), sum(d) / len(d), sum(t) / len(t), min(t), max(t)))
