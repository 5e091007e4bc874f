"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per pass) into one JSON list:
per kernel and counter the number of dispatches, the mean over dispatches and the last dispatch's value."""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
acc = collections.OrderedDict()
grid = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        name = re.sub(r"<.*", "", row["Kernel_Name"]).replace("void ", "").split("(")[0]
        if not name.startswith("qrw::"):
            continue
        key = (name, row["Counter_Name"], row["Dispatch_Id"])
        per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])  # sums the per-XCD/SE instances
        grid[key] = int(row.get("Grid_Size", 0) or 0)
    for (name, ctr, did), v in per_dispatch.items():
        acc.setdefault((name, ctr), []).append((grid[(name, ctr, did)], v))
out = []
for (k, c), gv in acc.items():
    gmax = max(g for g, _ in gv)
    v = [x for g, x in gv if g == gmax]  # full-batch dispatches only (qrw_create's one-instance self-test solve is left out)
    out.append({"kernel": k, "counter": c, "dispatches": len(v), "mean": sum(v) / len(v), "last": v[-1]})
json.dump(out, sys.stdout, indent=1)
