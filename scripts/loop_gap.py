"""From a rocprofv3 kernel trace of scripts/gpu_loop_kernels.py: the gap on the device between the end of control_pre_quad_kernel
and the start of the wbc16_kernel that follows it (what fusing the two launches of a non-solving iteration could save at most)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gaps, pre, wbc = [], [], []
for a, b in zip(rows, rows[1:]):
    if "control_pre_quad_kernel" in a["Kernel_Name"] and "wbc16_kernel" in b["Kernel_Name"]:
        gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
        pre.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"])); wbc.append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
g = np.array(gaps) / 1e3
print("pairs %d: control_pre_quad %.1f us, wbc16 %.1f us (medians); gap between them: median %.2f us, p10 %.2f, p90 %.2f" % (len(g), np.median(pre) / 1e3, np.median(wbc) / 1e3, np.median(g), np.percentile(g, 10), np.percentile(g, 90)))
