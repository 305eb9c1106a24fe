"""Inter-kernel gaps of a rocprofv3 --kernel-trace run: for every kernel of the trace (ordered by start), gap = start - max(end of all earlier
kernels).  Prints, for the densest window of the run (the graph-replayed sampler steps), busy time, idle time and the gap histogram.
usage: python tools/trace_gaps.py <dir with *kernel_trace.csv> [min_kernels_per_window]"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
print(json.dumps({"kernels_in_trace": len(rows)}))
# steady-state window: the last 40 % of the trace's kernels (sampler steps replayed from the graph come last before the VAE / roofline passes
# when bench.py runs with --no-roofline --no-vae)
lo, hi = int(len(rows) * 0.3), int(len(rows) * 0.95)
win = rows[lo:hi]
busy = sum(e - s for s, e, _ in win)
span = win[-1][1] - win[0][0]
gaps = []
last_end = win[0][1]
for s, e, n in win[1:]:
    gaps.append(max(0, s - last_end))
    last_end = max(last_end, e)
gaps_sorted = sorted(gaps)
q = lambda p: gaps_sorted[int(p * (len(gaps_sorted) - 1))]
out = {"window_kernels": len(win), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "idle_ms": sum(gaps) / 1e6, "idle_frac": sum(gaps) / span,
       "gap_us": {"median": q(0.5) / 1e3, "p10": q(0.1) / 1e3, "p90": q(0.9) / 1e3, "p99": q(0.99) / 1e3, "max": gaps_sorted[-1] / 1e3},
       "mean_kernel_us": busy / len(win) / 1e3}
print(json.dumps(out))
# by kernel family: mean duration and mean gap BEFORE the kernel
fam = {}
last_end = win[0][1]
for s, e, n in win[1:]:
    k = n.split("(")[0][-60:]
    d = fam.setdefault(k, [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    d[2] += max(0, s - last_end)
    last_end = max(last_end, e)
for k, (c, dur, gp) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
    print(json.dumps({"kernel": k, "count": c, "mean_us": round(dur / c / 1e3, 2), "mean_gap_before_us": round(gp / c / 1e3, 2)}))
