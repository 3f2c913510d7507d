#!/usr/bin/env python3
"""Per-kernel table of several PMC counters from a rocprofv3 counter_collection.csv (mean over dispatches)."""
import csv, re, subprocess, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    d = dur[r["Kernel_Name"]]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = sorted(agg, key=lambda k: -dur[k][1])
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
ctrs = sorted({c for k in agg for c in agg[k]})
print(f"{'kernel':46s} {'dur_us':>8s} " + " ".join(f"{c[-16:]:>16s}" for c in ctrs))
for n, d in zip(names[:int(sys.argv[2]) if len(sys.argv) > 2 else 12], dem):
    d = re.sub(r"^void ax::", "", d); d = re.sub(r"\(.*$", "", d)
    print(f"{d[:46]:46s} {dur[n][1]/dur[n][0]:8.1f} " + " ".join(f"{agg[n][c][1]/max(agg[n][c][0],1):16.4g}" for c in ctrs))
