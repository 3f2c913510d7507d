#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel and counter the mean RAW value per launch and the mean duration."""
import csv, re, subprocess, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    a = agg[(r["Kernel_Name"], r["Counter_Name"])]
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = sorted(agg)
dem = subprocess.run(["c++filt"], input="\n".join(n for n, _ in names), capture_output=True, text=True).stdout.splitlines()
for (n, c), d in zip(names, dem):
    a = agg[(n, c)]
    d = re.sub(r"^void ax::", "", d); d = re.sub(r"\(.*$", "", d)
    print(f"{d[:64]:64s} {c:24s} calls={a[0]:4d} mean={a[1]/a[0]:16.1f} dur_us={a[2]/a[0]:10.1f}")
