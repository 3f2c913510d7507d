#!/usr/bin/env python3
"""Pretty-print a rocprofv3 *_kernel_stats.csv (demangled, per-kernel avg duration)."""
import csv, re, subprocess, sys
rows = list(csv.DictReader(open(sys.argv[1])))
dem = subprocess.run(["c++filt"], input="\n".join(r["Name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"^void ax::", "", d); d = re.sub(r"\(.*$", "", d)
    print(f"{d[:64]:64s} calls={r['Calls']:>6s} total_us={float(r['TotalDurationNs'])/1e3:11.1f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):6.2f}")
