#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: every dispatch between two occurrences of an anchor kernel (default: the accept
step of the fused sweep) with its start offset, duration, queue and the idle gap on ITS queue since the previous dispatch; then the union busy time of the
anchor's queue.  usage: timeline.py <kernel_trace.csv> [anchor-substring] [which-step]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_fs_accept"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < 3:
    sys.exit(f"anchor {anchor} found {len(idx)} times")
a, b = idx[which - 1], idx[which]
t0 = rows[a]["e"]
mainq = rows[b]["Queue_Id"]
last = {}
print(f"step between dispatch {a} and {b}: {(rows[b]['e'] - t0) / 1e3:.1f} us; anchor queue {mainq}")
busy = 0
for r in rows[a + 1:b + 1]:
    q = r["Queue_Id"]
    gap = (r["s"] - last[q]) / 1e3 if q in last else (r["s"] - t0) / 1e3
    last[q] = r["e"]
    name = r["Kernel_Name"].split("(")[0].replace("void ax::", "")[:60]
    print(f"{(r['s'] - t0) / 1e3:9.1f} us  +{(r['e'] - r['s']) / 1e3:8.1f}  q{q:>3s} gap {gap:7.1f}  {name}")
    if q == mainq:
        busy += r["e"] - r["s"]
print(f"anchor queue busy {busy / 1e3:.1f} us of {(rows[b]['e'] - t0) / 1e3:.1f}")
