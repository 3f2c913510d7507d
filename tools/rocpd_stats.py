#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes
DIR/NAME_results.db on ROCm 7): the same columns as the *_kernel_stats.csv of older rocprofv3 releases.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]
"""
import csv
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60).stdout.splitlines()
        if len(out) == len(names):
            return out
    except Exception:
        pass
    return names


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(f"""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
                              max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.private_segment_size), max(d.group_segment_size)
                              from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""))
    names = demangle([r[0].removesuffix(".kd") for r in rows])
    tot = sum(r[2] for r in rows) or 1
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR", "SGPR", "ScratchBytes", "LDSBytes")]
    for n, r in zip(names, rows):
        out.append((n, r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5], r[6], r[7], r[8], r[9], r[10]))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(out)
    for row in out[1:]:
        print(f"{row[1]:5d} x {row[3] / 1e3:10.1f} us  {row[4]:6.2f}%  v{row[7]} a{row[8]} s{row[9]} scr{row[10]} lds{row[11]}  {row[0][:140]}")


if __name__ == "__main__":
    main()
