#!/usr/bin/env python3
"""SQ counter passes of the C2 headline command (tools/pmc_sq.sh <tag> -> gpurun_out/sq_<tag>/p*/.../*counter_collection.csv) -> the per-pass table committed as
profiles/r04_<x>_sq_counters_c2_fused.txt and the `fused_C2_sq_pass_*` entries of profiles/r04_traffic.json that bench.py prints as roofline.valu_issue:
VALU instructions per chain-step (SQ_INSTS_VALU / wave-steps; a wave is 64 chains), issue fraction SQ_INSTS_VALU x 4 cycles / (SQ_BUSY_CU_CYCLES x 4 SIMDs), mean
waves per SIMD (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SIMDs), core clock during the kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration).
usage: sq_fused_summary.py <sq dir> <out txt> [--chains 256 --T 65536] [--json profiles/r04_traffic.json]"""
import collections
import csv
import glob
import json
import re
import subprocess
import sys

src, out_txt = sys.argv[1], sys.argv[2]
C = int(sys.argv[sys.argv.index("--chains") + 1]) if "--chains" in sys.argv else 256
T = int(sys.argv[sys.argv.index("--T") + 1]) if "--T" in sys.argv else 65536
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else "profiles/r04_traffic.json"
vals = collections.defaultdict(dict)
lines = []
for f in sorted(glob.glob(f"{src}/p*/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for r in rows:
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    names = list(acc)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        d = re.sub(r"\(.*$", "", re.sub(r"^void ax::", "", d))
        if not re.match(r"k_fs_(ac|e|head|accept)<|k_aff_aggs<|k_fs_esfix<", d):
            continue
        for cn, (k, tot, dur) in sorted(acc[n].items()):
            vals[d][cn] = tot / k
            vals[d].setdefault("dur_us", dur / k)
            lines.append(f"{d:64s} {cn:24s} calls={k:4d} mean={tot / k:16.1f} dur_us={dur / k:10.1f}")
open(out_txt, "w").write("\n".join(lines) + "\n")
try:
    out = json.load(open(jpath))
except Exception:
    out = {}
wave_steps = C * T / 64.0
for name, key in (("k_fs_ac<", "pass_AC"), ("k_fs_e<", "pass_E")):
    k = next((q for q in vals if q.startswith(name)), None)
    if not k:
        continue
    v = vals[k]
    ent = dict(kernel=k, source=f"{out_txt} (tools/pmc_sq.sh: separate rocprofv3 --pmc passes of the headline command; issue fraction = SQ_INSTS_VALU x 4 cycles / "
                                "(SQ_BUSY_CU_CYCLES x 4 SIMDs))", us_per_launch=round(v["dur_us"], 1))
    if "SQ_INSTS_VALU" in v:
        ent["valu_per_chain_step"] = round(v["SQ_INSTS_VALU"] / wave_steps, 1)
    if "SQ_INSTS_VALU" in v and "SQ_BUSY_CU_CYCLES" in v:
        ent["valu_issue_frac"] = round(v["SQ_INSTS_VALU"] * 4.0 / (v["SQ_BUSY_CU_CYCLES"] * 4.0), 3)
    if "SQ_WAVE_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
        ent["waves_per_simd"] = round(v["SQ_WAVE_CYCLES"] / v["SQ_BUSY_CYCLES"] / 8.0, 2)
    if "GRBM_GUI_ACTIVE" in v:
        ent["core_clock_GHz"] = round(v["GRBM_GUI_ACTIVE"] / 8.0 / (v["dur_us"] * 1e3), 2)
    out[f"fused_C2_sq_{key}"] = ent
    print(key, ent)
json.dump(out, open(jpath, "w"), indent=1, sort_keys=True)
