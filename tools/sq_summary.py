#!/usr/bin/env python3
"""SQ counter passes of the cSMC kernels (tools/pmc_sq_c3.sh -> summary.txt) -> the VALU-issue figures bench.py reports beside the HBM roofline:
  python tools/sq_summary.py gpurun_out/sqc3_x/summary.txt <T> <chains> <N> [--out profiles/r03_traffic.json]
per kernel: VALU / SALU / LDS / branch instructions per wave and time step, and the VALU-issue fraction SQ_INSTS_VALU x 4 cycles / (SQ_BUSY_CU_CYCLES x 4 SIMDs)
(a wave64 VALU instruction occupies its SIMD for four cycles)."""
import json, re, sys
src, T, C, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "profiles/r03_traffic.json"
vals = {}
for l in open(src):
    m = re.match(r"(k_csmc_(?:fwd|bwd))<.*?>\s+(\S+)\s+calls=\s*\d+\s+mean=\s*([\d.]+)\s+dur_us=\s*([\d.]+)", l)
    if m:
        vals.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
        vals[m.group(1)]["dur_us"] = float(m.group(4))
try:
    out = json.load(open(out_path))
except Exception:
    out = {}
wave_steps = C * ((N + 63) // 64) * T
for k, v in vals.items():
    ent = dict(source=src, T=T, chains=C, N=N, us_per_launch=v.get("dur_us"))
    for name, key in (("valu_per_wave_step", "SQ_INSTS_VALU"), ("salu_per_wave_step", "SQ_INSTS_SALU"), ("lds_per_wave_step", "SQ_INSTS_LDS"),
                      ("branch_per_wave_step", "SQ_INSTS_BRANCH")):
        if key in v:
            ent[name] = round(v[key] / wave_steps, 1)
    if "SQ_INSTS_VALU" in v and "SQ_BUSY_CU_CYCLES" in v:
        ent["valu_issue_frac"] = round(v["SQ_INSTS_VALU"] * 4.0 / (v["SQ_BUSY_CU_CYCLES"] * 4.0), 3)
    out[f"csmc_C3_sq_{k}"] = ent
    print(k, ent)
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
