#!/usr/bin/env python3
"""one line per bench.py log: value, ms/step, the on-path launch groups (ms)"""
import json, sys
for p in sys.argv[1:]:
    for l in open(p):
        if l.startswith("{"):
            d = json.loads(l)
            ks = {k: v["ms_per_step"] for k, v in d.get("kernels", {}).items()}
            print(f"{p}: {d['value']:.0f} {d['unit']}  {d['ms_per_step']:.3f} ms  acc={d.get('accept_rate')} |la|={d.get('max_abs_log_alpha')}  " +
                  " ".join(f"{k}={v:.3f}" for k, v in ks.items()))
            gp = d.get("general_path")
            if gp:
                print(f"    general: {gp['value']:.0f}  {gp['ms_per_step']:.3f} ms")
