#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP unit from the compiler's own remarks
(hipcc -Rpass-analysis=kernel-resource-usage).   python tools/kernel_resources.py csrc/inst_f64_d4.hip [filter-substring ...]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aux_ssm_samplers_amd", "csrc")
KEYS = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")


def resources(src, extra_flags=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", "/dev/null", "-I", CSRC] + list(extra_flags)
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+(.+?): (\d+)", line)
        if m and cur and m.group(1) in KEYS:
            out[cur][m.group(1)] = int(m.group(2))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60).stdout.splitlines()
        return r if len(r) == len(names) else names
    except Exception:
        return names


def table(unit):
    """{demangled kernel or device-function name: resources} of one unit of csrc/"""
    src = os.path.join(CSRC, unit)
    res = resources(src, ["-ffp-contract=off"] if unit in ("csmc.hip", "pit.hip", "loop.hip") else [])
    names = list(res)
    return dict(zip(demangle(names), (res[n] for n in names)))


if __name__ == "__main__":
    if sys.argv[1] == "--json":  # python tools/kernel_resources.py --json wide.hip > profiles/r03_wide_resources.json
        import json
        print(json.dumps(table(sys.argv[2]), indent=1, sort_keys=True))
        sys.exit(0)
    src = sys.argv[1]
    if not os.path.exists(src):
        src = os.path.join(CSRC, os.path.basename(src))
    flt = sys.argv[2:]
    res = resources(src, ["-ffp-contract=off"] if os.path.basename(src) in ("csmc.hip", "pit.hip", "loop.hip") else [])
    names = list(res)
    for n, dn in zip(names, demangle(names)):
        if flt and not all(f in dn for f in flt):
            continue
        v = res[n]
        print(f"v{v.get('VGPRs', 0):3d} a{v.get('AGPRs', 0):3d} s{v.get('TotalSGPRs', 0):3d} scratch{v.get('ScratchSize [bytes/lane]', 0):5d} occ{v.get('Occupancy [waves/SIMD]', 0)} "
              f"spill(s{v.get('SGPRs Spill', 0)},v{v.get('VGPRs Spill', 0)}) lds{v.get('LDS Size [bytes/block]', 0):6d}  {dn[:150]}")
