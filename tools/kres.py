#!/usr/bin/env python3
"""Tabulate hipcc -Rpass-analysis=kernel-resource-usage output for one .hip file: VGPRs, AGPRs, scratch, occupancy."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?)(?: \[-Rpass)", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s}")
for r, d in zip(rows, demangle):
    d = re.sub(r"^void ax::", "", d); d = re.sub(r"\(.*$", "", d)
    print(f"{d[:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>6s}")
