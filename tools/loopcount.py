#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from hipcc's device assembly (build container, no GPU):

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on --cuda-device-only -S -o unit.s csrc/inst_f64_d4.hip
    tools/loopcount.py unit.s 'k_fs_aIdLi4ELi4E' [--dump]

For every backward branch (label .. branch) it prints the instruction counts between the label and the branch, by class:
VALU (v_*, incl. v_mfma), of which fp64 (v_*_f64), transcendental / quarter-rate (rcp / rsq / sqrt / log / exp / sin / cos / mul_hi / mul_lo_u32),
SALU, LDS (ds_*), VMEM (global_ / buffer_ / flat_ / scratch_), branches, waitcnts.  The loop bodies of the streaming passes are straight
line code (`#pragma unroll 1` time loops), so the VALU count of the innermost loop IS `SQ_INSTS_VALU` per wave-step, which is how the
per-chain-step instruction budgets in DESIGN.md are tracked between GPU runs."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


QUARTER = re.compile(r"^v_(rcp|rsq|sqrt|log|exp|sin|cos)_|^v_mul_(hi|lo)_[ui]32|^v_mad_[ui]64_[ui]32|^v_div_(scale|fmas|fixup)_f64|^v_(fma|mul|add)_f64$")


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z0-9.$]+:", l) and re.search(pat, l) and not l.startswith("."):
            start = i
            break
    if start is None:
        sys.exit(f"no kernel matching {pat}")
    end = start
    while end < len(lines) and not lines[end].strip().startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    labels, insts = {}, []
    for l in body:
        s = l.split(";")[0].strip()
        if not s or s.startswith((".", "//")) and not re.match(r"^\.LBB\d+_\d+:", s):
            if not re.match(r"^\.LBB\d+_\d+:", s):
                continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if re.match(r"^[_A-Za-z0-9.$]+:", s):
            continue
        op = s.split()[0]
        insts.append((op, s))
    total = {}
    for op, _ in insts:
        total[classify(op)] = total.get(classify(op), 0) + 1
    print(f"kernel {lines[start].split(':')[0]}: {len(insts)} instructions {total}")
    for idx, (op, s) in enumerate(insts):
        if classify(op) == "branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                seg = insts[labels[tgt]:idx + 1]
                c = {}
                f64 = tr = 0
                for o, _ in seg:
                    k = classify(o)
                    c[k] = c.get(k, 0) + 1
                    if k == "valu" and "_f64" in o:
                        f64 += 1
                    if k == "valu" and re.match(r"^v_(rcp|rsq|sqrt|log|exp|sin|cos)_|^v_mul_(hi|lo)_[ui]32|^v_mad_[ui]64_[ui]32", o):
                        tr += 1
                inner = sum(1 for o, t in seg[:-1] if classify(o) == "branch")
                print(f"  loop {tgt} [{labels[tgt]}..{idx}] {len(seg)} insts: valu {c.get('valu', 0)} (f64 {f64}, quarter-rate {tr}) salu {c.get('salu', 0)} "
                      f"lds {c.get('lds', 0)} vmem {c.get('vmem', 0)} branch {c.get('branch', 0)} wait {c.get('wait', 0)}  inner branches {inner}")
                if "--blocks" in sys.argv:
                    inv = {v: k for k, v in labels.items()}
                    cur, n, first = tgt, 0, labels[tgt]
                    for q in range(labels[tgt], idx + 1):
                        if q in inv and q != first:
                            print(f"      block {cur:12s} valu {n}")
                            cur, n, first = inv[q], 0, q
                        o, t = insts[q]
                        if classify(o) == "valu":
                            n += 1
                        if classify(o) == "branch":
                            print(f"      block {cur:12s} valu {n}   -> {t}")
                            cur, n = cur + "'", 0
                    print(f"      block {cur:12s} valu {n}")
                if dump:
                    hist = {}
                    for o, _ in seg:
                        hist[o] = hist.get(o, 0) + 1
                    for o, n in sorted(hist.items(), key=lambda kv: -kv[1]):
                        print(f"      {n:5d} {o}")


if __name__ == "__main__":
    main()
