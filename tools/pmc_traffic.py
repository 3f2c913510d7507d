#!/usr/bin/env python3
"""HBM traffic per launch group from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as the guide prescribes).

  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <prefix> [--config c2|c3] [--out profiles/r02_traffic.json]

Units / corrections (MI355X_MICROARCH.md, "HBM [CDNA4]"): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests
of wide coalesced reads at 64 bytes, so it is doubled (calibrated in round 1 on k_scan_reduce_cm, which reads exactly its element buffer);
WRITE_SIZE is taken as is.  Per kernel: mean over its dispatches.  Per launch group: sum over the group's kernels of mean bytes x
dispatches per sweep (dispatches / number of sweeps of that mode in the run).  Keys match bench.py's pmc_traffic() lookups."""
import collections
import csv
import json
import re
import subprocess
import sys

# launch groups (ctx.h ProfScope ids) -> kernels, per mode.  A kernel used by both modes (k_rng_sweep, k_select, ...) has the same
# bytes per dispatch in both; dispatches are split by the modes' sweep counts.
SHARED = {
    "filter_tab": [r"^k_filter_t0<", r"^k_filter_init<", r"^k_scan_reduce<ax::FilterOp<", r"^k_scan_aggs<ax::FilterOp<", r"^k_scan_down<ax::FilterOp<", r"^k_gain_tab<", r"^k_mask_obs<", r"^k_copy_cov<"],
    "filter_scan": [r"^k_aff_chunkprod<ax::FilterMeanOp<", r"^k_aff_reduce<ax::FilterMeanOp<", r"^k_aff_down<ax::FilterMeanOp<"],
    "sample_scan": [r"^k_sample_shared_tab<", r"^k_aff_chunkprod<ax::SampleAffOp<", r"^k_aff_reduce<ax::SampleAffOp<", r"^k_aff_down<ax::SampleAffOp<"],
    "logpdf": [r"^k_sweep_logpdf_tab<", r"^k_sweep_logpdf_cm_shared<"],
}
GENERAL = {
    "filter_scan": [r"^k_scan_reduce_cm<ax::FilterOpFly<", r"^k_scan_down_cm<ax::FilterOpFly<"],
    "sample_scan": [r"^k_scan_reduce_cm<ax::SampleOpFly<", r"^k_scan_down_cm<ax::SampleOpFly<"],
    "logpdf": [r"^k_sweep_logpdf_cm<", r"^k_sweep_logpdf_cm_semi<", r"^k_sweep_logpdf_tab<"],
    "filter_tab": [r"^k_obs_info_tab<"],
}
# round 4: the chain-shared sweep in TWO streaming passes (csrc/fused_shared.h: k_fs_ac, k_fs_e) + the two aggregate scans and the Psi completion between them; the
# model stage is memoised (its kernels return at once in the steady state: their bytes here are what the few REBUILT stages of the profiled run moved, per sweep)
STAGE_FILTER = [r"^k_filter_t0<", r"^k_filter_init<", r"^k_scan_reduce<ax::FilterOp", r"^k_scan_aggs<ax::FilterOp", r"^k_scan_down<ax::FilterOp", r"^k_ks_tile<ax::FilterOp",
                r"^k_ks_down<ax::FilterOp", r"^k_gain_tab<", r"^k_mask_obs<", r"^k_copy_cov<", r"^k_memo_"]
FUSED = {
    "filter_tab": STAGE_FILTER,
    "sample_init": [r"^k_sample_shared_tab<", r"^k_sweep_logpdf_tab<", r"^k_fs_fprod<", r"^k_fs_gpre<", r"^k_fs_psi<", r"^k_fs_rows<", r"^k_fs_clog"],
    "filter_scan": [r"^k_fs_ac<"],
    "sample_scan": [r"^k_aff_aggs<", r"^k_fs_esfix<"],
    "logpdf": [r"^k_fs_e<"],
    "select": [r"^k_fs_head<", r"^k_fs_accept<"],
    "factory": [r"^k_concat_model<", r"^k_fs_concat0<"],
    "rng": [r"^k_rng_sweep<"],
}
BOTH = {"rng": [r"^k_rng_sweep<"], "select": [r"^k_select(_rows)?<", r"^k_accept<"], "factory": [r"^k_concat_model<", r"^k_concat_obs<"]}
CSMC = {"csmc_fwd": [r"^k_csmc_fwd<"], "csmc_bwd": [r"^k_csmc_bwd<"], "csmc_ctrans": [r"^k_csmc_ctrans<"]}


def per_kernel(path):
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    names = list(acc)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    dem = [re.sub(r"^void ax::", "", d) for d in dem]
    return {d: (acc[n][0], acc[n][1] / acc[n][0] * 1024.0) for n, d in zip(names, dem)}  # name -> (dispatches, mean bytes)


def group_bytes(fetch, write, pats, sweeps):
    tot_f = tot_w = 0.0
    kern = {}
    for name in fetch:
        wgt = [p[1] if isinstance(p, tuple) else 1.0 for p in pats if re.search(p[0] if isinstance(p, tuple) else p, name)]
        if wgt:
            n, fb = fetch[name]
            wb = write.get(name, (0, 0.0))[1]
            per_sweep = n / sweeps * wgt[0]
            tot_f += 2.0 * fb * per_sweep
            tot_w += wb * per_sweep
            kern[re.sub(r"\(.*$", "", name)] = dict(dispatches_per_sweep=round(per_sweep, 3), fetch_bytes=int(2 * fb), write_bytes=int(wb))
    return tot_f, tot_w, kern


def mfma_busy(path, key, out_path):
    """MFMA busy fraction per wide kernel: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), summed over the dispatches of the pass"""
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    names = [n for n in acc if "wide" in n]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    try:
        out = json.load(open(out_path))
    except Exception:
        out = {}
    ent = {}
    for n, d in zip(names, dem):
        d = re.sub(r"\(.*$", "", re.sub(r"^void ax::", "", d))
        cu = acc[n].get("SQ_BUSY_CU_CYCLES", 0.0)
        if cu > 0:
            ent[d] = round(acc[n].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * cu), 4)
    out[key] = dict(mfma_busy=ent, source=f"{path}: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), all dispatches of the pass")
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(key, ent)


def main():
    if "--mfma" in sys.argv:  # python tools/pmc_traffic.py --mfma <counter_collection.csv> <key> [--out ...]
        i = sys.argv.index("--mfma")
        return mfma_busy(sys.argv[i + 1], sys.argv[i + 2], sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "profiles/r04_traffic.json")
    fetch, write, prefix = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "c2"
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "profiles/r04_traffic.json"
    src = f"{sys.argv[1]} + {sys.argv[2]} (FETCH_SIZE x2 per the gfx950 rule, + WRITE_SIZE; mean per dispatch x dispatches per sweep)"
    try:
        out = json.load(open(out_path))
    except Exception:
        out = {}
    if cfg == "c2":
        cnt = lambda pre: next((n for k, (n, _) in fetch.items() if re.match(pre, k)), 0)
        n_sh, n_ge, n_fu = cnt(r"k_sweep_logpdf_cm_shared<"), cnt(r"k_sweep_logpdf_cm<") or cnt(r"k_sweep_logpdf_cm_semi<"), cnt(r"k_fs_e<")
        if n_fu:  # (a run of fused sweeps only: bench.py --no-general-leg)
            tot = 0
            for g, pats in FUSED.items():
                f, w, kern = group_bytes(fetch, write, pats, n_fu)
                if kern:
                    out[f"{prefix}_fused_{g}"] = dict(hbm_bytes=int(f + w), fetch_bytes=int(f), write_bytes=int(w), kernels=kern, source=src)
                    tot += int(f + w)
            out[f"{prefix}_fused_sweep"] = dict(hbm_bytes=tot, fetch_bytes=0, write_bytes=0, source=src + "; sum over the launch groups of one sweep")
        for mode, table, sweeps in (("shared", SHARED, n_sh), ("general", GENERAL, n_ge)):
            if not sweeps or (n_fu and mode == "shared"):
                continue
            for g, pats in list(table.items()) + list(BOTH.items()):
                f, w, kern = group_bytes(fetch, write, pats, sweeps if g not in BOTH else n_sh + n_ge + n_fu)
                if kern:
                    out[f"{prefix}_{mode}_{g}"] = dict(hbm_bytes=int(f + w), fetch_bytes=int(f), write_bytes=int(w), kernels=kern, source=src)
    else:
        sweeps = next(n for k, (n, _) in fetch.items() if k.startswith("k_csmc_bwd<"))
        tf = tw = 0.0
        for g, pats in CSMC.items():
            f, w, kern = group_bytes(fetch, write, pats, sweeps)
            if kern:
                out[f"{prefix}_{g}"] = dict(hbm_bytes=int(f + w), fetch_bytes=int(f), write_bytes=int(w), kernels=kern, source=src)
                tf, tw = tf + f, tw + w
        out[f"{prefix}_sweep"] = dict(hbm_bytes=int(tf + tw), fetch_bytes=int(tf), write_bytes=int(tw), source=src)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        if k.startswith(prefix):
            print(f"{k:70s} {out[k]['hbm_bytes'] / 1e9:9.3f} GB  (fetch {out[k]['fetch_bytes'] / 1e9:.3f}, write {out[k]['write_bytes'] / 1e9:.3f})")


if __name__ == "__main__":
    main()
