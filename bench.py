#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/sec of the auxiliary-Kalman / conditional-SMC hot path on MI355X (BASELINE.json metric).

Headline (`value`, config.workload): BASELINE configs[1] = SURVEY 8(d) C2: linear-Gaussian SSM, T = 65536, d = 4 (p = 8 with the
auxiliary observations concatenated), parallel-in-time aux-Kalman sweep, fp64, `--chains` independent chains per GPU.  One "step" =
one sweep (kalman/generic.py:53-76) of every chain resident on this GPU: device Threefry draws, proposal LGSSM, filter scan, pathwise
sampler scan, log-densities, MH accept.  Inputs are resident in HBM before the timed region.  value = chains * n_gpus * steps /
seconds (independent chains: weak scaling, no data-path collective; torch.distributed (RCCL) is used only for the barrier, the
max-over-ranks and the final chain-gather).

On the same JSON line:
  roofline      the filter's scan (the parallel-in-time scan of the metric; the largest chain pass of the step), timed LIVE with HIP events on the library's stream inside the timed region
                (auxssm_prof_*, one event pair per launch of that group); `achieved` = the bytes THAT group has to move (its per-chain
                inputs read once + its outputs written once + chain-shared tables once; DESIGN.md section 4 lists the per-step reals of
                every group) / its average duration.  `traffic` = HBM bytes from the committed PMC passes of the same kernels, with the
                file it came from (null when no pass matches the kernels of this build).
  kernels       every group's average ms per step and its own algorithmic GB/s, from an UNTIMED profile pass of up to 3 steps right
                before the timed region (bracketing every group with events inside the timed region costs 2.7 % of the headline); the
                roofline group's entry is its live measurement.
  general_path  the same workload with nothing hoisted out of the chain loop (AUXSSM_OPT_SHARE_MODEL = 0: what every model with
                chain-specific dynamics runs), with its own `roofline` on the filter's associative scan, priced at SURVEY 8(d)'s K3
                bytes (the reference's element buffer in, filtered moments out).  This build never materialises the elements, so the
                PMC `traffic` is BELOW that figure.
  secondary     bounded legs on the other BASELINE configs, each with its own roofline: C3 cSMC (HBM), C4 Lorenz Kalman + cSMC with a
                FIXED total of 64 chains sharded over the ranks (parallel.shard_chains), C5 wide-state filter (MFMA flops).
  cpu_baseline  oracle/kalman_seq.c -- the reference's SEQUENTIAL sweep (its CPU code path) restated in C, chains over OpenMP threads
                -- on this box's host cores: 1 thread and all cores (rank 0, N = 1 only).

Launch: `python bench.py --gpus N ...`.  With N > 1 and no RANK in the environment this process starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py` as a CHILD (before anything touches the GPU) and exits with its
code; under a launcher (RANK set) WORLD_SIZE must equal --gpus.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak


# ------------------------------------------------------------------------------------------------------------------------
# launch
# ------------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=256, help="chains per GPU (SURVEY 8d lists 1, 8, 64, 256 for C2)")
    ap.add_argument("--T", type=int, default=65536)
    ap.add_argument("--d", type=int, default=4)
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket the kernel groups with HIP events")
    ap.add_argument("--no-share-model", action="store_true",
                    help="headline on the general per-chain path (AUXSSM_OPT_SHARE_MODEL = 0)")
    ap.add_argument("--no-general-leg", action="store_true", help="skip the extra run of the general per-chain path")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 / C5 legs")
    ap.add_argument("--secondary", default="c3,c4,c5,sv", help="comma list of secondary legs to run")
    ap.add_argument("--small-secondary", action="store_true", help="shrink the secondary legs (contract tests): not the BASELINE sizes")
    ap.add_argument("--workload", default="kalman", choices=["kalman", "csmc"],
                    help="kalman = BASELINE configs[1] (C2, the headline); csmc = configs[2] (C3) as the only workload")
    ap.add_argument("--N", type=int, default=1024, help="particles (csmc workload)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one rank per GPU) is what the driver runs; gloo + ranks sharing a GPU is a rehearsal mode")
    args = ap.parse_args(argv)
    if args.dtype is None:
        args.dtype = "f32" if args.workload == "csmc" else "f64"
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus >= 1, --steps >= 1, --warmup >= 0")
    return args


def free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launcher_command(args, argv):
    """The child command `--gpus N` (N > 1) turns into when no launcher set RANK: one rank per GPU over RCCL."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def resolve_launch(args, argv, environ):
    """('run', rank, world, local_rank) | ('spawn', cmd) | ('error', message).  Pure: tested on the CPU (tests/test_bench_launch.py)."""
    if "RANK" in environ:
        world = int(environ.get("WORLD_SIZE", "1"))
        if world != args.gpus:
            return ("error", f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: they must agree")
        return ("run", int(environ["RANK"]), world, int(environ.get("LOCAL_RANK", "0")))
    if args.gpus == 1:
        return ("run", 0, 1, 0)
    return ("spawn", launcher_command(args, argv))


# ------------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------------
def build_model(T, d, dtype):
    from aux_ssm_samplers_amd.workloads import lg_model
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    m = lg_model(T, d, dtype=dtype)
    bt = np.broadcast_to
    # time-varying arrays are materialised in full (T-1, d, d) as the reference's factories do (jnp.tile,
    # examples/stochastic_volatility/auxiliary_kalman.py:22-26); they are shared by all chains (vmap closure constants)
    full = lambda a, n: np.ascontiguousarray(bt(a, (n,) + a.shape))
    model = LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1),
                          full(m["Hobs"], T), full(m["Robs"], T), full(m["cobs"], T), m["y"])
    return m, model


def sv_data(T, seed=0):
    """Stochastic-volatility data of config C3 (examples/stochastic_volatility/model.py:11-53; nu=0, phi=.9, tau=2, d=1)."""
    phi, tau = 0.9, 2.0
    q = tau / (1 - phi ** 2)
    rng = np.random.Generator(np.random.PCG64(seed))
    e = rng.standard_normal(T)
    x = np.empty(T)
    x[0] = np.sqrt(q) * e[0]
    for t in range(1, T):
        x[t] = phi * x[t - 1] + np.sqrt(q) * e[t]
    y = np.exp(0.5 * x) * rng.standard_normal(T)
    return phi, q, x[:, None], y[:, None]


class Ctx:
    """rank / collective plumbing shared by the legs"""

    def __init__(self, args, rank, world, local_rank):
        import torch
        self.args, self.rank, self.world, self.torch = args, rank, world, torch
        ndev = torch.cuda.device_count()
        if args.dist_backend == "gloo":
            local_rank = local_rank % max(ndev, 1)  # rehearsal: ranks may share a GPU
        self.local_rank = local_rank
        self.dist = None
        launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run: initialise the group even for one rank
        torch.cuda.set_device(local_rank)
        if world > 1 or launched:
            import torch.distributed as dist
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")
            self.dist = dist
        self.coll_dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")
        from aux_ssm_samplers_amd import _lib
        self.lib = _lib
        self.handle = _lib.default_handle(local_rank)

    def barrier(self):
        self.handle.sync()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.handle.sync()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if self.dist is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v):
        if self.dist is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, step, steps, warmup, prof=True, focus=None):
        """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize pairs; max over ranks.
        Returns (seconds, {group: (launches, total ms)}) with the totals scaled to `steps` steps.
        Per-group times come from an UNTIMED profile pass (up to 3 extra steps with one HIP-event pair per launch group, right before the
        timed region): recording ~20 events per step inside the timed region costs 2.7 % of the headline (10 % at one chain).  Inside the
        timed region only ONE group is measured live -- `focus(per_step_ms) -> name` picks it (default: the slowest), it is the group
        `roofline` describes -- and its live measurement replaces the profile-pass figure."""
        for k in range(warmup):
            step(k)
        groups, live_id = {}, None
        if prof:
            pp = max(1, min(3, steps))
            self.barrier()
            self.handle.prof_enable(self.lib.K_ALL, 64 * (pp + 1))
            for k in range(pp):
                step(warmup + steps + k)  # its own step indices: a sampler must not see a key twice (the C4 leg replayed its warm-up keys here up to round 3, which
                                          # drove the chains of the reference's NaN-policy kernel into a corner where they reject: the 0.078 of BENCH_r03)
            self.barrier()
            prof_pass = self.handle.prof_read_groups()
            self.handle.prof_disable()
            groups = {g: (n * steps / pp, ms * steps / pp) for g, (n, ms) in prof_pass.items()}
            if groups:
                per_step = {g: ms / steps for g, (n, ms) in groups.items()}
                name = focus(per_step) if focus else max(per_step, key=per_step.get)
                if name in self.lib.K_NAMES:
                    live_id = self.lib.K_NAMES.index(name)
        self.barrier()
        if live_id is not None:
            self.handle.prof_enable(live_id, 8 * (steps + 1))
        self.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
        self.barrier()
        el = self.max_over_ranks(time.perf_counter() - t0)
        if live_id is not None:
            n, ms = self.handle.prof_read()
            self.handle.prof_disable()
            if n:
                groups[self.lib.K_NAMES[live_id]] = (n, ms)
                self.live_group = self.lib.K_NAMES[live_id]
        return el, groups

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ---- algorithmic bytes of the Kalman sweep's kernel groups (reals per chain and time step) -------------------
def kalman_group_reals(mode, d, po):
    """{group: (per-chain reals read, per-chain reals written, chain-shared reals per time step)} for one sweep of the LG-concat model.
    What each group HAS to move given its inputs and outputs (every input once, every output once); a kernel that re-reads shows up as
    traffic > algorithmic.  mode: 'shared' (chain-shared model parameters hoisted) or 'general'."""
    p = d + po
    sym = lambda n: n * (n + 1) // 2
    par = 2 * d * d + d + po * d + po * po + po  # F, Q, b, Hobs, Robs, cobs of one step
    gain = d * d + d + 2 * d * p + p + sym(p) + 1  # one GainRow (affine_shared.h): Mb, kc, K, HF, ym, Si, c0
    samp = 3 * d * d + d                                           # one SampShared row: G, M1, gb, Lc
    logrow = sym(d) + d * d + d + 1 + po * d + po + 1               # one LogShared row: WQ, WF, wb, cQ, WH, yw, cR
    if mode == "fused":
        # the chain-shared sweep in TWO streaming passes on a lazy state (csrc/fused_shared.h, auxssm_kalman_sweep_fused; round 4: passes A and C of round 3 merged):
        # the library's profiler files pass AC under filter_scan, the two aggregate scans (+ the Psi m_start completion between them) under sample_scan, pass E under
        # logpdf, the t = 0 terms + accept step under select; no noise buffers, no filtered-mean buffer, no select pass
        ntab = d * d                                                  # N_t = M1_t Phi_t: the first mean's contribution to an increment
        return {
            "factory": (0, 0, par),
            "filter_tab": (0, 0, par + gain + d * d),
            "sample_init": (0, 0, 2 * d * d + d + samp + logrow + 3 * d * d),  # sampler / log-density tables, within-chunk products (gains, filter matrices), N_t
            "filter_scan": (d, 2 * d, gain + samp + d * d + logrow),  # pass AC: x in, u and the local increments out (eps_aux, eps_samp drawn in registers; MH terms of x)
            "sample_scan": (0, 0, 0),                                 # aggregate scans: O(C nchunk) work
            "logpdf": (2 * d, d, samp + ntab + logrow),               # pass E: increments and u in, x' out (MH terms of x')
            "select": (0, 0, 0),                                      # t = 0 terms, accept, selector flip: O(C) work
        }
    if mode == "shared":
        return {
            # keyed sweep (auxssm_kalman_sweep_keyed): the noise is drawn inside the scans' reduce passes; the fill kernel only draws row t = 0 of
            # the two arrays and the acceptance uniforms (no per-step bytes)
            "factory": (0, 0, par),
            "filter_tab": (0, 0, par + gain + d * d),                 # parameters in; gain rows and filtered covariances out (ONE sequence)
            "filter_scan": (d, 2 * d, gain),                          # x in; filtered means and eps_aux (drawn here, read again by the log-density) out; + ell
            "sample_init": (0, 0, 2 * d * d + d),                     # filtered covariances -> sampler gains / factors, once per time step
            "sample_scan": (d, d, d * d + d),                         # ms in (eps_samp drawn here), x' out
            "logpdf": (3 * d, 0, par),                                # x, x', eps_aux
            "select": (d, d, 0),                                      # x' -> x (accepted chains)
        }
    return {
        "rng": (0, 2 * d, 0),
        "factory": (0, 0, par),
        "filter_tab": (0, 0, po * d + po * po + 2 * po + sym(d) + d + 4),  # observation-information rows (kalman_math.h ObsInfoRow)
        # the reference's element buffer in, filtered moments out.  This build never materialises the elements (they are
        # folded onto the prefix from x, eps_aux and the information rows), so its real traffic is BELOW this figure (profiles/r02_traffic.json).
        "filter_scan": (3 * d * d + 2 * d, d * d + d, 0),
        "sample_scan": (d * d + 2 * d, d, 2 * d * d + d),             # ms, Ps, eps_samp -> x' (elements rebuilt on the fly)
        "logpdf": (3 * d, 0, par),
        "select": (d, d, 0),
    }


# groups of the chain-shared sweep's MODEL STAGE: a few short dependent launches on ONE sequence that run on the library's second stream beside the
# previous sweep (AUXSSM_OPT_OVERLAP_MODEL_STAGE) -- off the critical path, so never the roofline kernel of the step
MODEL_STAGE_GROUPS = ("filter_tab", "factory", "sample_init")


def model_stage_overlapped(mode):
    return mode in ("shared", "fused") and os.environ.get("AUXSSM_OVERLAP_TAB", "1") != "0"


FUSED_PASS_NAMES = {"filter_scan": "pass AC: k_fs_ac (x -> u, local sampler increments; filter fold = chunk-local means; draws; MH terms of x)",
                    "sample_scan": "k_aff_aggs x 2 + k_fs_esfix (first means of the chunks, sampler chunk starts)",
                    "logpdf": "pass E: k_fs_e (increments, u -> x', MH terms of x')"}


def kalman_rooflines(groups, mode, C, T, d, po, s, steps):
    """per-group average ms per step and algorithmic GB/s; returns (kernels dict, dominant group name)"""
    reals = kalman_group_reals(mode, d, po)
    off_path = MODEL_STAGE_GROUPS if model_stage_overlapped(mode) else ()
    out, dom = {}, None
    for g, (n, ms) in groups.items():
        per_step = ms / steps
        ent = {"ms_per_step": round(per_step, 4), "launch_groups_per_step": round(n / steps, 2)}
        if g in reals:
            r, w, sh = reals[g]
            b = (C * (r + w) + sh) * T * s
            ent["algorithmic_bytes_per_step"] = int(b)
            ent["algorithmic_GBps"] = round(b / (per_step * 1e-3) / 1e9, 1) if per_step > 0 else None
        if g in off_path:
            ent["stream"] = "model stage: second stream, overlapped with the previous sweep"
        out[g] = ent
        if g not in off_path and (dom is None or per_step > out[dom]["ms_per_step"]):
            dom = g
    return out, dom


def pmc_traffic(key):
    """HBM bytes per launch group measured with the PMC counters (profiles/r02_traffic.json, written from committed rocprofv3 --pmc
    passes by tools/pmc_traffic.py); None when no pass exists for these kernels."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            ent = tj.get(key)
            if ent:
                return ent.get("hbm_bytes"), ent.get("source")
        except Exception:
            pass
    return None, None


def leg_c2(ctx, args, share, steps, warmup, chains_obj=None):
    """C2 aux-Kalman sweeps; returns dict(value, ms_per_step, kernels, roofline, accept...)"""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    _lib, handle = ctx.lib, ctx.handle
    dtype = np.float64 if args.dtype == "f64" else np.float32
    T, d, C = args.T, args.d, args.chains
    if chains_obj is None:
        m, model = build_model(T, d, dtype)
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        rng = np.random.Generator(np.random.PCG64(1000 + ctx.rank))
        x0 = (m["x_true"][None] + 0.3 * rng.standard_normal((C, T, d))).astype(dtype)
        chains_obj = (DeviceChains(handle, x0, model=model if share else None), kernel)
    chains, kernel = chains_obj
    state = KalmanSampler(x=chains, updated=None)
    handle.set_option(_lib.OPT_SHARE_MODEL, int(share))
    keys = R.split(R.PRNGKey(2024 + ctx.rank + (0 if share else 7919)), steps + warmup + 4)  # (+ the profile pass's own keys: timed())
    delta = 0.5
    mode_hint = "shared" if share and chains.chain_minor and C > 1 else "general"
    # the roofline kernel of either path is the filter's scan (the parallel-in-time scan north_star names; on the shared path it and the sampler's
    # scan are within a few per cent of each other, the model stage is off the critical path)
    def focus(ps):
        if getattr(chains, "fused", None):  # two streaming passes: the roofline kernel is the longer of them
            cand = {g: v for g, v in ps.items() if g in FUSED_PASS_NAMES}
            if cand:
                return max(cand, key=cand.get)
        return "filter_scan" if "filter_scan" in ps else max(ps, key=ps.get)
    el, groups = ctx.timed(lambda k: kernel(keys[k], state, delta), steps, warmup, prof=not args.no_prof, focus=focus)
    handle.set_option(_lib.OPT_SHARE_MODEL, 1)
    s = np.dtype(dtype).itemsize
    mode = "shared" if share and chains.chain_minor and C > 1 else "general"
    if mode == "shared" and getattr(chains, "fused", None):
        mode = "fused"
    kernels, dom = kalman_rooflines(groups, mode, C, T, d, d, s, steps)
    out = dict(value=C * ctx.world * steps / el, ms_per_step=el / steps * 1e3, kernels=kernels, mode=mode, chains_obj=chains_obj)
    roof = None
    if dom is not None:
        # the general path's roofline kernel is always the filter's associative scan (the d x d block-affine combine north_star names)
        g = "filter_scan" if "filter_scan" in kernels else dom
        if mode == "fused":
            g = max((q for q in FUSED_PASS_NAMES if q in kernels), key=lambda q: kernels[q]["ms_per_step"])
        k = kernels[g]
        if k.get("algorithmic_GBps"):
            key = f"kalman_C2_{args.dtype}_T{T}_d{d}_chains{C}_{mode}_{g}"
            traffic, src = pmc_traffic(key)
            roof = dict(bound="hbm", achieved=k["algorithmic_GBps"], peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(k["algorithmic_GBps"] / HBM_PEAK_GBPS, 4),
                        traffic=traffic, traffic_source=src, kernel=(FUSED_PASS_NAMES[g] if mode == "fused" else f"{g} ({mode} path)"), avg_launch_ms=k["ms_per_step"],
                        launches=int(groups[g][0]), algorithmic_bytes_per_launch=k["algorithmic_bytes_per_step"],
                        share_of_step=round(k["ms_per_step"] / (el / steps * 1e3), 3))
            if mode == "fused":  # the passes' VALU-issue fractions from the committed SQ counter passes of this command (tools/pmc_sq.sh, tools/sq_fused_summary.py): pass AC
                # is bound by instruction issue (fp64 Threefry + Box-Muller inside the pass), pass E by HBM -- the HBM fraction alone does not say that
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic.json")))
                    vi = {q: {kk: tj[f"fused_C2_sq_{q}"].get(kk) for kk in ("valu_issue_frac", "valu_per_chain_step", "us_per_launch", "waves_per_simd", "core_clock_GHz")}
                          for q in ("pass_AC", "pass_E") if f"fused_C2_sq_{q}" in tj}
                    if vi and args.dtype == "f64" and (T, d, C) == (65536, 4, 256):
                        roof["valu_issue"] = dict(vi, source=tj["fused_C2_sq_pass_AC"]["source"])
                except Exception:
                    pass
            if mode == "general" and g == "filter_scan":
                n = T - 1
                k3 = C * n * ((3 * d * d + 2 * d) + (d * d + d)) * s  # the reference's unpacked (A, b, C, eta, J) elements
                roof["k3_equivalent_GBps"] = round(k3 / (k["ms_per_step"] * 1e-3) / 1e9, 1)
                roof["k3_equivalent_frac"] = round(roof["k3_equivalent_GBps"] / HBM_PEAK_GBPS, 4)
    if roof is not None and mode == "fused":
        # the whole sweep against the same peak: the algorithmic bytes of all launch groups on the chains' stream (the two passes) over the step time -- beside the
        # dominant pass's own fraction, which says little while that pass is bound by instruction issue (valu_issue above)
        chain_groups = [q for q in FUSED_PASS_NAMES if q in kernels]
        tot = sum(kernels[q].get("algorithmic_bytes_per_step", 0) for q in chain_groups)
        if tot:
            gbps = tot / (el / steps) / 1e9
            roof["whole_sweep"] = dict(algorithmic_bytes_per_step=int(tot), achieved=round(gbps, 1), frac=round(gbps / HBM_PEAK_GBPS, 4), unit="GB/s",
                                       note="the chain passes' algorithmic bytes (AC + E: six array passes) / wall time of a step")
    out["roofline"] = roof
    return out


def leg_c3_csmc(ctx, T, N, Cn, steps, warmup, dtype=np.float32, cpu=False):
    """BASELINE configs[2] = C3: SV d=1, auxiliary cSMC with independent proposals, backward sampling, in-kernel Threefry noise."""
    import ctypes as C
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics, SVPotential
    _lib, handle = ctx.lib, ctx.handle
    phi, q, xtrue, y = sv_data(T)
    M0 = GaussianInit(m0=[0.0], P0=[[q]])
    Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
    fk = _device.describe_independent(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), Mt)
    rng = np.random.Generator(np.random.PCG64(1000 + ctx.rank))
    x0 = (xtrue[None] + 0.1 * rng.standard_normal((Cn, T, 1))).astype(dtype)
    xd = handle.to_device(x0)
    anc = handle.zeros((Cn, T), np.int32)
    yd = fk.ydev(handle, dtype)
    shd = handle.to_device(np.full(T, np.sqrt(0.25)), dtype)
    m = fk.struct(handle, dtype, T)
    keys = R.split(R.PRNGKey(77 + ctx.rank), steps + warmup + 4)

    def step(k):
        nz = _lib.CsmcNoise()
        nz.mode, nz.key0, nz.key1 = _lib.NOISE_THREEFRY, int(keys[k][0]), int(keys[k][1])
        _lib.check(handle.lib.auxssm_csmc_sweep(handle.h, _lib.dtype_code(dtype), C.byref(m), Cn, T, N, 1, shd.ptr, xd.ptr,
                                                C.byref(nz), anc.ptr, None, None, None))

    el, groups = ctx.timed(step, steps, warmup)
    moved = float((anc.to_host() != 0).mean())
    s = np.dtype(dtype).itemsize
    out = dict(workload=f"C3: stochastic volatility d=1 T={T}, auxiliary cSMC N={N}, independent proposals, backward sampling",
               chains_per_gpu=Cn, steps=steps, value=round(Cn * ctx.world * steps / el, 2), unit="sweeps/s", ms_per_step=round(el / steps * 1e3, 3),
               dtype="f32" if s == 4 else "f64", updated_fraction=moved)
    kern = {}
    alg = {"csmc_fwd": Cn * T * N * (1 * s + s),   # forward pass writes xs + log_ws (As is not stored with backward sampling)
           "csmc_bwd": Cn * T * N * (1 * s + s)}   # backward sampling re-reads them
    for g, (n, ms) in groups.items():
        kern[g] = {"ms_per_step": round(ms / steps, 3)}
        if g in alg:
            kern[g]["algorithmic_GBps"] = round(alg[g] / (ms / steps * 1e-3) / 1e9, 1)
    out["kernels"] = kern
    if "csmc_fwd" in groups:
        g = max(("csmc_fwd", "csmc_bwd"), key=lambda q_: kern.get(q_, {}).get("ms_per_step", 0))
        ach = kern[g]["algorithmic_GBps"]
        traffic, src = pmc_traffic(f"csmc_C3_{'f32' if dtype == np.float32 else 'f64'}_T{T}_N{N}_chains{Cn}_{g}")
        out["roofline"] = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4), traffic=traffic,
                               traffic_source=src, kernel=f"k_{g} (persistent, one workgroup per chain)", avg_launch_ms=kern[g]["ms_per_step"],
                               algorithmic_bytes_per_launch=alg[g])
        # these kernels are VALU-issue / latency bound, not HBM bound: the committed SQ counter passes (tools/pmc_sq_c3.sh -> tools/sq_summary.py) say how busy
        # the vector ALUs are -- SQ_INSTS_VALU x 4 cycles over SQ_BUSY_CU_CYCLES x 4 SIMDs -- and how many instructions a wave issues per time step
        sq = {}
        for kk in ("csmc_fwd", "csmc_bwd"):
            ent = None
            for tj in ("r04_traffic.json", "r03_traffic.json"):   # (round 4: profiles/r04_j_sq_counters_c3.txt, the same kernels re-measured)
                try:
                    ent = json.load(open(os.path.join(ROOT, "profiles", tj))).get(f"csmc_C3_sq_k_{kk}")
                except Exception:
                    ent = None
                if ent:
                    break
            if ent:
                sq[kk] = {q: ent.get(q) for q in ("valu_issue_frac", "valu_per_wave_step", "salu_per_wave_step", "lds_per_wave_step", "branch_per_wave_step", "source")}
        if sq:
            out["valu_issue"] = sq
    if cpu:
        from oracle import csmc as O
        Tb = min(T, 4096)
        r2 = np.random.default_rng(0)
        od = dict(proposal=O.AUX_INDEPENDENT, potential=O.POT_SV, m0=[0.0], chol_P0=[[np.sqrt(q)]], F=[[phi]], b=[0.0], chol_Q=[[np.sqrt(q)]])
        kw = dict(y=y[:Tb], sqrt_half_delta=np.full(Tb, 0.5), eps_aux=r2.standard_normal((Tb, 1)), eps_prop=r2.standard_normal((Tb, N, 1)),
                  u_res=r2.random((Tb - 1, N)), u_bwd=r2.random(Tb))
        t1 = time.perf_counter()
        O.sweep(od, xtrue[:Tb], N, True, dtype=np.float32, **kw)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = dict(value=round(1.0 / (dt * T / Tb), 4), unit="sweeps/s", cores=1, kind="port",
                                   sample=f"1 sweep of 1 chain at T={Tb} (scaled linearly to T={T}), N={N}, oracle/csmc_ref.c, {dt:.1f} s")
    return out


def leg_c4(ctx, total_chains=64, T=16384, N=512, steps=20, warmup=3):
    """BASELINE configs[3] = C4: Lorenz-63 T=16384, aux-Kalman sweep (extended linearisation on device) and cSMC N=512, a FIXED total of
    `total_chains` chains sharded over the ranks (parallel.shard_chains: 64 -> 8 per GPU on 8 GPUs): strong scaling."""
    from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup, lorenz_setup
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.parallel import shard_chains, chain_key
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.csmc import CsmcChains, CSMCState
    from aux_ssm_samplers_amd._primitives.csmc import get_kernel as get_csmc_kernel
    handle = ctx.handle
    lo, hi = shard_chains(total_chains, ctx.rank, ctx.world)
    Cn = hi - lo
    out = dict(workload=f"C4: Lorenz-63 T={T} dt=1.25e-4, (x2, x3) observed every 80 steps, fp32", total_chains=total_chains,
               chains_this_rank=Cn, scaling="strong", parallelism=f"{total_chains} chains block-partitioned over {ctx.world} rank(s), no collective")
    model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    M0, Mt, G0, Gt, xt, y, sig_y = lorenz_setup(T, every=80, dt=1.25e-4)
    cinit, ck = get_csmc_kernel(M0, G0, Mt, Gt, N, backward=True, Pt=Mt)
    s, d, po = 4, 3, 2

    def measure(Cn, lo, total, ksteps, kwarm, csteps):
        """both samplers on Cn resident chains; `total` chains are what the rate is scaled to (all ranks run the same number of chains)"""
        ch = DeviceChains(handle, np.repeat(xtrue[None], Cn, axis=0).astype(np.float32))
        st = KalmanSampler(x=ch, updated=None)
        keys = R.split(chain_key(R.PRNGKey(4), lo), ksteps + kwarm + 4)  # the rank's stream is folded from its first global chain id
        el, groups = ctx.timed(lambda k: kernel(keys[k], st, 1e-4), ksteps, kwarm)
        acc = ctx.sum_over_ranks(float(ch.accepted.to_host().sum())) / total
        # fused-sweep lower bound of SURVEY 8(d): x, eps_aux, eps_samp read, x' written per chain-step (the linearised F_t, b_t are functions of x)
        alg = Cn * T * 4 * d * s
        kal = dict(value=round(total * ksteps / el, 1), unit="sweeps/s", ms_per_step=round(el / ksteps * 1e3, 4), steps=ksteps,
                   accept_rate=round(acc, 3), layout="chain-minor" if ch.chain_minor else "dense",
                   kernels={g: round(ms / ksteps, 4) for g, (n, ms) in groups.items()},
                   roofline=dict(bound="hbm", achieved=round(alg / (el / ksteps) / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                 frac=round(alg / (el / ksteps) / 1e9 / HBM_PEAK_GBPS, 4), traffic=None,
                                 kernel="whole sweep (about 40 dependent launches; latency-bound at 8 chains per GPU)",
                                 algorithmic_bytes_per_launch=alg))
        del ch, st
        cc = CsmcChains(handle, np.repeat(xt[None], Cn, axis=0).astype(np.float32))
        cst = CSMCState(x=cc, updated=None)
        ckeys = R.split(chain_key(R.PRNGKey(5), lo), 12)
        el, groups = ctx.timed(lambda k: ck(ckeys[k], cst), csteps, 1)
        alg = Cn * T * N * (d * s + s)
        fwd = groups.get("csmc_fwd", (0, 0.0))[1] / csteps
        csm = dict(value=round(total * csteps / el, 2), unit="sweeps/s", ms_per_step=round(el / csteps * 1e3, 3), steps=csteps, particles=N,
                   updated_fraction=float((cc.ancestors.to_host() != 0).mean()),
                   kernels={g: round(ms / csteps, 3) for g, (n, ms) in groups.items()},
                   roofline=None if not fwd else dict(bound="hbm", achieved=round(alg / (fwd * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                                      frac=round(alg / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), traffic=None,
                                                      kernel="k_csmc_fwd<float,3>", avg_launch_ms=round(fwd, 3), algorithmic_bytes_per_launch=alg))
        return kal, csm

    out["kalman"], out["csmc"] = measure(Cn, lo, total_chains, steps, warmup, 3)
    if ctx.world == 1 and total_chains >= 16:
        # What the 8-GPU run of this leg will look like, from THIS GPU: the shard of one rank (total / 8 chains) measured alone.  Eight ranks run eight such
        # shards side by side with no exchange, so the 8-GPU rate is 8 x the shard's rate -- and because one shard leaves most of the chip idle (one
        # workgroup per chain in the cSMC kernels, short dependent launches in the Kalman sweep), that is NOT 8 x the single-GPU figure above.
        k8, c8 = measure(total_chains // 8, 0, total_chains // 8, max(5, steps // 2), 2, 2)
        out["prediction_8_gpus"] = dict(
            shard_chains=total_chains // 8,
            kalman=dict(shard_sweeps_per_s=k8["value"], predicted_sweeps_per_s=round(8 * k8["value"], 1), vs_one_gpu=round(8 * k8["value"] / out["kalman"]["value"], 2)),
            csmc=dict(shard_sweeps_per_s=c8["value"], predicted_sweeps_per_s=round(8 * c8["value"], 2), vs_one_gpu=round(8 * c8["value"] / out["csmc"]["value"], 2)),
            note="strong scaling of a FIXED 64 chains: the leg that carries north_star's >= 6x at 8 GPUs is the C2 headline (weak scaling, 256 chains per GPU, "
                 "no exchange between ranks), not this one")
    return out


def leg_sv_kalman(ctx, T=65536, chains=1024, steps=5, warmup=2):
    """The general per-chain path on a real nonlinear model: the SV model of C3 with the auxiliary KALMAN sampler, second-order
    observations (R_t depends on the chain's state: examples/stochastic_volatility/auxiliary_kalman.py:37-46), fp64, chain-minor.  Every chain folds its own steps in
    information form straight from (x, u, y) -- no observation arrays, no elements (csrc/kalman_bodies.h::FilterOpFlySV).  Fixed step size 0.0567 (what the reference's
    adaptation rule settles at for this model, tools/bench_configs.py c3k)."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.workloads import sv_setup
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, 1, rho=0.0)
    out = dict(workload=f"SV d=1 T={T} (the C3 model), auxiliary Kalman sampler with second-order observations, fp64, general per-chain path", chains_per_gpu=chains)
    for order in (2, 1):
        model = SVModel(y, m0, P0, F, Q, b, order=order)
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        ch = DeviceChains(ctx.handle, np.repeat(xtrue[None], chains, axis=0))
        st = KalmanSampler(x=ch, updated=None)
        keys = R.split(R.PRNGKey(11 + ctx.rank), steps + warmup + 4)
        delta = 0.0567 if order == 2 else 0.0212
        el, groups = ctx.timed(lambda k: kernel(keys[k], st, delta), steps, warmup)
        # the sweep's own lower bound (fused form): x, eps_aux, eps_samp read, x' written per chain-step
        alg = chains * T * 4 * 8
        ent = dict(value=round(chains * ctx.world * steps / el, 1), unit="sweeps/s", ms_per_step=round(el / steps * 1e3, 4), steps=steps, delta=delta,
                   accept_rate=round(float(ch.accepted.to_host().mean()), 3), kernels={g: round(ms / steps, 4) for g, (n, ms) in groups.items()},
                   roofline=dict(bound="hbm", achieved=round(alg / (el / steps) / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(alg / (el / steps) / 1e9 / HBM_PEAK_GBPS, 4),
                                 traffic=None, kernel="whole sweep against the fused lower bound (x, eps_aux, eps_samp read, x' written)", algorithmic_bytes_per_launch=alg))
        out["order_2" if order == 2 else "order_1_chain_shared"] = ent
        del ch, st
    return out


def leg_c5(ctx, T=8192, d=64, seqs=(1, 16), steps=3):
    """BASELINE configs[4] = C5: dense d = p = 64, T = 8192, fp32 -- the wide-state filter (MFMA d x d combine), one sequence (latency) and
    16 sequences per launch (throughput), resident in HBM.  Roofline = MFMA: SURVEY 8(d)'s K3 flops 2 n 19.3 d^3 per scan / scan time."""
    import ctypes as C
    from aux_ssm_samplers_amd.workloads import c5_model
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    _lib, handle = ctx.lib, ctx.handle
    u, lg64, x = c5_model(T, d)
    f32 = np.float32
    out = dict(workload=f"C5: dense LG-SSM d=p={d} T={T} fp32, first-order auxiliary observations, wide-state filter (parallel scan)", runs=[])
    for S in seqs:
        dl = DeviceLGSSM(handle, tuple(lg64), 1, T, 1, d, d, False, f32)  # chain-shared parameters (stride 0), S sequences of observations
        ys = (u[None] + np.concatenate([np.zeros((1, T, d)), 0.3 * np.random.default_rng(4).standard_normal((S - 1, T, d))])).astype(f32)  # S different sequences
        yd = handle.to_device(ys)
        yarr = yd.arr(T * d, d, 0)
        ms = handle.empty((S, T, 1, d), f32)
        Ps = handle.empty((S, T, 1, d, d), f32)
        ell = handle.empty((S,), f32)
        dims = _lib.Dims(S, T, 1, d, d)

        def step(k):
            _lib.check(handle.lib.auxssm_kalman_filter(handle.h, _lib.F32, C.byref(dims), C.byref(dl.c), C.byref(yarr), 1, ms.ptr, Ps.ptr, ell.ptr))

        el, groups = ctx.timed(step, steps, 1)
        scan = groups.get("filter_scan", (0, 0.0))[1] / steps
        shared = "filter_tab" in groups  # chain-shared parameters: ONE matrix recursion (sequence 0), the other sequences ride as columns of the mean scan
        flops = (1 if shared else S) * 2 * (T - 1) * 19.3 * d ** 3
        ent = dict(sequences_per_launch=S, path=("shared: one matrix filter + gain table (filter_tab) + mean / log-likelihood scan of all sequences (filter_ell)"
                                                 if shared else "per sequence"), filters_per_s=round(S * ctx.world * steps / el, 2), ms_per_filter_call=round(el / steps * 1e3, 3),
                   kernels={g: round(t / steps, 3) for g, (n, t) in groups.items()})
        if scan:
            tf = flops / (scan * 1e-3) / 1e12
            ent["roofline"] = dict(bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                   traffic=None, kernel="wide filter scan (wk_fold_reduce + tree of wk_scan_reduce / wk_scan_aggs / wk_scan_down_pre + wk_fold_down)", avg_launch_ms=round(scan, 3),
                                   algorithmic_flops_per_launch=flops)
            try:  # matrix-core busy fractions of the wide kernels from the committed PMC pass of this command (round 4: profiles/r04_e_pmc_mfma_busy_c5.csv -> r04_traffic.json)
                mb = None
                for tj in ("r04_traffic.json", "r02_traffic.json"):
                    try:
                        mb = json.load(open(os.path.join(ROOT, "profiles", tj))).get(f"wide_C5_f32_d{d}_T{T}")
                    except Exception:
                        mb = None
                    if mb:
                        break
                if mb:
                    ent["roofline"]["mfma_busy_pmc"] = mb["mfma_busy"]
                    ent["roofline"]["mfma_busy_source"] = mb["source"]
            except Exception:
                pass
        out["runs"].append(ent)
        del dl, yd, ms, Ps, ell
    # "benchmark both" -- the same grid as the reference's spatial example actually runs it (examples/spatial/model.py:103-112, auxiliary_kalman.py:18-28):
    # B = 64 INDEPENDENT scalar LGSSMs on the batch axis, dx = dy = 1, through the per-lane register kernels.  HBM-bound: K3 bytes of SURVEY 8(d) with d = 1
    # (read 3 d^2 + 2 d, write d^2 + d reals per element and sequence) against the filter scan's launch-group time.
    out["batched_scalar"] = []
    try:
        from aux_ssm_samplers_amd.workloads import c5_batched_model
        for Tb, Cb in ((1024, 64), (T, 16)):
            ub, lgb, xb = c5_batched_model(Tb)
            B = ub.shape[1]
            dlb = DeviceLGSSM(handle, tuple(lgb), 1, Tb, B, 1, 1, True, f32)    # parameters shared by the Cb sequences of observations (chain stride 0), batch axis B
            ysb = (ub[None] + np.concatenate([np.zeros((1, Tb, B, 1)), 0.3 * np.random.default_rng(5).standard_normal((Cb - 1, Tb, B, 1))])).astype(f32)
            ydb = handle.to_device(ysb)
            yarrb = ydb.arr(Tb * B, B, 1)
            msb, Psb, ellb = handle.empty((Cb, Tb, B, 1), f32), handle.empty((Cb, Tb, B, 1, 1), f32), handle.empty((Cb,), f32)
            dimsb = _lib.Dims(Cb, Tb, B, 1, 1)

            def stepb(k):
                _lib.check(handle.lib.auxssm_kalman_filter(handle.h, _lib.F32, C.byref(dimsb), C.byref(dlb.c), C.byref(yarrb), 1, msb.ptr, Psb.ptr, ellb.ptr))

            el, groups = ctx.timed(stepb, max(steps, 5), 1)
            nst = max(steps, 5)
            scan = groups.get("filter_scan", (0, 0.0))[1] / nst
            alg = Cb * B * (Tb - 1) * (5 + 2) * 4                                # K3 at d = 1: (3 + 2) reals read, (1 + 1) written, fp32
            ent = dict(sequences=Cb, batch=B, T=Tb, scalar_filters_per_s=round(Cb * B * ctx.world * nst / el, 1), ms_per_call=round(el / nst * 1e3, 4),
                       kernels={g: round(t / nst, 4) for g, (n, t) in groups.items()})
            if scan:
                ent["roofline"] = dict(bound="hbm", achieved=round(alg / (scan * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                                       frac=round(alg / (scan * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), traffic=None, kernel="filter scan, d = 1, one sequence per lane (k_scan_reduce_cm<FilterOpBuildCm> / k_scan_aggs / k_scan_down_cm<FilterOpSeqWalk>: no element buffer)",
                                       avg_launch_ms=round(scan, 4), algorithmic_bytes_per_launch=alg)
            out["batched_scalar"].append(ent)
            del dlb, ydb, msb, Psb, ellb
    except Exception as e:
        out["batched_scalar"] = {"error": f"{type(e).__name__}: {e}"}
    # a whole auxiliary Kalman SWEEP of 16 chains on one wide model ("the C5 16-sequence full pass >= 1 000 passes/s"): d = 60 states + 4 real
    # observations = 64 concatenated observations (the blocked eliminations hold n <= 64), chain-shared matrix filter, ONE covariance copy, the sampler's gain / factor
    # tables and the log-densities' inverse tables once per time step with the chains as columns (wide.hip::run_sample_shared, wide_shared.h::wk_lp_cols)
    try:
        from aux_ssm_samplers_amd import random as R
        from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel
        from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
        ds, pos, Cs = (60, 4, 16) if d == 64 else (max(d - 4, 6), 4, 4)
        rng = np.random.default_rng(0)
        Fw = 0.9 * np.eye(ds) + 0.04 * (np.eye(ds, k=1) + np.eye(ds, k=-1))
        Hw = rng.standard_normal((pos, ds)) / np.sqrt(ds)
        xw = np.zeros((T, ds))
        xw[0] = rng.standard_normal(ds)
        for t in range(1, T):
            xw[t] = Fw @ xw[t - 1] + np.sqrt(0.2) * rng.standard_normal(ds)
        yw = xw @ Hw.T + np.sqrt(0.5) * rng.standard_normal((T, pos))
        bt = np.broadcast_to
        wmodel = LGConcatModel(np.zeros(ds), np.eye(ds), bt(Fw, (T - 1, ds, ds)), bt(0.2 * np.eye(ds), (T - 1, ds, ds)), bt(np.zeros(ds), (T - 1, ds)),
                               bt(Hw, (T, pos, ds)), bt(0.5 * np.eye(pos), (T, pos, pos)), bt(np.zeros(pos), (T, pos)), yw)
        winit, wkernel = get_kernel(wmodel.dynamics_factory, wmodel.observations_factory, wmodel.log_likelihood_fn, True)
        wch = DeviceChains(handle, (xw[None] + 0.3 * rng.standard_normal((Cs, T, ds))).astype(f32))
        wst = KalmanSampler(x=wch, updated=None)
        wkeys = R.split(R.PRNGKey(21 + ctx.rank), 16)
        nst = max(steps, 3)
        el, groups = ctx.timed(lambda k: wkernel(wkeys[k], wst, 0.3), nst, 1)
        out["sweep_16_chains"] = dict(workload=f"auxiliary Kalman sweep, dense LG-SSM d={ds} + {pos} observations, T={T}, {Cs} chains on one model, fp32", chains=Cs,
                                      chain_sweeps_per_s=round(Cs * ctx.world * nst / el, 1), ms_per_sweep_call=round(el / nst * 1e3, 3),
                                      accept_rate=float(wch.accepted.to_host().mean()), kernels={g: round(ms / nst, 3) for g, (n, ms) in groups.items()})
        del wch, wst
    except Exception as e:
        out["sweep_16_chains"] = {"error": f"{type(e).__name__}: {e}"}
    # one whole pass of the path on one chain (filter -> pathwise sampler -> joint log-density, what a sweep of this size strings together):
    # device time of each launch group (HIP events), host <-> device copies of the NumPy front end excluded
    try:
        import aux_ssm_samplers_amd._primitives.kalman as P
        lg = P.LGSSM(*[np.ascontiguousarray(a, f32) for a in lg64])
        uu, eps = u.astype(f32), np.random.default_rng(0).standard_normal((T, d)).astype(f32)
        tm = {}
        for rep in range(2):
            for kid, name in ((_lib.K_FILTER_INIT, "filter_init"), (_lib.K_FILTER_SCAN, "filter_scan")):
                handle.prof_enable(kid, 8)
                fm, fP, fell = P.filtering(uu, lg, True)
                tm[name] = handle.prof_read()[1]
                handle.prof_disable()
            for kid, name in ((_lib.K_SAMPLE_INIT, "sample_init"), (_lib.K_SAMPLE_SCAN, "sample_scan")):
                handle.prof_enable(kid, 8)
                xs = P.sampling(None, fm, fP, lg, True, eps=eps)
                tm[name] = handle.prof_read()[1]
                handle.prof_disable()
            handle.prof_enable(_lib.K_LOGPDF, 8)
            P.posterior_logpdf(uu, xs, fell, lg)
            tm["logpdf"] = handle.prof_read()[1]
            handle.prof_disable()
        tot = sum(tm.values())
        out["one_chain_pass"] = dict(kernels_ms={k: round(v, 3) for k, v in tm.items()}, total_ms=round(tot, 3), passes_per_s=round(1e3 / tot, 1),
                                     note="filter + pathwise sampler + joint log-density, device time of the launch groups (second repetition)")
    except Exception as e:
        out["one_chain_pass"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cpu_baseline_c2(T, d, budget_s=12.0):
    """oracle/kalman_seq.c (the reference's sequential sweep, its CPU code path) on this box's host cores: all cores and one thread."""
    from oracle import kalman_seq as S
    from aux_ssm_samplers_amd.workloads import lg_model
    m = lg_model(T, d)
    cm = S.Model(m["m0"], m["P0"], m["F"], m["Q"], m["b"], m["Hobs"], m["Robs"], m["cobs"], m["y"])
    ncore = os.cpu_count() or 1
    nthr = min(S.max_threads(), ncore)
    rng = np.random.default_rng(0)

    def run(Cn, nt):
        x = m["x_true"][None] + 0.3 * rng.standard_normal((Cn, T, d))
        ea, es, ua = rng.standard_normal((Cn, T, d)), rng.standard_normal((Cn, T, d)), rng.random(Cn)
        S.sweep(cm, x[:1], 0.5, ea[:1], es[:1], ua[:1], nthreads=1)  # warm the pages
        reps, t0 = 0, time.perf_counter()
        while True:
            S.sweep(cm, x, 0.5, ea, es, ua, nthreads=nt)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or reps >= 5:
                return Cn * reps / el, reps, el

    v1, r1, e1 = run(1, 1)
    vn, rn, en = run(2 * nthr, nthr)
    return dict(value=round(vn, 3), unit="sweeps/s", cores=nthr, kind="port", single_thread_value=round(v1, 3), host_cpus=ncore,
                sample=(f"oracle/kalman_seq.c (C restatement of the reference's sequential filter / sampler sweep, its CPU path; not JAX): {rn} x {2 * nthr} chains "
                        f"on {nthr} OpenMP threads in {en:.1f} s; 1 thread: {r1} sweep(s) in {e1:.1f} s; same T={T} d={d} fp64 workload"))


# ------------------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    res = resolve_launch(args, argv, os.environ)
    if res[0] == "error":
        print("bench.py: " + res[1], file=sys.stderr)
        return 2
    if res[0] == "spawn":
        # nothing in this process has touched the GPU (torch is not even imported): start the ranks as children and pass their code on
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        return subprocess.call(res[1], env=env)
    _, rank, world, local_rank = res
    ctx = Ctx(args, rank, world, local_rank)
    assert ctx.world == args.gpus
    sec_on = [] if args.no_secondary else [s for s in args.secondary.split(",") if s]

    if args.workload == "csmc":
        r = leg_c3_csmc(ctx, args.T, args.N, args.chains, args.steps, args.warmup, np.float32 if args.dtype == "f32" else np.float64,
                        cpu=not args.no_cpu_baseline and world == 1 and rank == 0)
        if rank == 0:
            print(json.dumps({
                "metric": "Gibbs sweeps/sec (auxiliary cSMC, backward sampling)", "value": r["value"], "unit": "sweeps/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": r["dtype"], "data": "synthetic",
                "config": {"workload": r["workload"], "chains_per_gpu": args.chains, "delta": 0.5, "parallelism": f"chains x{world} (independent, no collective)"},
                "updated_fraction": r["updated_fraction"], "kernels": r["kernels"], "roofline": r.get("roofline"), "cpu_baseline": r.get("cpu_baseline")}))
        ctx.close()
        return 0

    share = not args.no_share_model
    main_leg = leg_c2(ctx, args, share, args.steps, args.warmup)
    chains, kernel = main_leg["chains_obj"]

    general = None
    if share and not args.no_general_leg and not args.no_prof and main_leg["mode"] in ("shared", "fused"):
        # (below 32 chains the headline's chains are chain-minor only because the model is chain-shared: the general path keeps its own, time-minor, layout there)
        general = leg_c2(ctx, args, False, max(3, args.steps // 2), 1, chains_obj=main_leg["chains_obj"] if args.chains >= 32 else None)

    # the trivial chain-gather (RCCL): acceptance flags + last log-alphas of every chain to rank 0
    from aux_ssm_samplers_amd.parallel import gather_chains
    acc = chains.accepted.to_host()
    logs = chains.logs.to_host()
    C = args.chains
    if ctx.dist is not None:
        per_chain = np.concatenate([acc[:, None].astype(np.float64), logs.astype(np.float64)], axis=1)  # (C, 6)
        g = gather_chains(per_chain, C * world, ctx.dist, dst=0, device=ctx.coll_dev)
        if rank == 0:
            acc, logs = g[:, 0], g[:, 1:]
    del chains, kernel
    main_leg.pop("chains_obj")
    if general is not None:
        general.pop("chains_obj")

    secondary = {}
    for name in sec_on:
        try:
            small = args.small_secondary
            if name == "c3":
                secondary["C3_csmc"] = leg_c3_csmc(ctx, 2048 if small else 65536, 128 if small else 1024, 8 if small else 256, 2, 1,
                                                   cpu=not args.no_cpu_baseline and world == 1 and rank == 0)
            elif name == "c4":
                secondary["C4_lorenz"] = leg_c4(ctx, 8, 1040, 64, 3, 1) if small else leg_c4(ctx)
            elif name == "c5":
                secondary["C5_wide"] = leg_c5(ctx, 96, 64, (1, 2), 2) if small else leg_c5(ctx)
            elif name == "sv":
                secondary["SV_kalman_general_path"] = leg_sv_kalman(ctx, 2048, 64, 2, 1) if small else leg_sv_kalman(ctx)
        except Exception as e:  # a secondary leg never takes the headline down; the failure is reported in the line
            secondary[name] = {"error": f"{type(e).__name__}: {e}"}
            ctx.barrier()

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_c2(args.T, args.d)
        T, d = args.T, args.d
        out = {
            "metric": "Gibbs sweeps/sec (aux-Kalman, parallel-in-time scan)", "value": round(main_leg["value"], 2),
            "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(main_leg["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C2: linear-Gaussian SSM T={T} d={d} p={2 * d}, aux-Kalman sweep, parallel scan",
                       "chains_per_gpu": C, "delta": 0.5, "parallelism": f"chains x{world} (independent, no collective)",
                       "model_sharing": ("chain-shared model parameters hoisted out of the chain loop (jax.vmap semantics)" + ("; the chain passes fused into three "
                                         "streaming passes on a lazy ping-pong state (auxssm_kalman_sweep_fused)" if main_leg["mode"] == "fused" else "")
                                         if main_leg["mode"] in ("shared", "fused") else "off: general per-chain path"),
                       "model_stage": ("off: one stream" if os.environ.get("AUXSSM_OVERLAP_TAB", "1") == "0" or main_leg["mode"] not in ("shared", "fused") else
                                       "the chain-independent stage of a sweep (matrix filter, gain / sampler / log-density tables: the filter_tab, factory and "
                                       "sample_init groups below) runs on a second stream and overlaps the chain passes of the sweep before -- the groups' "
                                       "times therefore add up to more than ms_per_step")},
            "accept_rate": float(np.mean(acc)), "max_abs_log_alpha": float(np.max(np.abs(logs[:, 0]))),
            "roofline": main_leg["roofline"], "kernels": main_leg["kernels"], "cpu_baseline": cpu,
        }
        if general is not None:
            out["general_path"] = {"value": round(general["value"], 2), "unit": "sweeps/s", "ms_per_step": round(general["ms_per_step"], 4),
                                   "roofline": general["roofline"], "kernels": general["kernels"],
                                   "note": "same workload with AUXSSM_OPT_SHARE_MODEL = 0: every chain builds and combines its own d x d elements"}
        if secondary:
            out["secondary"] = secondary
        print(json.dumps(out))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
