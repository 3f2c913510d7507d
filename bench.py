#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/sec of the auxiliary-Kalman hot path on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE configs[1] = SURVEY 8(d) C2: linear-Gaussian SSM, T = 65536, d = 4 (p = 8 with
the auxiliary observations concatenated), parallel-in-time aux-Kalman sweep, fp64, `--chains` independent chains per
GPU.  One "step" = one sweep (kalman/generic.py:53-76) of every chain resident on this GPU: device Threefry draws,
proposal LGSSM, filter scan, pathwise-sampler scan, log-densities, MH accept.  Inputs are resident in HBM before
the timed region.  value = chains * n_gpus * steps / seconds (independent chains: weak scaling, no data-path
collective; torch.distributed (RCCL) is used only for the barrier / max-over-ranks / the final chain-gather).

Also reports, on the same JSON line:
  roofline      -- the filter's associative scan (dominant kernel group: k_scan_reduce + k_scan_aggs + k_scan_down of
                   FilterOp), algorithmic bytes n(3d^2+2d)s read + n(d^2+d)s written per chain divided by
                   its HIP-event duration measured inside the timed region on the library's stream.
  cpu_baseline  -- the NumPy oracle (a port of the reference's parallel path) timed on this box's host, rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_model(T, d, dtype):
    from tests.helpers import lg_model
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    m = lg_model(T, d, dtype=dtype)
    bt = np.broadcast_to
    # time-varying arrays are materialised in full (T-1, d, d) as the reference's factories do (jnp.tile,
    # examples/stochastic_volatility/auxiliary_kalman.py:22-26); they are shared by all chains (vmap closure constants)
    full = lambda a, n: np.ascontiguousarray(bt(a, (n,) + a.shape))
    model = LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1),
                          full(m["Hobs"], T), full(m["Robs"], T), full(m["cobs"], T), m["y"])
    return m, model


def cpu_baseline(T, d, budget_s=20.0):
    """Oracle sweep (NumPy, 1 thread) on one chain; bounded sample."""
    from oracle import kalman_np as K
    m, model = build_model(T, d, np.float64)
    lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    rng = np.random.default_rng(0)
    x = m["x_true"] + 0.3 * rng.standard_normal((T, d))
    target = lambda z: K.log_likelihood(m["y"], z, lgo) + K.prior_logpdf(z, lgo)
    n, t0 = 0, time.perf_counter()
    while True:
        noise = dict(eps_aux=rng.standard_normal((T, d)), eps_samp=rng.standard_normal((T, d)), u_accept=rng.random())
        out = K.kalman_sweep(x, 0.5, model.dynamics_factory, model.observations_factory, target, True, **noise)
        x = out["x"]
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 8:
            break
    return dict(value=n / el, unit="sweeps/s", cores=1, kind="port",
                sample=f"{n} sweep(s) of 1 chain, same T={T} d={d} fp64 workload, NumPy oracle parallel path, {el:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=64, help="chains per GPU")
    ap.add_argument("--T", type=int, default=65536)
    ap.add_argument("--d", type=int, default=4)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket the scan with HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler

    dtype = np.float64 if args.dtype == "f64" else np.float32
    T, d, C = args.T, args.d, args.chains
    handle = _lib.default_handle(local_rank)
    m, model = build_model(T, d, dtype)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.Generator(np.random.PCG64(1000 + rank))
    x0 = (m["x_true"][None] + 0.3 * rng.standard_normal((C, T, d))).astype(dtype)
    chains = DeviceChains(handle, x0)
    state = KalmanSampler(x=chains, updated=None)
    key = R.PRNGKey(2024 + rank)
    delta = 0.5

    def barrier():
        handle.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        handle.sync()
        torch.cuda.synchronize()

    keys = R.split(key, args.steps + args.warmup + 1)

    def step(k):
        kernel(keys[k], state, delta)

    for k in range(args.warmup):
        step(k)
    barrier()
    if not args.no_prof:
        handle.prof_enable(_lib.K_FILTER_SCAN, args.steps + 1)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    scan_n, scan_ms = (0, 0.0)
    if not args.no_prof:
        scan_n, scan_ms = handle.prof_read()
        handle.prof_disable()

    # the trivial chain-gather (RCCL): acceptance flags + last log-alphas of every chain to rank 0
    from aux_ssm_samplers_amd.parallel import gather_chains
    acc = chains.accepted.to_host()
    logs = chains.logs.to_host()
    if dist is not None:
        per_chain = np.concatenate([acc[:, None].astype(np.float64), logs.astype(np.float64)], axis=1)  # (C, 6)
        g = gather_chains(per_chain, C * world, dist, dst=0, device=torch.device("cuda", local_rank))
        if rank == 0:
            acc, logs = g[:, 0], g[:, 1:]

    if rank == 0:
        s = np.dtype(dtype).itemsize
        n = T - 1
        alg_bytes = C * (n * (3 * d * d + 2 * d) * s + n * (d * d + d) * s)
        roof = None
        if scan_n:
            avg_s = scan_ms / scan_n * 1e-3
            ach = alg_bytes / avg_s / 1e9
            roof = dict(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4),
                        traffic=None, kernel="filter associative scan (k_scan_reduce+k_scan_aggs+k_scan_down<FilterOp>)",
                        avg_launch_ms=round(scan_ms / scan_n, 4), launches=scan_n, algorithmic_bytes_per_launch=alg_bytes)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(T, d)
        out = {
            "metric": "Gibbs sweeps/sec (aux-Kalman, parallel-in-time scan)", "value": round(C * world * args.steps / el, 2),
            "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C2: linear-Gaussian SSM T={T} d={d} p={2 * d}, aux-Kalman sweep, parallel scan",
                       "chains_per_gpu": C, "delta": delta, "parallelism": f"chains x{world} (independent, no collective)"},
            "accept_rate": float(np.mean(acc)), "max_abs_log_alpha": float(np.max(np.abs(logs[:, 0]))),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
