#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/sec of the auxiliary-Kalman hot path on MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE configs[1] = SURVEY 8(d) C2: linear-Gaussian SSM, T = 65536, d = 4 (p = 8 with
the auxiliary observations concatenated), parallel-in-time aux-Kalman sweep, fp64, `--chains` independent chains per
GPU.  One "step" = one sweep (kalman/generic.py:53-76) of every chain resident on this GPU: device Threefry draws,
proposal LGSSM, filter scan, pathwise-sampler scan, log-densities, MH accept.  Inputs are resident in HBM before
the timed region.  value = chains * n_gpus * steps / seconds (independent chains: weak scaling, no data-path
collective; torch.distributed (RCCL) is used only for the barrier / max-over-ranks / the final chain-gather).

Also reports, on the same JSON line:
  roofline      -- the filter's associative scan (dominant kernel group: k_scan_reduce_cm + k_scan_aggs + k_scan_down_cm<FilterOp>);
                   algorithmic bytes = K3 n(3d^2+2d)s read + n(d^2+d)s written per chain, divided by its HIP-event
                   duration measured inside the timed region on the library's stream.  The marginal log-likelihood (K4, the
                   reference's second pass over the filtered moments) is the log-scale the scan elements carry: it costs no pass.
  general_path  -- the same workload with nothing hoisted out of the chain loop (what chain-specific parameters cost); `value`
                   is the default mode: the model's parameters are the same for every chain, so the element matrices, gains and
                   Cholesky factors are computed once per time step and sweep (what jax.vmap leaves unbatched in the reference).
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


def run_csmc(args, rank, world, local_rank, dist, torch, coll_dev):
    """Secondary workload: BASELINE configs[2] = C3, SV d=1 T=65536, auxiliary cSMC with independent proposals, N=1024,
    backward sampling, fp32, in-kernel Threefry noise.  One step = one sweep of every chain on this GPU."""
    import ctypes as C
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics, SVPotential
    from aux_ssm_samplers_amd.parallel import gather_chains
    T, N, Cn = args.T, args.N, args.chains
    dtype = np.float32 if args.dtype == "f32" else np.float64
    phi, q, xtrue, y = sv_data(T)
    M0 = GaussianInit(m0=[0.0], P0=[[q]])
    Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
    fk = _device.describe_independent(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), Mt)
    handle = _lib.default_handle(local_rank)
    rng = np.random.Generator(np.random.PCG64(1000 + rank))
    x0 = (xtrue[None] + 0.1 * rng.standard_normal((Cn, T, 1))).astype(dtype)
    xd = handle.to_device(x0)
    anc = handle.zeros((Cn, T), np.int32)
    yd = fk.ydev(handle, dtype)
    shd = handle.to_device(np.full(T, np.sqrt(0.25)), dtype)
    m = _lib.FkModel(fk.proposal, fk.potential, 1, fk.transition, fk.m0.ctypes.data, fk.chol_P0.ctypes.data, fk.F.ctypes.data,
                     fk.b.ctypes.data, fk.chol_Q.ctypes.data, yd.ptr.value, 1.0)
    keys = R.split(R.PRNGKey(77 + rank), args.steps + args.warmup + 1)

    def step(k):
        nz = _lib.CsmcNoise()
        nz.mode, nz.key0, nz.key1 = _lib.NOISE_THREEFRY, int(keys[k][0]), int(keys[k][1])
        _lib.check(handle.lib.auxssm_csmc_sweep(handle.h, _lib.dtype_code(dtype), C.byref(m), Cn, T, N, 1, shd.ptr, xd.ptr,
                                                C.byref(nz), anc.ptr, None, None, None))

    def barrier():
        handle.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        handle.sync()

    for k in range(args.warmup):
        step(k)
    barrier()
    handle.prof_enable(_lib.K_CSMC_FWD, args.steps + 1)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    fn, fms = handle.prof_read()
    handle.prof_disable()
    moved = (anc.to_host() != 0).mean(axis=1)  # per-chain fraction of updated time steps
    if dist is not None:
        g = gather_chains(moved[:, None], Cn * world, dist, dst=0, device=coll_dev)
        moved = g[:, 0] if rank == 0 else moved
    if rank == 0:
        s = np.dtype(dtype).itemsize
        alg = Cn * T * N * (1 * s + s)  # forward pass writes xs + log_ws (As is not stored with backward sampling)
        roof = None
        if fn:
            ach = alg / (fms / fn * 1e-3) / 1e9
            roof = dict(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4), traffic=None,
                        kernel="k_csmc_fwd<float,1> (persistent forward pass)", avg_launch_ms=round(fms / fn, 3), launches=fn,
                        algorithmic_bytes_per_launch=alg)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import csmc as O
            Tb = min(T, 4096)
            r2 = np.random.default_rng(0)
            od = dict(proposal=O.AUX_INDEPENDENT, potential=O.POT_SV, m0=[0.0], chol_P0=[[np.sqrt(q)]], F=[[phi]], b=[0.0], chol_Q=[[np.sqrt(q)]])
            kw = dict(y=y[:Tb], sqrt_half_delta=np.full(Tb, 0.5), eps_aux=r2.standard_normal((Tb, 1)), eps_prop=r2.standard_normal((Tb, N, 1)),
                      u_res=r2.random((Tb - 1, N)), u_bwd=r2.random(Tb))
            t1 = time.perf_counter()
            O.sweep(od, xtrue[:Tb], N, True, dtype=np.float32, **kw)
            dt = time.perf_counter() - t1
            cpu = dict(value=1.0 / (dt * T / Tb), unit="sweeps/s", cores=1, kind="port",
                       sample=f"1 sweep of 1 chain at T={Tb} (scaled linearly to T={T}), N={N}, C oracle csmc_ref.c, {dt:.1f} s")
        print(json.dumps({
            "metric": "Gibbs sweeps/sec (auxiliary cSMC, backward sampling)", "value": round(Cn * world * args.steps / el, 2),
            "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C3: stochastic volatility d=1 T={T}, auxiliary cSMC N={N}, independent proposals, backward sampling",
                       "chains_per_gpu": Cn, "delta": 0.5, "parallelism": f"chains x{world} (independent, no collective)"},
            "updated_fraction": float(np.mean(moved)), "roofline": roof, "cpu_baseline": cpu}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=256, help="chains per GPU (SURVEY 8d lists 1, 8, 64, 256 for C2)")
    ap.add_argument("--T", type=int, default=65536)
    ap.add_argument("--d", type=int, default=4)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket the scan with HIP events")
    ap.add_argument("--no-share-model", action="store_true",
                    help="force the general per-chain path (AUXSSM_OPT_SHARE_MODEL = 0): nothing is hoisted out of the chain loop even though "
                         "the parameters of this linear-Gaussian model are the same for every chain")
    ap.add_argument("--no-general-leg", action="store_true", help="skip the extra (untimed-for-value) run of the general per-chain path")
    ap.add_argument("--workload", default="kalman", choices=["kalman", "csmc"],
                    help="kalman = BASELINE configs[1] (C2, the headline); csmc = configs[2] (C3), secondary")
    ap.add_argument("--N", type=int, default=1024, help="particles (csmc workload)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one rank per GPU) is what the driver runs; gloo + ranks sharing a GPU is a rehearsal mode")
    args = ap.parse_args()
    if args.workload == "csmc" and args.dtype == "f64" and "--dtype" not in " ".join(sys.argv):
        args.dtype = "f32"

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    ndev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(ndev, 1)  # rehearsal: ranks may share a GPU
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run: initialise the group even for one rank
    if world > 1 or launched:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else torch.device("cpu")

    if args.workload == "csmc":
        return run_csmc(args, rank, world, local_rank, dist, torch, coll_dev)

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
    share = not args.no_share_model
    handle.set_option(_lib.OPT_SHARE_MODEL, int(share))
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
        t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    scan_n, scan_ms = (0, 0.0)
    if not args.no_prof:
        scan_n, scan_ms = handle.prof_read()
        handle.prof_disable()

    # second leg (reported beside the headline, never as `value`): the same workload on the general per-chain path, i.e. what a
    # model with chain-specific parameters (any nonlinear model: a linearisation per chain) costs
    general = None
    if share and not args.no_general_leg and not args.no_prof:
        handle.set_option(_lib.OPT_SHARE_MODEL, 0)
        gsteps = max(3, args.steps // 2)
        step(args.warmup + args.steps)  # warm the other code path (workspace, code objects)
        barrier()
        handle.prof_enable(_lib.K_FILTER_SCAN, gsteps + 1)
        barrier()
        g0 = time.perf_counter()
        for k in range(gsteps):
            step(k % (args.steps + args.warmup))
        barrier()
        gel = time.perf_counter() - g0
        if dist is not None:
            t = torch.tensor([gel], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gel = float(t.item())
        gn, gms = handle.prof_read()
        handle.prof_disable()
        handle.set_option(_lib.OPT_SHARE_MODEL, 1)
        general = dict(steps=gsteps, value=C * world * gsteps / gel, ms_per_step=gel / gsteps * 1e3, scan_ms=gms / max(gn, 1))

    # the trivial chain-gather (RCCL): acceptance flags + last log-alphas of every chain to rank 0
    from aux_ssm_samplers_amd.parallel import gather_chains
    acc = chains.accepted.to_host()
    logs = chains.logs.to_host()
    if dist is not None:
        per_chain = np.concatenate([acc[:, None].astype(np.float64), logs.astype(np.float64)], axis=1)  # (C, 6)
        g = gather_chains(per_chain, C * world, dist, dst=0, device=coll_dev)
        if rank == 0:
            acc, logs = g[:, 0], g[:, 1:]

    if rank == 0:
        s = np.dtype(dtype).itemsize
        n = T - 1
        # the timed group = the filter's associative scan (K3: read n(3d^2+2d)s, write n(d^2+d)s per chain); the marginal
        # log-likelihood (K4, filtering.py:60-62) rides along as the elements' log-scale, so it adds no algorithmic bytes
        alg_bytes = C * (n * (3 * d * d + 2 * d) * s + n * (d * d + d) * s)
        roof = None
        traffic = None  # HBM bytes per launch group from the PMC passes committed under profiles/ (same config only)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            ent = tj.get(f"kalman_C2_{args.dtype}_T{T}_d{d}_chains{C}" + ("_shared_model" if share else ""), {})
            traffic = ent.get("filter_scan_group_hbm_bytes")
        except Exception:
            pass
        if scan_n:
            avg_s = scan_ms / scan_n * 1e-3
            ach = alg_bytes / avg_s / 1e9
            roof = dict(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4),
                        traffic=traffic, kernel="filter associative scan incl. the marginal log-likelihood (k_scan_reduce_cm + k_scan_aggs + k_scan_down_cm, " +
                               ("FilterOpShared: element matrices read once per time step, (b, eta, z) per chain)" if share else "FilterOp: general per-chain elements)"),
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
        out["config"]["model_sharing"] = ("chain-shared model parameters hoisted out of the chain loop (jax.vmap semantics)" if share
                                          else "off: general per-chain path")
        if general is not None:
            gach = alg_bytes / (general["scan_ms"] * 1e-3) / 1e9 if general["scan_ms"] else None
            out["general_path"] = {"value": round(general["value"], 2), "unit": "sweeps/s", "steps": general["steps"],
                                   "ms_per_step": round(general["ms_per_step"], 4), "scan_avg_launch_ms": round(general["scan_ms"], 4),
                                   "scan_achieved_GBps": None if gach is None else round(gach, 1),
                                   "scan_frac": None if gach is None else round(gach / 8000.0, 4),
                                   "note": "same workload with AUXSSM_OPT_SHARE_MODEL = 0: every chain builds and combines its own d x d elements"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
