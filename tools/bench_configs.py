#!/usr/bin/env python3
"""Secondary measurements on the BASELINE configs that are not the headline (bench.py measures C2 and, with --workload csmc, C3):
   C3k  stochastic volatility d=1 T=65536, auxiliary-Kalman sweep (first / second order), fp64
   C4   Lorenz-63 T=16384 dt=1.25e-4 obs every 80 steps, fp32: Kalman sweep (extended linearisation) + cSMC sweep N=512 (bootstrap, backward sampling)
   C5   dense d = p = 64 T=8192 fp32: filter + sampler + joint log-density of one chain (wide-state path)
   loop the MCMC loop around the sweeps (aux_ssm_samplers_amd.loop: running moments, acceptance averages, adaptation, Lorenz theta step) on C2 / C3 / C4
Prints one JSON line per measurement.  Inputs are resident in HBM (DeviceChains) where the API allows it; device Threefry noise."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel, SVModel, LorenzModel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402


def timed_sweeps(kernel, chains, delta, steps=5, warmup=2, seed=1):
    h = chains.handle
    state = KalmanSampler(x=chains, updated=None)
    keys = R.split(R.PRNGKey(seed), steps + warmup)
    for k in range(warmup):
        kernel(keys[k], state, delta)
    h.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        kernel(keys[warmup + k], state, delta)
    h.sync()
    el = time.perf_counter() - t0
    return chains.C * steps / el, el / steps * 1e3, float(chains.accepted.to_host().mean())


def c3_kalman(order, chains=64, T=65536, chain_minor=None):
    from aux_ssm_samplers_amd.workloads import sv_setup
    from aux_ssm_samplers_amd.common import delta_adaptation
    from aux_ssm_samplers_amd.loop import loop
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, 1, rho=0.0)
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    ch = DeviceChains(h, np.repeat(xtrue[None], chains, axis=0), chain_minor=chain_minor)
    # burn-in with the reference's adaptation rule so that the timed sweeps run at a step size that moves (target 0.234)
    out = loop(R.PRNGKey(0), 0.05, KalmanSampler(x=ch, updated=True), kernel, delta_adaptation, 400, target_alpha=0.234, lr=0.3, beta=0.2)
    delta = out[3]
    v, ms, acc = timed_sweeps(kernel, ch, delta, steps=10)
    print(json.dumps(dict(config=f"C3 SV d=1 T={T}, aux-Kalman order {order}, fp64, {'chain-minor' if ch.chain_minor else 'dense'} layout", chains=chains,
                          delta=float(f"{delta:.3g}"), sweeps_per_s=round(v, 1), ms_per_step=round(ms, 2), accept=acc)), flush=True)


def c4(chains=8, T=16384, N=512):
    from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup
    from aux_ssm_samplers_amd.workloads import lorenz_setup
    from aux_ssm_samplers_amd.csmc import _device
    model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    ch = DeviceChains(h, np.repeat(xtrue[None], chains, axis=0).astype(np.float32), chain_minor=False)
    v, ms, acc = timed_sweeps(kernel, ch, 1e-4)
    print(json.dumps(dict(config=f"C4 Lorenz-63 T={T} dt=1.25e-4 obs/80, aux-Kalman (extended linearisation), fp32", chains=chains,
                          sweeps_per_s=round(v, 1), ms_per_step=round(ms, 2), accept=acc)))
    from aux_ssm_samplers_amd.csmc import CsmcChains, CSMCState
    from aux_ssm_samplers_amd._primitives.csmc import get_kernel as get_csmc_kernel
    M0, Mt, G0, Gt, xt, y, sig_y = lorenz_setup(T, every=80, dt=1.25e-4)
    init, ck = get_csmc_kernel(M0, G0, Mt, Gt, N, backward=True, Pt=Mt)
    cc = CsmcChains(h, np.repeat(xt[None], chains, axis=0).astype(np.float32))
    st = CSMCState(x=cc, updated=None)
    ck(0, st)
    h.sync()
    t0 = time.perf_counter()
    reps = 3
    for k in range(reps):
        ck(1 + k, st)
    h.sync()
    el = time.perf_counter() - t0
    print(json.dumps(dict(config=f"C4 Lorenz-63 T={T}, cSMC N={N} bootstrap + backward sampling, fp32, resident chains", chains=chains,
                          sweeps_per_s=round(chains * reps / el, 1), ms_per_step=round(el / reps * 1e3, 2),
                          updated=float((cc.ancestors.to_host() != 0).mean()))))


def c5(T=8192):
    from aux_ssm_samplers_amd.workloads import c5_model
    import aux_ssm_samplers_amd._primitives.kalman as P
    u, lg64, x = c5_model(T, 64)
    lg = P.LGSSM(*[np.ascontiguousarray(a, np.float32) for a in lg64])
    u = u.astype(np.float32)
    h = _lib.default_handle()
    eps = np.random.default_rng(0).standard_normal((T, 64)).astype(np.float32)
    for rep in range(2):
        for kid, name in ((_lib.K_FILTER_INIT, "init"), (_lib.K_FILTER_SCAN, "scan (incl. log-likelihood)")):
            h.prof_enable(kid, 4)
            ms, Ps, ell = P.filtering(u, lg, True)
            n, t = h.prof_read()
            h.prof_disable()
            if rep:
                print(json.dumps(dict(config=f"C5 dense d=p=64 T={T} fp32, 1 chain, filter {name} kernels", ms=round(t, 2))))
        for kid, name in ((_lib.K_SAMPLE_INIT, "sampler init"), (_lib.K_SAMPLE_SCAN, "sampler scan")):
            h.prof_enable(kid, 4)
            xs = P.sampling(None, ms, Ps, lg, True, eps=eps)
            n, t = h.prof_read()
            h.prof_disable()
            if rep:
                print(json.dumps(dict(config=f"C5 dense d=p=64 T={T} fp32, 1 chain, {name} kernels", ms=round(t, 2))))
        h.prof_enable(_lib.K_LOGPDF, 4)
        lp = P.posterior_logpdf(u, xs, ell, lg)
        n, t = h.prof_read()
        h.prof_disable()
        if rep:
            print(json.dumps(dict(config=f"C5 dense d=p=64 T={T} fp32, 1 chain, joint log-density kernel", ms=round(t, 2), lp=float(lp))))


def timed_loop(label, kernel, state, delta, n_iter=40, **kw):
    """loop() twice (first run warms code objects and the workspace); wall time of the second, host-synchronised at the end only"""
    from aux_ssm_samplers_amd.loop import loop
    chains = state.x
    h = chains.handle
    delta_fn = kw.pop("delta_fn", None)
    for rep in range(2):
        out = None  # frees the previous run's moment arrays outside the timed region
        h.sync()
        t0 = time.perf_counter()
        out = loop(R.PRNGKey(5 + rep), delta, state, kernel, delta_fn, n_iter, **kw)
        h.sync()
        el = time.perf_counter() - t0
    print(json.dumps(dict(config=label, chains=chains.C, sweeps_per_s=round(chains.C * n_iter / el, 1), ms_per_step=round(el / n_iter * 1e3, 3),
                          avg_accept=round(float(out[5].to_host().mean()), 3))))
    return out


def loops():
    import bench
    from aux_ssm_samplers_amd.common import delta_adaptation
    from aux_ssm_samplers_amd.loop import LorenzThetaStep
    h = _lib.default_handle()
    # C2: 256 chains, chain-minor, chain-shared model
    T, d, C = 65536, 4, 256
    m, model = bench.build_model(T, d, np.float64)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    x0 = m["x_true"][None] + 0.3 * np.random.default_rng(0).standard_normal((C, T, d))
    ch = DeviceChains(h, x0)
    v, ms, acc = timed_sweeps(kernel, ch, 0.5, steps=20, warmup=3)
    print(json.dumps(dict(config="C2 LG-SSM T=65536 d=4 fp64, bare sweeps (no statistics)", chains=C, sweeps_per_s=round(v, 1), ms_per_step=round(ms, 3))))
    timed_loop("C2 LG-SSM T=65536 d=4 fp64, loop(): sweeps + running sq-jump/mean/sq-mean + acceptance averages, no host sync", kernel,
               KalmanSampler(x=ch, updated=True), 0.5, beta=0.01)
    timed_loop("C2 ..., loop() while adapting delta (one 2 KB read-back per sweep)", kernel, KalmanSampler(x=ch, updated=True), 0.5, beta=0.01,
               delta_fn=delta_adaptation, target_alpha=0.5, lr=0.1)
    del ch
    # C4: the (x, theta) Gibbs sampler of the Lorenz example, 8 chains each with its own theta
    from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup
    from aux_ssm_samplers_amd.workloads import lorenz_setup
    T, C = 16384, 8
    model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    ch = DeviceChains(h, np.repeat(xtrue[None], C, axis=0).astype(np.float32), chain_minor=False)
    v, ms, acc = timed_sweeps(kernel, ch, 1e-4, steps=20, warmup=3)
    print(json.dumps(dict(config="C4 Lorenz-63 T=16384 fp32, bare Kalman sweeps", chains=C, sweeps_per_s=round(v, 1), ms_per_step=round(ms, 3))))
    step = LorenzThetaStep(model, 1e3 ** 0.5)  # sigma_theta of examples/lorenz/experiment.py:75
    timed_loop("C4 Lorenz-63 T=16384 fp32, loop(): Kalman sweep + theta | x draw per chain + running statistics, no host sync", kernel,
               KalmanSampler(x=ch, updated=True), 1e-4, beta=0.01, theta_step=step)
    print(json.dumps(dict(theta_after=np.round(step.theta(ch), 3).tolist())))
    # C3 / C4 cSMC on resident chains
    from aux_ssm_samplers_amd.csmc import CsmcChains, CSMCState, get_independent_kernel, GaussianInit, LinearGaussianDynamics, SVPotential
    from aux_ssm_samplers_amd._primitives.csmc import get_kernel as get_csmc_kernel
    M0, Mt, G0, Gt, xt, y, sig_y = lorenz_setup(T, every=80, dt=1.25e-4)
    init, ck = get_csmc_kernel(M0, G0, Mt, Gt, 512, backward=True, Pt=Mt)
    cc = CsmcChains(h, np.repeat(xt[None], C, axis=0).astype(np.float32))
    timed_loop("C4 Lorenz-63 T=16384 fp32, loop(): cSMC N=512 bootstrap + backward sampling + per-step statistics, resident chains", lambda k, s, dl: ck(k, s),
               CSMCState(x=cc, updated=np.ones(T, bool)), None, n_iter=4, beta=0.01)
    T3, C3 = 65536, 16
    phi, q, xsv, ysv = bench.sv_data(T3, 0)
    M0 = GaussianInit(m0=[0.0], P0=[[q]])
    Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
    init, ik = get_independent_kernel(M0, SVPotential(y=ysv[0]), Mt, SVPotential(params=ysv[1:]), 1024, backward=True, Pt=Mt)
    cc = CsmcChains(h, np.repeat(xsv.reshape(1, T3, 1), C3, axis=0).astype(np.float32))
    timed_loop("C3 SV T=65536 fp32, loop(): aux-cSMC N=1024 backward sampling + per-step statistics + per-step delta adaptation on device", ik,
               CSMCState(x=cc, updated=np.zeros(T3, bool)), 0.5, n_iter=4, beta=0.01, delta_fn=delta_adaptation, target_alpha=0.5, lr=0.1)


def pit(T=65536):
    """C3's model (SV, d = 1, T = 65536, fp32): parallel-in-time cSMC (auxssm_csmc_pit_sweep) vs the sequential sweep at the same N,
    few chains (where the sequential sweep is latency-bound: T dependent steps)"""
    import bench
    from aux_ssm_samplers_amd.csmc import CsmcChains, CSMCState, get_independent_kernel, GaussianInit, LinearGaussianDynamics, SVPotential
    h = _lib.default_handle()
    phi, q, xsv, ysv = bench.sv_data(T, 0)
    M0 = GaussianInit(m0=[0.0], P0=[[q]])
    Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
    for N, chains in ((32, 1), (32, 16), (64, 1), (64, 16), (256, 1), (256, 8), (1024, 1), (1024, 4)):
        row = dict(config=f"C3 SV T={T} fp32, N={N}", chains=chains)
        for par in (True, False):
            init, k = get_independent_kernel(M0, SVPotential(y=ysv[0]), Mt, SVPotential(params=ysv[1:]), N, backward=not par, Pt=Mt, parallel=par)
            cc = CsmcChains(h, np.repeat(xsv.reshape(1, T, 1), chains, axis=0).astype(np.float32), delta=0.5)
            st = CSMCState(x=cc, updated=None)
            keys = R.split(R.PRNGKey(3), 8)
            k(keys[0], st, None)
            h.sync()
            reps = 5 if par else 2
            t0 = time.perf_counter()
            for i in range(reps):
                k(keys[1 + i], st, None)
            h.sync()
            el = (time.perf_counter() - t0) / reps
            name = "pit" if par else "sequential_backward_sampling"
            row[name + "_ms_per_sweep"] = round(el * 1e3, 2)
            row[name + "_updated"] = round(float((cc.ancestors.to_host() != 0).mean()), 3)
        row["speedup"] = round(row["sequential_backward_sampling_ms_per_sweep"] / row["pit_ms_per_sweep"], 1)
        print(json.dumps(row), flush=True)


def c5_batched_scalar(T=8192, B=64):
    """C5 as the reference's spatial example actually runs it (examples/spatial/model.py:103-112, auxiliary_kalman.py:18-28): d^2 = 64
    INDEPENDENT scalar chains on the batch axis B, not one dense 64 x 64 state.  AR(1) rows of the grid model, first-order aux observations."""
    import aux_ssm_samplers_amd._primitives.kalman as P
    rng = np.random.default_rng(0)
    delta = 0.1
    f32 = np.float32
    lg = P.LGSSM(np.zeros((B, 1), f32), np.ones((B, 1, 1), f32), np.full((T - 1, B, 1, 1), 0.9, f32), np.ones((T - 1, B, 1, 1), f32),
                 np.zeros((T - 1, B, 1), f32), np.ones((T, B, 1, 1), f32), np.full((T, B, 1, 1), delta / 2, f32), np.zeros((T, B, 1), f32))
    u = rng.standard_normal((T, B, 1)).astype(f32)
    eps = rng.standard_normal((T, B, 1)).astype(f32)
    h = _lib.default_handle()
    for rep in range(2):
        out = {}
        for kid, name in ((_lib.K_FILTER_INIT, "filter_init_ms"), (_lib.K_FILTER_SCAN, "filter_scan_ms")):
            h.prof_enable(kid, 4)
            ms, Ps, ell = P.filtering(u, lg, True)
            out[name] = round(h.prof_read()[1], 4)
            h.prof_disable()
        for kid, name in ((_lib.K_SAMPLE_INIT, "sampler_init_ms"), (_lib.K_SAMPLE_SCAN, "sampler_scan_ms")):
            h.prof_enable(kid, 4)
            xs = P.sampling(None, ms, Ps, lg, True, eps=eps)
            out[name] = round(h.prof_read()[1], 4)
            h.prof_disable()
    print(json.dumps(dict(config=f"C5 as B = {B} independent scalar chains on the batch axis (the reference's spatial example), T={T} fp32, per-lane kernels",
                          **out)))


def sv30(T=250, D=30, N=25):
    """the reference's own timed stochastic-volatility protocol (examples/stochastic_volatility/experiment.sh:1-10: D = 30, T = 250, cSMC with N = 25
    particles, backward sampling): auxiliary cSMC sweeps per second at several chain counts, classical sweep (csrc/csmc_wide.hip)"""
    from aux_ssm_samplers_amd.workloads import sv_setup
    from aux_ssm_samplers_amd.csmc import CsmcChains, CSMCState, get_independent_kernel, GaussianInit, LinearGaussianDynamics, SVPotential
    h = _lib.default_handle()
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, D)
    M0, Mt = GaussianInit(m0=m0, P0=P0), LinearGaussianDynamics(F=F, b=b, Q=Q)
    for gradient in (False, True, "exact"):  # (--gradient of the protocol: experiment.py:18-57; "exact" = the per-particle weighting, AUXSSM_GRAD_EXACT)
        init, kernel = get_independent_kernel(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), N, backward=True, Pt=Mt, gradient=gradient)
        for chains in (1, 256, 4096):
            cc = CsmcChains(h, np.repeat(xtrue[None], chains, axis=0).astype(np.float32))
            st = CSMCState(x=cc, updated=None)
            kernel(0, st, 0.05)
            h.sync()
            reps = 5
            t0 = time.perf_counter()
            for k in range(reps):
                kernel(1 + k, st, None)
            h.sync()
            el = time.perf_counter() - t0
            print(json.dumps(dict(config=f"SV protocol D={D} T={T}, aux-cSMC N={N} independent proposals (gradient={gradient}) + backward sampling, fp32, resident chains",
                                  chains=chains, sweeps_per_s=round(chains * reps / el, 1), ms_per_sweep_call=round(el / reps * 1e3, 3),
                                  updated=float((cc.ancestors.to_host() != 0).mean()))), flush=True)


def sv30_kalman(T=250, D=30, chains=(1, 16, 64)):
    """the other sampler of the same protocol: the auxiliary Kalman sampler with first / second order linearisation of the SV observation model at D = 30
    (examples/stochastic_volatility/auxiliary_kalman.py:22-48), fp64 as the reference runs it, parallel scan; wide-state kernels (dx = 30)"""
    from aux_ssm_samplers_amd.workloads import sv_setup
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, D)
    h = _lib.default_handle()
    for order in (1, 2):
        model = SVModel(y, m0, P0, F, Q, b, order=order)
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        for C_ in chains:
            try:
                ch = DeviceChains(h, np.repeat(xtrue[None], C_, axis=0), chain_minor=False)
                v, ms, acc = timed_sweeps(kernel, ch, 0.01, steps=3, warmup=1)
                print(json.dumps(dict(config=f"SV protocol D={D} T={T}, aux-Kalman order {order}, fp64, parallel scan, wide-state kernels", chains=C_,
                                      sweeps_per_s=round(v, 1), ms_per_step=round(ms, 2), accept=acc)), flush=True)
            except Exception as e:  # noqa: BLE001
                print(json.dumps(dict(config=f"SV protocol D={D} T={T}, aux-Kalman order {order}", chains=C_, error=f"{type(e).__name__}: {e}")), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3k", "c4", "c5"]
    if "sv30" in which:
        sv30()
    if "sv30k" in which:
        sv30_kalman()
    if "pit" in which:
        pit()
    if "loop" in which:
        loops()
    if "c3k" in which:
        for chains, cmin in ((64, False), (64, None), (256, None), (1024, None)):
            c3_kalman(1, chains, chain_minor=cmin)
            c3_kalman(2, chains, chain_minor=cmin)
    if "c4" in which:
        c4()
    if "c5" in which:
        c5()
        c5_batched_scalar()
