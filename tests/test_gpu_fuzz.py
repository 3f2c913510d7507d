"""Randomized differential test of the device sweeps (LG_CONCAT / SV first and second order / Lorenz): random horizon (tile and wave
boundaries included), chain count (both sides of the 32-chain layout switch and of the 64-lane wave), layout, parallel / sequential scan,
chain-shared tables on / off, missing observations -- every case against the oracle's sweep on the same explicit noise, for the first, the
last and one random chain.  Fixed seeds; 1200 further random cases of the same generator were run clean before committing."""
import numpy as np
import pytest

from oracle import kalman_np as K
from tests.helpers import lg_model, sv_setup, lorenz_kalman_setup

pytestmark = pytest.mark.gpu


def _seeds(default):
    """AUXSSM_FUZZ_SEEDS="100:120" widens a run (extra seeds 100..119 for every generator) without touching the committed defaults"""
    import os
    e = os.environ.get("AUXSSM_FUZZ_SEEDS")
    if not e:
        return default
    lo, hi = (int(v) for v in e.split(":"))
    return list(range(lo, hi))


@pytest.mark.parametrize("seed", _seeds([11, 12, 13]))
def test_random_shapes_vs_oracle(seed):
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel, SVModel, LorenzModel, DeviceChains, KalmanSampler
    ncase = 60
    h = _lib.default_handle()
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncase):
        kind = rng.choice(["lg", "lg", "sv1", "sv2", "lorenz"])
        T = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 127, 130, 257, 300]))
        C = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 100, 129]))
        parallel = bool(rng.integers(0, 2))
        cmin = bool(rng.integers(0, 2)) if C < 32 else True
        share = int(rng.integers(0, 2))
        bt = np.broadcast_to
        if kind == "lg":
            d = int(rng.choice([1, 2, 4]))
            m = lg_model(max(T, 2), d)
            y = m["y"][:T].copy()
            y[rng.random(T) < 0.2] = np.nan
            model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (max(T - 1, 0), d, d)), bt(m["Q"], (max(T - 1, 0), d, d)), bt(m["b"], (max(T - 1, 0), d)),
                                  bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), y)
            lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
            target = lambda z: K.log_likelihood(y, z, lgo) + K.prior_logpdf(z, lgo)
            xtrue, delta, tol = m["x_true"][:T], 0.5, 1e-8
        elif kind in ("sv1", "sv2"):
            d = int(rng.integers(1, 5))
            y, xtrue, (m0, P0, F, Q, b) = sv_setup(max(T, 2), d, seed=case)
            model = SVModel(y[:T], m0, P0, F, Q, b, order=1 if kind == "sv1" else 2)
            lg = (model.m0, model.P0, model.Fs, model.Qs, model.bs, None, None, None)
            target = lambda z: K.prior_logpdf(z, lg) + model.log_potential(z)
            xtrue, delta, tol = xtrue[:T], 0.3, 1e-8
        else:
            d = 3
            base, xtrue = lorenz_kalman_setup(max(T, 9), seed=case)
            model = LorenzModel(base.yobs[:T], base.Hobs[:T], base.Robs[:T], base.cobs[:T], base.m0, base.P0, base.theta, base.sigma_x, base.dt)
            target = model.log_likelihood_fn
            xtrue, delta, tol = xtrue[:T], 0.02, 1e-7
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel)
        x0 = xtrue[None] + 0.1 * rng.standard_normal((C, T, d))
        noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
        h.set_option(_lib.OPT_SHARE_MODEL, share)
        try:
            chains = DeviceChains(h, x0, chain_minor=cmin)
            kernel(None, KalmanSampler(x=chains, updated=None), delta, noise=noise)
            xs, acc, logs = chains.to_host(), chains.accepted.to_host(), chains.logs.to_host()
        finally:
            h.set_option(_lib.OPT_SHARE_MODEL, 1)
        for c in sorted(set([0, C - 1, int(rng.integers(0, C))])):
            ref = K.kalman_sweep(x0[c], delta, model.dynamics_factory, model.observations_factory, target, parallel, eps_aux=noise["eps_aux"][c],
                                 eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
            ex = np.max(np.abs(xs[c] - ref["x"]) / (1e-2 + np.abs(ref["x"])))
            rl = np.array([ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]])
            el = np.max(np.abs(logs[c, 1:] - rl) / (1 + np.abs(rl)))
            flip = bool(acc[c]) != ref["accepted"]
            near = abs(np.log(max(noise["u_accept"][c], 1e-300)) - min(0.0, ref["log_alpha"])) < 1e-6
            if (ex > 100 * tol and not flip) or el > 10 * tol or (flip and not near):
                print("CASE", case, kind, "T", T, "C", C, "d", d, "par", parallel, "cm", cmin, "share", share, "chain", c, f"ex {ex:.2e} el {el:.2e} flip {flip}")
                bad += 1
    assert bad == 0


@pytest.mark.parametrize("seed", _seeds([21, 22]))
def test_random_particle_sweeps_bit_exact_vs_c_oracle(seed):
    """Sequential cSMC (bootstrap / auxiliary proposals, ancestor tracing / backward sampling) and the parallel-in-time sweep at random
    (d, N, T, dtype, potential): ancestors and trajectories bit-exact against oracle/csmc_ref.c.  450 further random cases were run clean."""
    from oracle import csmc as O
    from tests.test_gpu_csmc import _models, _odesc, _pot
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(seed)
    ncase = 60
    bad = 0
    for case in range(ncase):
        d = int(rng.integers(1, 5)); N = int(rng.choice([2, 3, 31, 32, 33, 63, 64, 65, 100, 128, 129, 255, 256, 257, 500, 1000, 1024]))
        T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 50, 64, 65]))
        dtype = [np.float32, np.float64][int(rng.integers(0, 2))]
        potential = int(rng.choice([O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV]))
        pit = bool(rng.integers(0, 2)) and T >= 2
        backward = bool(rng.integers(0, 2))
        proposal = O.AUX_INDEPENDENT if pit else int(rng.choice([O.BOOTSTRAP_LG, O.AUX_INDEPENDENT]))
        M0, Mt = _models(d, rng)
        y = rng.standard_normal((T, d))
        G0, Gt = _pot(potential, y)
        x0 = rng.standard_normal((T, d)).astype(dtype)
        delta = 0.5 + rng.random(T)
        cvt = lambda a: np.asarray(a, dtype)
        try:
            if pit:
                fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
                noise = dict(eps_aux=cvt(rng.standard_normal((T, d))), eps_prop=cvt(rng.standard_normal((T, N, d))), u_res=cvt(rng.random((T, N))))
                x, anc = _device.pit_sweep(fk, x0, N, noise={k: v[None] for k, v in noise.items()}, delta=delta)
                ref = O.pit_sweep(_odesc(O.AUX_INDEPENDENT, potential, M0, Mt, 0.7), x0, N, y=y if potential else None,
                                  sqrt_half_delta=np.sqrt(0.5 * delta), dtype=dtype, **noise)
            else:
                noise = dict(eps_prop=cvt(rng.standard_normal((T, N, d))), u_res=cvt(rng.random((max(T - 1, 0), N))), u_bwd=cvt(rng.random(T)))
                okw = {}
                if proposal == O.AUX_INDEPENDENT:
                    noise["eps_aux"] = cvt(rng.standard_normal((T, d)))
                    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
                    okw = dict(sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
                else:
                    fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
                x, anc, _ = _device.sweep(fk, x0, N, backward, noise={k: v[None] for k, v in noise.items()}, delta=delta if proposal == O.AUX_INDEPENDENT else None)
                ref = O.sweep(_odesc(proposal, potential, M0, Mt, 0.7), x0, N, backward, y=y if potential else None, eps_prop=noise["eps_prop"],
                              u_res=noise["u_res"], u_bwd=noise["u_bwd"], dtype=dtype, **okw)
            if not (np.array_equal(anc, ref["ancestors"]) and np.array_equal(x, ref["x"])):
                print("CASE", case, "pit" if pit else "seq", "d", d, "N", N, "T", T, dtype.__name__, "pot", potential, "prop", proposal, "bwd", backward, "MISMATCH")
                bad += 1
        except Exception as e:
            print("CASE", case, "pit" if pit else "seq", d, N, T, dtype.__name__, potential, proposal, backward, "EXC", repr(e)[:300]); bad += 1
    assert bad == 0


@pytest.mark.parametrize("seed", _seeds([31, 32]))
def test_random_primitives_vs_oracle(seed):
    """filtering / sampling / posterior_logpdf at random (dx, dy, T) on both sides of the per-lane / wide-state switch (dx up to 40, dy up to
    40), parallel and sequential, missing observations: fp64, 1e-7 relative to the oracle.  300 further random cases were run clean."""
    from tests.test_gpu_wide import stable_model
    import aux_ssm_samplers_amd._primitives.kalman as P
    rng = np.random.default_rng(seed)
    ncase = 60
    bad = 0
    for case in range(ncase):
        wide = rng.random() < 0.6
        if wide:
            d = int(rng.integers(1, 41)); p = int(rng.integers(1, 41))
            if d <= 4 and p <= 8:
                d = int(rng.integers(5, 41))
        else:
            d = int(rng.integers(1, 5)); p = int(rng.integers(1, 9))
        T = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 70]))
        parallel = bool(rng.integers(0, 2))
        ys, lg = stable_model(rng, max(T, 2), d, p)
        m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
        ys, lg = ys[:T], (m0, P0, Fs[:T - 1], Qs[:T - 1], bs[:T - 1], Hs[:T], Rs[:T], cs[:T])
        eps = rng.standard_normal((T, d))
        try:
            ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), parallel)
            xs = P.sampling(None, ms, Ps, P.LGSSM(*lg), parallel, eps=eps)
            lp = P.posterior_logpdf(ys, xs, ell, P.LGSSM(*lg))
            oms, oPs, oell = K.filtering(ys, lg, parallel)
            oxs = K.sampling(eps, oms, oPs, lg, parallel)
            olp = K.posterior_logpdf(ys, oxs, oell, lg)
            e1 = np.max(np.abs(ms - oms)) / (1 + np.max(np.abs(oms)))
            e2 = np.max(np.abs(Ps - oPs)) / (1 + np.max(np.abs(oPs)))
            e3 = abs(ell - oell) / (1 + abs(oell))
            e4 = np.max(np.abs(xs - oxs)) / (1 + np.max(np.abs(oxs)))
            e5 = abs(lp - olp) / (1 + abs(olp))
            if max(e1, e2, e3, e4, e5) > 1e-7 or not np.isfinite([e1, e2, e3, e4, e5]).all():
                print("CASE", case, "d", d, "p", p, "T", T, "par", parallel, f"{e1:.1e} {e2:.1e} {e3:.1e} {e4:.1e} {e5:.1e}")
                bad += 1
        except Exception as e:
            print("CASE", case, d, p, T, parallel, "EXC", repr(e)[:300]); bad += 1
    assert bad == 0
