"""The C restatement of the reference's SEQUENTIAL auxiliary-Kalman sweep (oracle/kalman_seq.c, the CPU baseline of bench.py) against
the NumPy oracle's sweep (oracle/kalman_np.py::kalman_sweep(parallel=False)), which is pinned to the reference's known answers."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from oracle import kalman_seq as S
from tests.helpers import lg_model
from aux_ssm_samplers_amd.kalman import LGConcatModel


def _np_sweep(model, m, x, delta, ea, es, ua):
    lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    target = lambda z: K.log_likelihood(model.yobs, z, lgo) + K.prior_logpdf(z, lgo)
    return K.kalman_sweep(x, delta, model.dynamics_factory, model.observations_factory, target, False, eps_aux=ea, eps_samp=es, u_accept=ua)


@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("nan", [False, True])
def test_c_sweep_equals_numpy_oracle_sequential_sweep(d, nan):
    T, Cn = 41, 3
    m = lg_model(T, d)
    bt = np.broadcast_to
    y = m["y"].copy()
    if nan:
        y[3] = np.nan          # a fully missing real observation
        y[7, 0] = np.nan       # a partially missing one (d > 1), or fully missing (d = 1)
    model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                          bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), y)
    cm = S.Model(m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs, y)
    rng = np.random.default_rng(d)
    x = m["x_true"][None] + 0.3 * rng.standard_normal((Cn, T, d))
    ea, es, ua = rng.standard_normal((Cn, T, d)), rng.standard_normal((Cn, T, d)), rng.random(Cn)
    out = S.sweep(cm, x, 0.5, ea, es, ua, nthreads=2, want_prop=True)
    for c in range(Cn):
        ref = _np_sweep(model, m, x[c], 0.5, ea[c], es[c], ua[c])
        npt.assert_allclose(out["x_prop"][c], ref["x_prop"], rtol=1e-9, atol=1e-10)
        npt.assert_allclose(out["logs"][c], [ref["log_alpha"], ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9, atol=1e-8)
        assert bool(out["accepted"][c]) == ref["accepted"]
        npt.assert_allclose(out["x"][c], ref["x"], rtol=1e-9, atol=1e-10)
        if not nan:  # a linear-Gaussian model with exact auxiliary observations: the MH ratio is 1 (with missing data the reference's
            assert abs(out["logs"][c, 0]) < 1e-7  # nansum drops partially observed steps from posterior_logpdf only: SURVEY 8(a) K7 quirk)


def test_threads_do_not_change_results():
    T, d, Cn = 64, 2, 5
    m = lg_model(T, d)
    cm = S.Model(m["m0"], m["P0"], m["F"], m["Q"], m["b"], m["Hobs"], m["Robs"], m["cobs"], m["y"])
    rng = np.random.default_rng(0)
    x = rng.standard_normal((Cn, T, d))
    ea, es, ua = rng.standard_normal((Cn, T, d)), rng.standard_normal((Cn, T, d)), rng.random(Cn)
    a = S.sweep(cm, x, 0.5, ea, es, ua, nthreads=1)
    b = S.sweep(cm, x, 0.5, ea, es, ua, nthreads=3)
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["logs"], b["logs"])
    assert S.max_threads() >= 1
