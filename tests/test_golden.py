"""Committed known-answer fixtures (tests/golden/kalman_known_answers.npz, made by tests/golden/make_golden.py):
the oracle (CPU) and the HIP path (GPU) against the stored explicit-filter / RTS-smoother answers on the
reference's own seeded test inputs."""
import os

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kalman_known_answers.npz"))
CASES = [str(c) for c in G["cases"]]
LG = ("m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs")


def _check_filter(filtering, name, parallel, tol):
    if name.startswith("batch"):
        lg = tuple(G[f"{name}/b{k}"] for k in LG)
        ms, Ps, ell = filtering(G[f"{name}/bys"], lg, parallel)
        T, B, dx = ms.shape
        npt.assert_allclose(ms.reshape(T, B * dx), G[f"{name}/ms"], **tol)
        dense = np.zeros((T, B * dx, B * dx))
        for b in range(B):
            dense[:, b * dx:(b + 1) * dx, b * dx:(b + 1) * dx] = Ps[:, b]
        npt.assert_allclose(dense, G[f"{name}/Ps"], **tol)
        npt.assert_allclose(ell, G[f"{name}/ell"], **tol)
    else:
        lg = tuple(G[f"{name}/{k}"] for k in LG)
        ms, Ps, ell = filtering(G[f"{name}/ys"], lg, parallel)
        npt.assert_allclose(ms, G[f"{name}/ms"], **tol)
        npt.assert_allclose(Ps, G[f"{name}/Ps"], **tol)
        if f"{name}/ell" in G:
            npt.assert_allclose(ell, G[f"{name}/ell"], **tol)


def _check_smoother(sampling, name, parallel, tol):
    lg = tuple(G[f"{name}/{k}"] for k in LG)
    ms, Ps = G[f"{name}/ms"], G[f"{name}/Ps"]
    T, dx = ms.shape
    mean = sampling(np.zeros((T, dx)), ms, Ps, lg, parallel)
    npt.assert_allclose(mean, G[f"{name}/sm"], **tol)
    J = np.stack([sampling(np.eye(T * dx)[k].reshape(T, dx), ms, Ps, lg, parallel) - mean for k in range(T * dx)], -1)
    npt.assert_allclose(np.einsum("tik,tjk->tij", J, J), G[f"{name}/sP"], **tol)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("parallel", [False, True])
def test_oracle_vs_golden(name, parallel):
    tol = dict(rtol=1e-7) if not parallel else dict(rtol=1e-6, atol=1e-9)  # test_filtering.py:52-55 uses rtol 1e-7
    if name.startswith("smooth"):
        _check_smoother(K.sampling, name, parallel, dict(rtol=1e-7, atol=1e-9))
    else:
        _check_filter(K.filtering, name, parallel, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("parallel", [False, True])
def test_hip_vs_golden(name, parallel):
    import aux_ssm_samplers_amd._primitives.kalman as P
    tol = dict(rtol=1e-6, atol=1e-9)
    if name.startswith("smooth"):
        _check_smoother(lambda eps, ms, Ps, lg, par: P.sampling(None, ms, Ps, P.LGSSM(*lg), par, eps=eps), name, parallel,
                        dict(rtol=1e-7, atol=1e-9))
    else:
        _check_filter(lambda ys, lg, par: P.filtering(ys, P.LGSSM(*lg), par), name, parallel, tol)


def test_c5_fixture_prefix_reproduces():
    """tests/golden/c5_T8192_known_answers.npz (the fp64 sequential filter of config C5 at T = 8192) -- the filter at time t depends on the data up
    to t only, so its first entries are regenerated here from a 600-step prefix (the whole file: python tests/golden/make_c5_fixture.py --check)."""
    import os
    from oracle import kalman_np as K
    from tests.helpers import c5_model
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_T8192_known_answers.npz"))
    u, lg, _ = c5_model(8192, 64)
    n = 600
    cut = lambda a, m: a[:m]
    lgp = (lg[0], lg[1], cut(lg[2], n - 1), cut(lg[3], n - 1), cut(lg[4], n - 1), cut(lg[5], n), cut(lg[6], n), cut(lg[7], n))
    ms, Ps, _ = K.filtering(u[:n], lgp, False)
    sel = ref["idx"] < n
    assert sel.sum() >= 3
    npt.assert_allclose(ms[ref["idx"][sel]], ref["ms"][sel], rtol=1e-10, atol=1e-12)
    npt.assert_allclose(np.einsum("tii->ti", Ps[ref["idx"][sel]]), ref["Ps_diag"][sel], rtol=1e-10, atol=1e-12)
    npt.assert_allclose(u.sum(), ref["u_checksum"], rtol=1e-12)
