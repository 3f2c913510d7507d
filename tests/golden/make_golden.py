#!/usr/bin/env python3
"""Generates tests/golden/kalman_known_answers.npz FROM THE REFERENCE'S OWN TEST ORACLE.

JAX is not installed in the build container, so the reference's filter / sampler cannot run here.  What the
reference's tests pin at this boundary are *known answers*: an explicit covariance-form Kalman filter and an RTS
smoother evaluated on np.random.seed-ed inputs (aux_samplers/_primitives/test_kalman/common.py:5-79,
test_filtering.py:20-107, test_sampling.py:23-127).  That file imports only numpy and scipy, so this script loads
it BY PATH from /root/reference (build container only; it never travels to the GPU box) and stores

    ms, Ps, ell   = reference `explicit_kalman_filter`      (common.py:27-79)
    sm, sP        = reference `explicit_kalman_smoothing`   (common.py:5-24)

on the inputs the reference's tests draw (the legacy MT19937 stream is stable across NumPy versions; the drawing
order is restated in tests/helpers.py).  The repo's own textbook restatements (oracle/kalman_np.py `explicit_filter`,
`explicit_smoother`) are evaluated beside them, stored as a second column (`*_own`) and asserted to agree to 1e-10,
so the fixture is pinned by the reference's file, not by the builder's word.  Both the oracle and the HIP path are
then tested against the stored reference answers (tests/test_golden.py).

    python tests/golden/make_golden.py            # regenerate
    python tests/golden/make_golden.py --check    # regenerate in memory and compare with the committed file
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kalman_np as K  # noqa: E402
from tests.helpers import ref_lgssm_inputs, ref_batched_inputs  # noqa: E402

REF_COMMON = "/root/reference/aux_samplers/_primitives/test_kalman/common.py"
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kalman_known_answers.npz")
LG = ("m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs")


def load_reference_oracle():
    """The reference's NumPy/SciPy test oracle, loaded by file path (no package import: the package's __init__ needs JAX)."""
    spec = importlib.util.spec_from_file_location("_ref_test_kalman_common", REF_COMMON)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _agree(name, ref, own):
    for r, o, what in zip(ref, own, ("ms", "Ps", "ell")):
        if not np.allclose(r, o, rtol=1e-10, atol=1e-10, equal_nan=True):
            raise AssertionError(f"{name}: reference oracle and own restatement disagree on {what}: {np.max(np.abs(r - o))}")


def generate():
    R = load_reference_oracle()
    out, cases = {}, []

    def ref_filter(ys, lg):
        m0, P0, Fs, Qs, bs, Hs, Rs, cs = (np.array(a, copy=True) for a in lg)   # the reference oracle rebinds its inputs
        ms, Ps, ell = R.explicit_kalman_filter(np.array(ys, copy=True), m0, P0, Hs, Rs, cs, Fs, Qs, bs)
        return ms, Ps, np.float64(ell)

    def own_filter(ys, lg):
        m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
        ms, Ps, ell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
        return ms, Ps, np.float64(ell)

    # test_filtering.py::test_vs_explicit parametrisation (:20-55; + the nan_index=False variant)
    for seed in (0, 1234):
        for T in (5, 7):
            for dx in (1, 2):
                for dy in (1, 3):
                    for nan in (True, False):
                        name = f"filt_s{seed}_T{T}_dx{dx}_dy{dy}_nan{int(nan)}"
                        ys, lg = ref_lgssm_inputs(seed, T, dx, dy, nan)
                        ref, own = ref_filter(ys, lg), own_filter(ys, lg)
                        _agree(name, ref, own)
                        cases.append(name)
                        out[f"{name}/ys"] = ys
                        for k, v in zip(LG, lg):
                            out[f"{name}/{k}"] = v
                        for k, v, w in zip(("ms", "Ps", "ell"), ref, own):
                            out[f"{name}/{k}"], out[f"{name}/{k}_own"] = v, w
    # test_filtering.py::test_batched_model (:58-107; dense block-diagonal answer)
    for seed in (0, 1234):
        for T in (3, 5):
            for dx in (1, 2):
                for dy in (1, 3):
                    name = f"batch_s{seed}_T{T}_dx{dx}_dy{dy}"
                    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, 3)
                    ref, own = ref_filter(ys, lg), own_filter(ys, lg)
                    _agree(name, ref, own)
                    cases.append(name)
                    out[f"{name}/bys"] = bys
                    for k, v in zip(LG, blg):
                        out[f"{name}/b{k}"] = v
                    for k, v, w in zip(("ms", "Ps", "ell"), ref, own):
                        out[f"{name}/{k}"], out[f"{name}/{k}_own"] = v, w
    # test_sampling.py::test_parallel_vs_sequential (:23-68): smoother moments (exact form of the 500k-sample estimate)
    for seed in (42, 666):
        for T in (3, 5):
            for dx in (1, 2):
                for dy in (1, 3):
                    name = f"smooth_s{seed}_T{T}_dx{dx}_dy{dy}"
                    ys, lg = ref_lgssm_inputs(seed, T, dx, dy)
                    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
                    ms, Ps, ell = ref_filter(ys, lg)
                    _agree(name, (ms, Ps, ell), own_filter(ys, lg))
                    sm, sP = R.explicit_kalman_smoothing(ms.copy(), Ps.copy(), Fs, Qs, bs)
                    sm_o, sP_o = K.explicit_smoother(ms, Ps, Fs, Qs, bs)
                    _agree(name, (sm, sP), (sm_o, sP_o))
                    cases.append(name)
                    out[f"{name}/ys"] = ys
                    for k, v in zip(LG, lg):
                        out[f"{name}/{k}"] = v
                    for k, v in zip(("ms", "Ps", "sm", "sP", "sm_own", "sP_own"), (ms, Ps, sm, sP, sm_o, sP_o)):
                        out[f"{name}/{k}"] = v
    out["cases"] = np.array(cases)
    out["provenance"] = np.array("answers: /root/reference/aux_samplers/_primitives/test_kalman/common.py "
                                 "(explicit_kalman_filter :27-79, explicit_kalman_smoothing :5-24) loaded by path; "
                                 "*_own: oracle/kalman_np.py explicit_filter / explicit_smoother")
    return out


def main():
    out = generate()
    if "--check" in sys.argv:
        G = np.load(PATH)
        assert set(G.files) == set(out), sorted(set(G.files) ^ set(out))
        worst = 0.0
        for k, v in out.items():
            if v.dtype.kind in "US":
                assert np.array_equal(G[k], v), k
            else:
                assert np.allclose(G[k], v, rtol=1e-12, atol=1e-12, equal_nan=True), k
                if np.isfinite(v).any():
                    worst = max(worst, float(np.nanmax(np.abs(G[k] - v))))
        print(f"committed fixture == regenerated from the reference's file (max |diff| {worst:.1e}, {len(out['cases'])} cases)")
        return
    np.savez_compressed(PATH, **out)
    print(PATH, len(out["cases"]), "cases", os.path.getsize(PATH), "bytes")


if __name__ == "__main__":
    main()
