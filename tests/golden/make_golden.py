#!/usr/bin/env python3
"""Generates tests/golden/kalman_known_answers.npz.

JAX is not installed in the build container, so no vector can be produced by running the reference itself.
What the reference's tests pin at this boundary are *known answers*: an explicit covariance-form Kalman filter
and an RTS smoother evaluated on np.random.seed-ed inputs (aux_samplers/_primitives/test_kalman/common.py:5-79,
test_filtering.py:20-107, test_sampling.py:23-127).  This script re-draws those inputs (the legacy MT19937 stream
is stable across NumPy versions) and stores inputs + answers computed by the independent textbook restatements in
oracle/kalman_np.py (`explicit_filter`, `explicit_smoother`).  Both the oracle and the HIP path are then tested
against the stored answers (tests/test_golden.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kalman_np as K  # noqa: E402
from tests.helpers import ref_lgssm_inputs, ref_batched_inputs  # noqa: E402

out = {}
cases = []
# test_filtering.py::test_vs_explicit parametrisation (+ the nan_index=False variant)
for seed in (0, 1234):
    for T in (5, 7):
        for dx in (1, 2):
            for dy in (1, 3):
                for nan in (True, False):
                    name = f"filt_s{seed}_T{T}_dx{dx}_dy{dy}_nan{int(nan)}"
                    ys, lg = ref_lgssm_inputs(seed, T, dx, dy, nan)
                    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
                    ms, Ps, ell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
                    cases.append(name)
                    for k, v in zip(("ys", "m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs", "ms", "Ps", "ell"),
                                    (ys, *lg, ms, Ps, np.float64(ell))):
                        out[f"{name}/{k}"] = v
# test_filtering.py::test_batched_model (dense block-diagonal answer)
for seed in (0, 1234):
    for T in (3, 5):
        for dx in (1, 2):
            for dy in (1, 3):
                name = f"batch_s{seed}_T{T}_dx{dx}_dy{dy}"
                (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, 3)
                m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
                ms, Ps, ell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
                cases.append(name)
                out[f"{name}/bys"] = bys
                for k, v in zip(("m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs"), blg):
                    out[f"{name}/b{k}"] = v
                out[f"{name}/ms"], out[f"{name}/Ps"], out[f"{name}/ell"] = ms, Ps, np.float64(ell)
# test_sampling.py::test_parallel_vs_sequential: smoother moments (exact form of the 500k-sample estimate)
for seed in (42, 666):
    for T in (3, 5):
        for dx in (1, 2):
            for dy in (1, 3):
                name = f"smooth_s{seed}_T{T}_dx{dx}_dy{dy}"
                ys, lg = ref_lgssm_inputs(seed, T, dx, dy)
                m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
                ms, Ps, ell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
                sm, sP = K.explicit_smoother(ms, Ps, Fs, Qs, bs)
                cases.append(name)
                for k, v in zip(("ys", "m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs", "ms", "Ps", "sm", "sP"),
                                (ys, *lg, ms, Ps, sm, sP)):
                    out[f"{name}/{k}"] = v
out["cases"] = np.array(cases)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kalman_known_answers.npz")
np.savez_compressed(path, **out)
print(path, len(cases), "cases", os.path.getsize(path), "bytes")
