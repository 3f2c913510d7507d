"""Known answers for BASELINE config C5 at its BENCHMARKED horizon (d = p = 64, T = 8192): the fp64 SEQUENTIAL
filter and pathwise sampler of oracle/kalman_np.py (filtering.py:66-79, sampling.py:34-39 restated) on the model of tests/helpers.py::c5_model,
sampled at 32 time points -- the benchmarked fp32 parallel scan (8191 combines of unpivoted blocked eliminations) is compared against them on the GPU
(tests/test_gpu_wide.py::test_C5_benchmarked_horizon_vs_fp64_sequential_fixture).

    python tests/golden/make_c5_fixture.py            writes tests/golden/c5_T8192_known_answers.npz   (about 15 s)
    python tests/golden/make_c5_fixture.py --check    regenerates and compares with the committed file

Stored: the time points, ms (32, 64), diag Ps (32, 64), four full covariances, ell, the sampled trajectory at the time points, and checksums of
the inputs (u, eps) so that a drift of the synthetic model shows up as such."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "c5_T8192_known_answers.npz")


def compute(T=8192, d=64):
    from oracle import kalman_np as K
    from tests.helpers import c5_model
    u, lg, _ = c5_model(T, d)
    ms, Ps, ell = K.filtering(u, lg, False)
    eps = np.random.default_rng(1).standard_normal((T, d))
    xs = K.sampling(eps, ms, Ps, lg, False)
    idx = np.unique(np.linspace(0, T - 1, 32).astype(np.int64))
    full = idx[[0, 10, 21, -1]]
    return dict(idx=idx, ms=ms[idx], Ps_diag=np.einsum("tii->ti", Ps[idx]), full_idx=full, Ps_full=Ps[full], ell=np.float64(ell), xs=xs[idx],
                u_checksum=np.float64(u.sum()), eps_checksum=np.float64(eps.sum()))


if __name__ == "__main__":
    got = compute()
    if "--check" in sys.argv:
        ref = np.load(OUT)
        worst = max(float(np.max(np.abs(got[k] - ref[k]) / (1e-300 + np.maximum(np.abs(ref[k]), 1.0)))) for k in ref.files)
        print(f"max scaled |diff| over {len(ref.files)} arrays: {worst:.3e}")
        sys.exit(0 if worst < 1e-10 else 1)
    np.savez_compressed(OUT, **got)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
