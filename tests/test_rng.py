"""Threefry-2x32-20: Random123 known-answer vectors for both host implementations (product
aux_ssm_samplers_amd/random.py, oracle/rng_np.py); device fills vs the oracle on the GPU."""
import numpy as np
import numpy.testing as npt
import pytest

KATS = [((0, 0), (0, 0), (0x6b200159, 0x99ba4efe)),
        ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
        ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_threefry_known_answers(which):
    if which == "product":
        from aux_ssm_samplers_amd.random import threefry2x32
    else:
        from oracle.rng_np import threefry2x32
    for key, ctr, want in KATS:
        a, b = threefry2x32(np.uint32(key[0]), np.uint32(key[1]), np.uint32(ctr[0]), np.uint32(ctr[1]))
        assert (int(a), int(b)) == want


def test_split_is_deterministic_and_distinct():
    from aux_ssm_samplers_amd import random as R
    k = R.split(R.PRNGKey(7), 1000)
    assert len({tuple(x) for x in k}) == 1000
    npt.assert_array_equal(k, R.split(7, 1000))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_fill_vs_oracle(dtype):
    from aux_ssm_samplers_amd import _lib, random as R
    from oracle import rng_np as O
    h = _lib.default_handle()
    key = R.PRNGKey(2**40 + 12345)
    n = 100_000
    npt.assert_array_equal(h.rng_uniform(key, 5, (n,), dtype).to_host(), O.uniform(key, 5, n, dtype))
    z = h.rng_normal(key, 9, (n,), dtype).to_host()
    npt.assert_allclose(z, O.normal(key, 9, n, dtype), rtol=2e-5 if dtype == np.float32 else 1e-12, atol=4e-6 if dtype == np.float32 else 1e-13)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
