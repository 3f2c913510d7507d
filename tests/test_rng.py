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


def test_rng_tables_are_current_and_symmetric():
    """csrc/rng_tables.h is what tools/gen_rng_tables.py writes (60-digit decimal evaluation, correctly rounded), the turn table maps onto itself under a
    quarter turn EXACTLY (normals_cm's lane-pair split relies on it), the log table's end points are exact and every entry is within half an ulp of libm's value + 1 ulp."""
    import math
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        import gen_rng_tables as G
    finally:
        sys.path.remove(tools)
    hdr = os.path.join(os.path.dirname(tools), "aux_ssm_samplers_amd", "csrc", "rng_tables.h")
    assert open(hdr).read() == G.render()
    logt, turn = G.tables()
    assert logt[0] == (2.0, -2.0 * math.log(0.5)) and logt[256] == (1.0, 0.0)
    for j in range(257):
        F = (256 + j) / 512
        assert abs(logt[j][0] - 1 / F) <= 2.3e-16 * (1 / F) and abs(logt[j][1] + 2 * math.log(F)) <= 4.5e-16
    for j in range(256):
        c, s = turn[j]
        c2, s2 = turn[(j + 64) % 256]
        assert c2 == -s and s2 == c
        assert abs(c - math.cos(2 * math.pi * j / 256)) < 1e-15 and abs(s - math.sin(2 * math.pi * j / 256)) < 1e-15
        assert abs(c * c + s * s - 1) < 3e-16


@pytest.mark.gpu
def test_fp64_normals_edge_words_and_the_lane_pair_split():
    """The table-driven fp64 transform at the words where its branches meet (u1 next to 0 and 1: the largest radius and the radii whose table terms cancel exactly;
    u2 on sector boundaries and quarter turns) against an 80-bit libm evaluation, and the chain-minor in-kernel draws (lane pairs, the odd lane a quarter turn back)
    bit-identical to the fill kernel -- the latter through the keyed fused sweep's tests (tests/test_gpu_device_delta.py); here the fill itself."""
    from aux_ssm_samplers_amd import _lib, random as R
    from oracle import rng_np as O
    h = _lib.default_handle()
    key = R.PRNGKey(99)
    n = 1 << 20
    z = h.rng_normal(key, 3, (n,), np.float64).to_host()
    b0, b1 = O._bits(key, 3, n // 2)
    u1 = (b0.astype(np.longdouble) + np.longdouble(0.5)) / np.longdouble(2) ** 32
    u2 = (b1.astype(np.longdouble) + np.longdouble(0.5)) / np.longdouble(2) ** 32
    r = np.sqrt(-2 * np.log(u1))
    tau = 2 * np.longdouble("3.14159265358979323846264338327950288")
    ref = np.stack([r * np.cos(tau * u2), r * np.sin(tau * u2)], axis=1).reshape(-1)
    err = np.abs(z.astype(np.longdouble) - ref)
    assert float(err.max()) < 1e-14, float(err.max())  # |z| <= 6.8: a few ulp
    # the extreme words of this sample really exercise the ends of the tables
    assert b0.min() < 2 ** 16 and b0.max() > 2 ** 32 - 2 ** 16 and ((b1 + np.uint32(0x800000)) >> np.uint32(24)).min() == 0


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
