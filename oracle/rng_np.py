"""ORACLE -- TEST INFRASTRUCTURE ONLY.  NumPy restatement of the device RNG fills (csrc/rng.h, api.hip::k_rng_fill):
uniform[i] = word (i & 1) of block i >> 1, normal[i] = Box-Muller branch (i & 1) of block i >> 1, block b = threefry2x32(key,
counter = (lo32(b), stream ^ (hi32(b) << 16))).
Threefry-2x32-20 itself is pinned by the Random123 known-answer vectors (tests/test_rng.py).  Contract of the normal transform (csrc/rng.h):
uniforms are bit-exact; normals agree to a tolerance -- this restatement uses libm's log / sqrt and the quadrant-split sincos below without fma, the device
a table-driven log and rotation (csrc/rng.h::bm_fp64, tables csrc/rng_tables.h: ~2e-15 absolute of the exact transform) with explicit fma in
fp64 (<= 1e-12 relative / 1e-13 absolute asserted) and the hardware log2 / sqrt in fp32 (<= 2e-5 relative / 2e-6 absolute), so fp32 device normals are reproducible on the
device only and bit-exact checks that use keyed noise draw it there first.

The `jax_*` functions at the end restate jax.random's own bit stream (the threefry2x32 implementation, non-partitionable layout -- JAX's default up to 0.4.x, what
the reference ran on): `split`, `uniform`, `normal`.  JAX is not installable here, so they are pinned by the known answers JAX's documentation prints for PRNGKey(0)
(tests/test_rng.py): split -> [4146024105 967050713] / [2718843009 1272950319], uniform(key) -> 0.41845703, normal(key, (1,)) -> -0.20584226,
normal(subkey, (1,)) -> -1.2515389.  float64 draws follow the same code path of jax/_src/prng.py (two words of one block per value); no float64 known answer is
available offline, and the float64 erfinv is SciPy's, not XLA's rational approximation: agreement with JAX there is to rounding, by construction, and stated so."""
import numpy as np

_ROT = (13, 15, 26, 6, 17, 29, 16, 24)
_M = np.uint64(0xFFFFFFFF)


def threefry2x32(k0, k1, x0, x1):
    k0, k1 = np.uint64(k0), np.uint64(k1)
    x0 = np.asarray(x0, np.uint64).copy()
    x1 = np.asarray(x1, np.uint64).copy()
    ks = (k0, k1, np.uint64(0x1BD11BDA) ^ k0 ^ k1)
    x0 = (x0 + ks[0]) & _M
    x1 = (x1 + ks[1]) & _M
    for r in range(20):
        x0 = (x0 + x1) & _M
        rot = np.uint64(_ROT[r % 8])
        x1 = ((x1 << rot) | (x1 >> (np.uint64(32) - rot))) & _M
        x1 ^= x0
        if r % 4 == 3:
            j = r // 4 + 1
            x0 = (x0 + ks[j % 3]) & _M
            x1 = (x1 + ks[(j + 1) % 3] + np.uint64(j)) & _M
    return x0.astype(np.uint32), x1.astype(np.uint32)


def _bits(key, stream, n):
    i = np.arange(n, dtype=np.uint64)
    x0 = (i & _M).astype(np.uint32)
    x1 = (np.uint64(stream) ^ ((i >> np.uint64(32)) << np.uint64(16))).astype(np.uint32)
    return threefry2x32(key[0], key[1], x0, x1)


def uniform(key, stream, n, dtype):
    b0, b1 = _bits(key, stream, (n + 1) // 2)
    b = np.stack([b0, b1], axis=1).reshape(-1)[:n]
    if np.dtype(dtype) == np.float32:
        return ((b >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-8)).astype(np.float32)
    return b.astype(np.float64) * 2.3283064365386963e-10


_S = {np.float32: [-1.6666667163e-01, 8.3333337680e-03, -1.9841270114e-04, 2.7557314297e-06],
      np.float64: [-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                   2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10]}
_C = {np.float32: [4.1666667908e-02, -1.3888889225e-03, 2.4801587642e-05, -2.7557314297e-07],
      np.float64: [4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                   -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11]}


def sincos_2pi(u):
    """cos(2 pi u), sin(2 pi u) the way csrc/rng.h::sincos_2pi computes them (quadrant split of u, fdlibm kernels on
    theta = 2 pi f).  NumPy has no fma, so the last bit can differ from the device; tests compare with a tolerance."""
    R = u.dtype.type
    q = np.rint(R(4) * u)
    f = u - R(0.25) * q  # exact
    th = R(6.283185307179586476925286766559) * f
    z = th * th
    ps = np.full_like(u, R(_S[R][-1]))
    pc = np.full_like(u, R(_C[R][-1]))
    for k in _S[R][-2::-1]:
        ps = ps * z + R(k)
    for k in _C[R][-2::-1]:
        pc = pc * z + R(k)
    sn = th * z * ps + th
    cs = z * z * pc + (R(1) - R(0.5) * z)
    qi = q.astype(np.int64) & 3
    a = np.where(qi & 1, sn, cs)
    b = np.where(qi & 1, cs, sn)
    c = np.where((qi == 1) | (qi == 2), -a, a)
    s = np.where(qi >= 2, -b, b)
    return c.astype(R), s.astype(R)


def normal(key, stream, n, dtype):
    nb = (n + 1) // 2
    b0, b1 = _bits(key, stream, nb)
    if np.dtype(dtype) == np.float32:
        u1 = ((b0 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(5.9604644775390625e-8)
        u2 = ((b1 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(5.9604644775390625e-8)
        r = np.sqrt(np.float32(-2.0) * np.log(u1))
    else:
        u1 = (b0.astype(np.float64) + 0.5) * 2.3283064365386963e-10
        u2 = (b1.astype(np.float64) + 0.5) * 2.3283064365386963e-10
        r = np.sqrt(-2.0 * np.log(u1))
    c, s = sincos_2pi(u2)
    z = np.stack([r * c, r * s], axis=1).astype(dtype)
    return z.reshape(-1)[:n]


# ---- jax.random (threefry2x32, legacy layout): jax/_src/prng.py::threefry_2x32, threefry_split, threefry_random_bits; jax/_src/random.py::_uniform, _normal_real ----
def jax_threefry_2x32(key, count):
    """threefry_2x32(keypair, count): the counters split in two halves (padded with one 0 when odd), one block per pair, outputs concatenated"""
    count = np.asarray(count, np.uint32).ravel()
    odd = count.size % 2
    c = np.concatenate([count, np.zeros(1, np.uint32)]) if odd else count
    h = c.size // 2
    o0, o1 = threefry2x32(np.uint32(key[0]), np.uint32(key[1]), c[:h], c[h:])
    out = np.concatenate([o0, o1])
    return out[:-1] if odd else out


def jax_split(key, num=2):
    """jax.random.split(key, num) -> (num, 2) uint32"""
    return jax_threefry_2x32(key, np.arange(2 * num, dtype=np.uint32)).reshape(num, 2)


def jax_bits(key, n, width=32):
    """threefry_random_bits(key, width, (n,))"""
    if width == 32:
        return jax_threefry_2x32(key, np.arange(n, dtype=np.uint32))
    b = jax_threefry_2x32(key, np.arange(2 * n, dtype=np.uint32))
    return (b[:n].astype(np.uint64) << np.uint64(32)) | b[n:].astype(np.uint64)


def jax_uniform(key, n, dtype, minval=0.0, maxval=1.0):
    """jax.random.uniform(key, (n,), dtype, minval, maxval): mantissa bits | exponent of 1, minus 1, scaled, clamped below"""
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        f = ((jax_bits(key, n, 32) >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    else:
        f = ((jax_bits(key, n, 64) >> np.uint64(12)) | np.uint64(0x3FF0000000000000)).view(np.float64) - 1.0
    lo, hi = dtype.type(minval), dtype.type(maxval)
    return np.maximum(lo, (f * (hi - lo) + lo).astype(dtype))


def _erfinv_f32(x):
    """XLA's ErfInvF32 (M. Giles, "Approximating the erfinv function"), evaluated in float32"""
    x = np.asarray(x, np.float32)
    f = np.float32
    w = -np.log1p(-x * x).astype(np.float32)
    lt = w < f(5.0)
    wa = np.where(lt, w - f(2.5), np.sqrt(np.maximum(w, f(0))).astype(np.float32) - f(3.0)).astype(np.float32)
    ca = [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941]
    cb = [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682]
    p = np.where(lt, f(ca[0]), f(cb[0])).astype(np.float32)
    for a, b in zip(ca[1:], cb[1:]):
        p = (np.where(lt, f(a), f(b)) + p * wa).astype(np.float32)
    return (p * x).astype(np.float32)


def jax_normal(key, n, dtype):
    """jax.random.normal(key, (n,), dtype) = sqrt(2) erfinv(uniform(key, (n,), dtype, nextafter(-1, 0), 1))"""
    dtype = np.dtype(dtype)
    lo = np.nextafter(dtype.type(-1), dtype.type(0))
    u = jax_uniform(key, n, dtype, lo, 1.0)
    if dtype == np.float32:
        return (np.float32(np.sqrt(2.0)) * _erfinv_f32(u)).astype(np.float32)
    from scipy.special import erfinv
    return np.sqrt(2.0) * erfinv(u)
