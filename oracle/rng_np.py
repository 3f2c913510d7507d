"""ORACLE -- TEST INFRASTRUCTURE ONLY.  NumPy restatement of the device RNG fills (csrc/rng.h, api.hip::k_rng_fill):
uniform[i] = f(block i), normal[i] = Box-Muller branch (i & 1) of block i >> 1, block b = threefry2x32(key,
counter = (lo32(b), stream ^ (hi32(b) << 16))).
Threefry-2x32-20 itself is pinned by the Random123 known-answer vectors (tests/test_rng.py).  The normal transform uses
libm log/cos on both sides, so device-vs-oracle agreement for normals is to rounding (a few ulp), not bitwise; uniforms are
bit-exact.  jax.random bit-compatibility is NOT claimed (unverifiable offline)."""
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
    b0, _ = _bits(key, stream, n)
    if np.dtype(dtype) == np.float32:
        return ((b0 >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-8)).astype(np.float32)
    return b0.astype(np.float64) * 2.3283064365386963e-10


def normal(key, stream, n, dtype):
    nb = (n + 1) // 2
    b0, b1 = _bits(key, stream, nb)
    if np.dtype(dtype) == np.float32:
        u1 = ((b0 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(5.9604644775390625e-8)
        u2 = ((b1 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(5.9604644775390625e-8)
        r = np.sqrt(np.float32(-2.0) * np.log(u1))
        a = np.float32(6.283185307179586) * u2
        z = np.stack([r * np.cos(a), r * np.sin(a)], axis=1).astype(np.float32)
    else:
        u1 = (b0.astype(np.float64) + 0.5) * 2.3283064365386963e-10
        u2 = (b1.astype(np.float64) + 0.5) * 2.3283064365386963e-10
        r = np.sqrt(-2.0 * np.log(u1))
        a = 6.283185307179586476925286766559 * u2
        z = np.stack([r * np.cos(a), r * np.sin(a)], axis=1)
    return z.reshape(-1)[:n]
