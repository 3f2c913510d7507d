"""Counter-based PRNG keys for the samplers (host side).

Keys are (2,) uint32 arrays, split with Threefry-2x32-20 exactly as csrc/rng.h does on the device; the bulk
normal / uniform draws themselves are generated on the GPU (auxssm_rng_normal / auxssm_rng_uniform).
This is NOT bit-compatible with jax.random (JAX is unavailable offline, so that could not be pinned); the parity
contract of this package is on explicit noise arrays, which every kernel also accepts.
"""
import numpy as np

_ROT = (13, 15, 26, 6, 17, 29, 16, 24)
_M = np.uint64(0xFFFFFFFF)


def threefry2x32(k0, k1, x0, x1):
    """Vectorised Threefry-2x32-20 block function; all arguments uint32 arrays/scalars."""
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
        x1 = x1 ^ x0
        if r % 4 == 3:
            j = r // 4 + 1
            x0 = (x0 + ks[j % 3]) & _M
            x1 = (x1 + ks[(j + 1) % 3] + np.uint64(j)) & _M
    return x0.astype(np.uint32), x1.astype(np.uint32)


def PRNGKey(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def as_key(key):
    if isinstance(key, (int, np.integer)):
        return PRNGKey(key)
    key = np.asarray(key)
    if key.shape != (2,):
        raise ValueError(f"a key is an int seed or a (2,) uint32 array, got shape {key.shape}")
    return key.astype(np.uint32)


def split(key, num=2):
    """num child keys: child i = threefry(key, counter=(i, 0xFFFFFFFF)) (a counter no fill kernel ever uses)."""
    key = as_key(key)
    i = np.arange(num, dtype=np.uint32)
    a, b = threefry2x32(key[0], key[1], i, np.full(num, 0xFFFFFFFF, np.uint32))
    return np.stack([a, b], axis=1)


def uniform_scalar(key):
    """One U[0,1) double from a key (used for the MH accept draw on the host path)."""
    key = as_key(key)
    a, _ = threefry2x32(key[0], key[1], np.uint32(0), np.uint32(0xFFFFFFFE))
    return float(a) * 2.3283064365386963e-10


def normal(key, shape=(), dtype=np.float64, handle=None):
    """N(0, 1) draws of `shape`, generated on the device (auxssm_rng_normal, stream 0) and returned as a NumPy array."""
    from . import _lib
    handle = handle or _lib.default_handle()
    key = as_key(key)
    shape = tuple(np.atleast_1d(shape).astype(int)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    out = handle.rng_normal((int(key[0]), int(key[1])), 0, (n,), dtype).to_host()
    return out.reshape(shape) if shape else out[0]


def uniform(key, shape=(), dtype=np.float64, handle=None):
    """U[0, 1) draws of `shape`, generated on the device (auxssm_rng_uniform, stream 0) and returned as a NumPy array."""
    from . import _lib
    handle = handle or _lib.default_handle()
    key = as_key(key)
    shape = tuple(np.atleast_1d(shape).astype(int)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    out = handle.rng_uniform((int(key[0]), int(key[1])), 0, (n,), dtype).to_host()
    return out.reshape(shape) if shape else out[0]
