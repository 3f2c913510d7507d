"""Counter-based PRNG keys for the samplers (host side).

Keys are (2,) uint32 arrays, split with Threefry-2x32-20 exactly as csrc/rng.h does on the device; the bulk
normal / uniform draws themselves are generated on the GPU (auxssm_rng_normal / auxssm_rng_uniform).
The default streams are this package's own (csrc/rng.h); the parity contract is on explicit noise arrays, which every kernel also accepts.

jax.random compatibility (round 4): `jax_split / jax_uniform / jax_normal` reproduce jax.random's threefry2x32 implementation in its non-partitionable layout (JAX's
default up to 0.4.x, what the reference ran on) -- the draws on the device (auxssm_rng_jax), the key arithmetic on the host -- pinned by the values JAX's documentation
prints (tests/test_rng.py: split(PRNGKey(0)) = [4146024105 967050713] / [2718843009 1272950319], uniform -> 0.41845703, normal -> -0.20584226, ...).  `set_compat("jax")`
makes the kernels of kalman.get_kernel and the auxiliary particle-Gibbs kernels draw exactly what the reference's code draws from the same key (kalman/generic.py:58-73,
csmc/generic.py:64-67, _primitives/csmc/csmc.py:71-85, :129-138): the explicit-noise sweeps then run on those arrays.
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
    """num child keys: child i = threefry(key, counter=(i, 0xFFFFFFFF)) (a counter no fill kernel ever uses).  set_compat("jax"): jax.random.split."""
    if _COMPAT == "jax":
        return jax_split(key, num)
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
    """N(0, 1) draws of `shape`, generated on the device (auxssm_rng_normal, stream 0) and returned as a NumPy array.  set_compat("jax"): jax.random.normal."""
    if _COMPAT == "jax":
        return jax_normal(key, shape, dtype, handle)
    from . import _lib
    handle = handle or _lib.default_handle()
    key = as_key(key)
    shape = tuple(np.atleast_1d(shape).astype(int)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    out = handle.rng_normal((int(key[0]), int(key[1])), 0, (n,), dtype).to_host()
    return out.reshape(shape) if shape else out[0]


def uniform(key, shape=(), dtype=np.float64, handle=None):
    """U[0, 1) draws of `shape`, generated on the device (auxssm_rng_uniform, stream 0) and returned as a NumPy array.  set_compat("jax"): jax.random.uniform."""
    if _COMPAT == "jax":
        return jax_uniform(key, shape, dtype, handle=handle)
    from . import _lib
    handle = handle or _lib.default_handle()
    key = as_key(key)
    shape = tuple(np.atleast_1d(shape).astype(int)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    out = handle.rng_uniform((int(key[0]), int(key[1])), 0, (n,), dtype).to_host()
    return out.reshape(shape) if shape else out[0]


# ---- jax.random compatibility -------------------------------------------------------------------------------------------------------------------------------
_COMPAT = None


def set_compat(mode):
    """None: this package's own streams (default).  "jax": kernels called with a key draw what the reference's code draws from that key with jax.random (threefry2x32,
    non-partitionable layout).  Returns the previous mode."""
    global _COMPAT
    if mode not in (None, "jax"):
        raise ValueError('mode must be None or "jax"')
    prev, _COMPAT = _COMPAT, mode
    return prev


def compat():
    return _COMPAT


def _jax_threefry_2x32(key, count):
    count = np.asarray(count, np.uint32).ravel()
    odd = count.size % 2
    c = np.concatenate([count, np.zeros(1, np.uint32)]) if odd else count
    h = c.size // 2
    o0, o1 = threefry2x32(key[0], key[1], c[:h], c[h:])
    out = np.concatenate([o0, o1])
    return out[:-1] if odd else out


def jax_split(key, num=2):
    """jax.random.split(key, num) -> (num, 2) uint32; `key` may also be (K, 2): -> (K, num, 2) (a vmap over keys)"""
    key = np.asarray(key, np.uint32)
    if key.ndim == 2:   # the same counters for every key: one vectorised block evaluation (2 num counters: halves (0 .. num) and (num .. 2 num), never odd)
        x0, x1 = np.arange(num, dtype=np.uint32)[None, :], np.arange(num, 2 * num, dtype=np.uint32)[None, :]
        o0, o1 = threefry2x32(key[:, :1], key[:, 1:], x0, x1)
        return np.concatenate([o0, o1], axis=1).reshape(key.shape[0], num, 2)
    return _jax_threefry_2x32(as_key(key), np.arange(2 * num, dtype=np.uint32)).reshape(num, 2)


def _jax_fill(kind, keys, n, dtype, minval, maxval, handle, out=None, key_stride=None, elem_stride=1):
    from . import _lib
    handle = handle or _lib.default_handle()
    keys = np.ascontiguousarray(np.asarray(keys, np.uint32).reshape(-1, 2))
    kd = handle.to_device(keys)
    dtype = np.dtype(dtype)
    if out is None:
        out = handle.empty((keys.shape[0], n), dtype)
        key_stride = n
    _lib.check(handle.lib.auxssm_rng_jax(handle.h, _lib.dtype_code(dtype), kind, keys.shape[0], n, kd.ptr, float(minval), float(maxval), out.ptr, int(key_stride),
                                         int(elem_stride)))
    return out


def jax_uniform(key, shape=(), dtype=np.float32, minval=0.0, maxval=1.0, handle=None):
    """jax.random.uniform(key, shape, dtype, minval, maxval); `key` (2,) -> array of `shape`; keys (K, 2) -> (K,) + shape (jax.vmap over the keys)"""
    key = np.asarray(key, np.uint32)
    shape = tuple(int(v) for v in np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    out = _jax_fill(0, key, n, dtype, minval, maxval, handle).to_host()
    return out.reshape(((key.shape[0],) if key.ndim == 2 else ()) + shape)


def jax_normal(key, shape=(), dtype=np.float32, handle=None):
    """jax.random.normal(key, shape, dtype) (sqrt(2) erfinv of a uniform on (-1, 1)); keys (K, 2) -> (K,) + shape"""
    key = np.asarray(key, np.uint32)
    shape = tuple(int(v) for v in np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    out = _jax_fill(1, key, n, dtype, 0.0, 1.0, handle).to_host()
    return out.reshape(((key.shape[0],) if key.ndim == 2 else ()) + shape)


def jax_kalman_noise(keys, T, d, dtype, handle=None):
    """the draws of the reference's auxiliary Kalman kernel for each key of `keys` (C, 2) (kalman/generic.py:58-61, :73; _primitives/kalman/sampling.py:128):
    auxiliary_key, sampling_key, accept_key = split(key, 3); normal(auxiliary_key, (T, d)); normal(sampling_key, (T, d)); bernoulli(accept_key, alpha) = uniform < alpha.
    -> dict(eps_aux (C, T, d), eps_samp (C, T, d), u_accept (C,)) host arrays"""
    ks = jax_split(np.asarray(keys, np.uint32).reshape(-1, 2), 3)      # (C, 3, 2)
    return dict(eps_aux=jax_normal(ks[:, 0], (T, d), dtype, handle), eps_samp=jax_normal(ks[:, 1], (T, d), dtype, handle),
                u_accept=jax_uniform(ks[:, 2], (), dtype, handle=handle))


def jax_csmc_noise(key, T, N, d, dtype, backward, handle=None, auxiliary=True):
    """the draws of the reference's auxiliary particle-Gibbs kernel with independent proposals for ONE key (csmc/generic.py:64-67; _primitives/csmc/csmc.py:53, :71-85,
    :111, :129-138; csmc/independent.py:155-158, :194-198; resamplings.py:35: `choice` draws one uniform per index):
        auxiliary_key, key = split(key); eps_aux = normal(auxiliary_key, (T, d)); key_fwd, key_bwd = split(key); keys = split(key_fwd, T);
        eps_prop[0] = normal(keys[0], (N, d)); t >= 1: resampling_key, sampling_key = split(keys[t]); u_res[t - 1] = uniform(resampling_key, (N,));
        eps_prop[t] = normal(sampling_key, (N, d)); backward sampling: kb = split(key_bwd, T), B_{T-1} from uniform(kb[0], ()), B_t from kb[T - 1 - t];
        ancestor tracing: B_{T-1} from uniform(key_bwd, ()).
    auxiliary=False: the plain cSMC kernel (_primitives/csmc/csmc.py:52-59: no auxiliary split) for models whose M0.sample / Mt.sample draw ONE normal(key, (N, d))
    per call, as every model of the reference does (test_csmc/common.py:26-28, :42-43; examples/stochastic_volatility/auxiliary_csmc.py:23, :37).
    -> dict(eps_aux (T, d) [auxiliary], eps_prop (T, N, d), u_res (T - 1, N), u_bwd (T,)) host arrays"""
    aux_key, k = jax_split(key, 2) if auxiliary else (None, as_key(key))
    k_fwd, k_bwd = jax_split(k, 2)
    keys = jax_split(k_fwd, T)
    eps_prop = np.empty((T, N, d), dtype)
    eps_prop[0] = jax_normal(keys[0], (N, d), dtype, handle)
    u_res = np.zeros((max(T - 1, 0), N), dtype)
    if T > 1:
        rs = jax_split(keys[1:], 2)                                      # (T - 1, 2, 2)
        u_res[:] = jax_uniform(rs[:, 0], (N,), dtype, handle=handle)
        eps_prop[1:] = jax_normal(rs[:, 1], (N, d), dtype, handle)
    u_bwd = np.zeros((T,), dtype)
    if backward:
        kb = jax_split(k_bwd, T)
        u_bwd[:] = jax_uniform(kb[::-1], (), dtype, handle=handle)
    else:
        u_bwd[T - 1] = jax_uniform(k_bwd, (), dtype, handle=handle)
    out = dict(eps_prop=eps_prop, u_res=u_res, u_bwd=u_bwd)
    if auxiliary:
        out["eps_aux"] = jax_normal(aux_key, (T, d), dtype, handle)
    return out


def jax_pit_noise(key, T, N, d, dtype, handle=None):
    """the draws of the reference's PARALLEL-IN-TIME auxiliary kernel for one key (csmc/independent.py:105-110; _primitives/csmc/pit/csmc.py:70-75; pit/operator.py:76-81):
        auxiliary_key, key = split(key); eps_aux = normal(auxiliary_key, (T, d)); sampling_key, resampling_key = split(key);
        eps_prop[t] = normal(split(sampling_key, T)[t], (N, d)); the stitch at the boundary (t - 1 | t) resamples with split(resampling_key, T)[t]:
        N uniforms (conditional multinomial over the N^2 pairs), except the LAST stitch of the tree -- boundary 2^(ceil(log2 T) - 1) -- which is one `choice` of shape ().
    -> dict(eps_aux (T, d), eps_prop (T, N, d), u_res (T, N)) host arrays"""
    aux_key, k = jax_split(key, 2)
    sk, rk = jax_split(k, 2)
    sks, rks = jax_split(sk, T), jax_split(rk, T)
    u_res = jax_uniform(rks, (N,), dtype, handle=handle)
    if T > 1:
        K = int(np.ceil(np.log2(T)))
        root = 1 << (K - 1)
        u_res[root, 0] = jax_uniform(rks[root], (), dtype, handle=handle)
    return dict(eps_aux=jax_normal(aux_key, (T, d), dtype, handle), eps_prop=jax_normal(sks, (N, d), dtype, handle), u_res=u_res)
