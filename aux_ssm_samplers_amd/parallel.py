"""Multi-GPU layout of the samplers: independent MCMC chains never interact (the reference's only multi-chain code is
jax.vmap over chain keys, examples/rare_event/experiment.py:189-196), so chains are sharded over ranks -- one process
per GPU -- with NO collective on the data path.  The only exchange is the final chain-gather of small per-chain
results (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests)."""
import numpy as np


def shard_chains(total_chains, rank, world):
    """Contiguous block partition of chain ids: rank r owns [lo, hi).  Sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(total_chains), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chain_key(key, chain_id):
    """Decorrelated per-chain key: fold the global chain id into the Threefry key."""
    from . import random as R
    key = R.as_key(key)
    a, b = R.threefry2x32(key[0], key[1], np.uint32(chain_id & 0xFFFFFFFF), np.uint32(0xC4A10000))
    return np.array([a, b], np.uint32)


def gather_chains(local, total_chains, dist=None, dst=0, device=None):
    """Gather per-chain rows (local: (n_local, ...) ndarray) from every rank to `dst` in global chain order.
    `dist` is torch.distributed (initialised) or None for a single process.  Returns the (total_chains, ...) array on
    `dst`, None elsewhere.  One padded all-gather: the payload is KBs-MBs, latency-bound, so a single collective over
    the xGMI mesh (not a ring of many small ones)."""
    local = np.ascontiguousarray(local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_chains(total_chains, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = np.zeros((nmax,) + local.shape[1:], local.dtype)
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    if rank != dst:
        return None
    return np.concatenate([o.cpu().numpy()[:hi - lo] for o, (lo, hi) in zip(out, sizes)], axis=0)
