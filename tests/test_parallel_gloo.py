"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising chain sharding, per-chain key folding and the
final chain-gather exactly as bench.py uses them (there the backend is nccl = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_chains_partition():
    from aux_ssm_samplers_amd.parallel import shard_chains
    for total in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            parts = [shard_chains(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_chains(4, 2, 2)


def test_chain_keys_distinct():
    from aux_ssm_samplers_amd.parallel import chain_key
    keys = {tuple(chain_key(5, c)) for c in range(1000)}
    assert len(keys) == 1000


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from aux_ssm_samplers_amd.parallel import shard_chains, gather_chains, chain_key
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_chains(total, rank, world)
    # stand-in for the per-chain sweep result: a deterministic function of the chain's folded key
    local = np.stack([np.concatenate([[c], chain_key(123, c).astype(np.float64)]) for c in range(lo, hi)]) if hi > lo \
        else np.zeros((0, 3))
    got = gather_chains(local, total, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, None if got is None else got.tolist()))


@pytest.mark.parametrize("world,total", [(2, 64), (2, 5), (3, 7), (8, 64)])  # (8, 64): the C4 split, 64 chains -> 8 per rank
def test_gather_chains_gloo(world, total):
    import torch.multiprocessing as mp
    from aux_ssm_samplers_amd.parallel import chain_key
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.concatenate([[c], chain_key(123, c).astype(np.float64)]) for c in range(total)])
    np.testing.assert_array_equal(np.array(res[0]), want)
    assert all(res[r] is None for r in range(1, world))
