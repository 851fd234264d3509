"""CPU suite: the N>1 path (row shards, replicated queries, one all-gather of candidates, merge) with
world_size 2 over gloo.  The local shard search is the oracle here (no GPU in this container); on the GPU box
the same ShardedIndex wraps the HIP FlatIPIndex (test_sharded_gpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


class OracleShard:
    """Test double with the FlatIPIndex surface, backed by oracle/search_ref.py."""

    def __init__(self, d):
        self.d = d
        self.rows = np.zeros((0, d), np.float32)

    @property
    def ntotal(self):
        return len(self.rows)

    def add(self, x):
        self.rows = np.concatenate([self.rows, np.asarray(x, np.float32)])

    def search_device(self, q, k, normalize=False, id_base=0):
        from oracle import search_ref as S
        q = np.asarray(q, np.float32)
        if normalize:
            q = S.normalize_rows_core(q).astype(np.float32)
        D, I = S.flat_ip_search(self.rows, q, k)
        I = np.where(I >= 0, I + id_base, -1)
        return torch.from_numpy(D), torch.from_numpy(I)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, k, ret):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ivr_amd.sharded import ShardedIndex, shard_bounds
    from oracle import search_ref as S
    rng = np.random.default_rng(123)
    X = rng.standard_normal((n, 48), dtype=np.float32)
    if n > 8:
        X[5] = X[n - 3]                   # an exact tie across the two shards: the lower global id must win
    Q = rng.standard_normal((6, 48), dtype=np.float32)
    lo, hi = shard_bounds(n, world)[rank]
    sh = ShardedIndex(OracleShard(48), 48, merge="host")
    sh.add_local(X[lo:hi])
    assert sh.ntotal == n and sh.id_base == lo
    D, I = sh.search(Q, k)
    Dr, Ir = S.flat_ip_search(X, Q, k)
    ok = np.array_equal(I.numpy(), Ir) and np.allclose(D.numpy(), Dr, rtol=1e-6, atol=1e-6)   # BLAS blocking differs per shard
    # every rank holds the same merged answer
    gathered = [None] * world
    dist.all_gather_object(gathered, I.numpy().tolist())
    ok = ok and all(g == gathered[0] for g in gathered)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(1000, 10), (7, 10), (3, 2)])
def test_two_rank_sharded_search_over_gloo(n, k):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_bounds_and_host_merge():
    from ivr_amd.sharded import merge_host, shard_bounds, stride_frames
    from oracle import search_ref as S
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert stride_frames(10, 1, 4).tolist() == [1, 5, 9]
    rng = np.random.default_rng(4)
    Dp = np.sort(rng.standard_normal((3, 5, 4)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    Ip = np.arange(60).reshape(3, 5, 4).astype(np.int64)
    Ip[2, :, 2:] = -1
    Dp[2, :, 2:] = S.NEG_FLT_MAX
    D, I = merge_host(torch.from_numpy(Dp), torch.from_numpy(Ip), 4)
    Dr, Ir = S.merge_shards(Dp, Ip, 4)
    assert np.array_equal(I.numpy(), Ir) and np.array_equal(D.numpy(), Dr)
