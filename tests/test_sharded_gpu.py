"""GPU: the sharded path with real HIP shards in one process (world_size 1 plus an emulated 3-shard merge)."""
import numpy as np
import pytest
import torch

from oracle import search_ref as S

pytestmark = pytest.mark.gpu


def test_emulated_shards_device_and_host_merge_agree():
    from ivr_amd.index import FlatIPIndex, topk_merge, topk_merge_packed, topk_pack
    from ivr_amd.sharded import ShardedIndex, merge_host, shard_bounds
    rng = np.random.default_rng(77)
    X = rng.standard_normal((5000, 512), dtype=np.float32)
    X[10] = X[4000]
    Q = rng.standard_normal((12, 512), dtype=np.float32)
    Dp, Ip = [], []
    for lo, hi in shard_bounds(5000, 3):
        idx = FlatIPIndex(512)
        idx.add(X[lo:hi])
        D, I = idx.search_device(Q, 10, id_base=lo)
        Dp.append(D)
        Ip.append(I)
    Dp, Ip = torch.stack(Dp), torch.stack(Ip)
    Dd, Id = topk_merge(Dp, Ip)
    Dh, Ih = merge_host(Dp, Ip, 10)
    Dr, Ir = S.flat_ip_search(X, Q, 10, dtype=np.float64)
    assert np.array_equal(Id.cpu().numpy(), Ir) and np.array_equal(Ih.numpy(), Ir)
    assert np.array_equal(Dd.cpu().numpy(), Dh.numpy())
    # the wire format of the all-gather: pack each shard's candidates, merge straight from the stacked buffer (ids beyond 2^32 too)
    big = 5_000_000_000
    packed = torch.stack([topk_pack(d, torch.where(i >= 0, i + big, i)) for d, i in zip(Dp, Ip)])
    Dk, Ik = topk_merge_packed(packed)
    assert torch.equal(Dk, Dd) and torch.equal(Ik, Id + big)
    short = FlatIPIndex(512)
    short.add(X[:3])
    Ds, Is = short.search_device(Q, 10, id_base=7)                       # 3 rows: seven unused (-1) slots per query
    Dm, Im = topk_merge_packed(torch.stack([topk_pack(Ds, Is), topk_pack(Dp[2], Ip[2])]))
    Dn, In = topk_merge(torch.stack([Ds, Dp[2]]), torch.stack([Is, Ip[2]]))
    assert torch.equal(Dm, Dn) and torch.equal(Im, In)
    one = ShardedIndex(FlatIPIndex(512), 512)
    one.add_local(X)
    D1, I1 = one.search(Q, 10)
    assert np.array_equal(I1.cpu().numpy(), Ir) and one.ntotal == 5000
