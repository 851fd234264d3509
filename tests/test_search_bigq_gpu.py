"""GPU parity of the LARGE-batch candidate scan (search_scanq.hip; more than 64 queries per ivr_index_search call - BASELINE
configs[2]: 1,000 queries against a 1.25M-row shard; replaces faiss.IndexFlatIP.search at unified_index.py:503).

The scan only ranks 16-row tiles; every reported score is an exact float32 re-score and every query is verified on the device
(failed ones are redone by the list-driven exact pass), so the contract is the small-batch one:
  (1) D and I bit-identical to the exact float32 scan of the same build (IVR_SCAN_BF16=0) and to the 64-query chunks
      (IVR_SCAN_BIGQ=0), whatever the data does to the approximate ranking;
  (2) ids equal to the float64 brute force (oracle/search_ref.py) outside float32 near-ties, scores within 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import search_ref as S

pytestmark = pytest.mark.gpu


def _index_with(env, d, X, normalize=False):
    from ivr_amd.index import FlatIPIndex
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        idx = FlatIPIndex(d, capacity=len(X))
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    idx.add(X, normalize=normalize)
    return idx


def _check(idx, X, Q, k, normalize_q=True, oracle_rows=None):
    """X = the stored rows.  Returns the number of queries the large-batch path redid exactly."""
    Dt, It = idx.search_device(Q, k, normalize=normalize_q)
    stats = idx.scan_stats()
    exact = _index_with({"IVR_SCAN_BF16": "0"}, idx.d, X)
    De, Ie = exact.search_device(Q, k, normalize=normalize_q)
    assert torch.equal(It, Ie), (It != Ie).nonzero()[:5].tolist()
    assert torch.equal(Dt, De)
    chunks = _index_with({"IVR_SCAN_BIGQ": "0"}, idx.d, X)
    Dc, Ic = chunks.search_device(Q, k, normalize=normalize_q)
    assert torch.equal(It, Ic) and torch.equal(Dt, Dc)
    sel = np.arange(len(Q)) if oracle_rows is None else oracle_rows
    qn = S.normalize_rows_core(Q[sel].astype(np.float64)) if normalize_q else Q[sel].astype(np.float64)
    Dr, Ir = S.flat_ip_search(X.astype(np.float64), qn, k + 1, dtype=np.float64)
    D, I = Dt.cpu().numpy()[sel], It.cpu().numpy()[sel]
    gap = np.minimum(np.abs(np.diff(Dr, axis=1, prepend=np.inf))[:, :k], np.abs(np.diff(Dr, axis=1))[:, :k])
    firm = gap > 1e-6 * max(1.0, np.abs(Dr).max())
    assert np.array_equal(I[firm], Ir[:, :k][firm]), np.argwhere((I != Ir[:, :k]) & firm)[:5]
    assert np.abs(D - Dr[:, :k]).max() < 1e-5 * max(1.0, np.abs(Dr).max())
    return stats


@pytest.mark.parametrize("d", [48, 96, 512, 768, 1000])
@pytest.mark.parametrize("nq,k", [(65, 10), (256, 1), (300, 50), (1000, 10)])
def test_large_batch_scan_is_exact_on_random_rows(d, nq, k):
    rng = np.random.default_rng(d * 13 + nq)
    N = 100_003 if d != 512 else 180_017          # not a multiple of 16, 128 or 256: the last tile, block and slab are ragged
    X = S.normalize_rows_core(rng.standard_normal((N, d)).astype(np.float32)).astype(np.float32)
    idx = _index_with({}, d, X)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    has16, redone = _check(idx, X, Q, k, oracle_rows=np.arange(0, nq, max(1, nq // 48)))
    assert has16
    print(f"d={d} nq={nq} k={k}: {redone} of {min(nq, 1024)} queries redone exactly")
    assert redone <= nq // 50                      # well separated random scores: the measured bound must pass nearly always


def test_clustered_rows_every_query_goes_through_the_list_driven_exact_pass():
    rng = np.random.default_rng(3)
    d, N = 512, 120_000
    base = rng.standard_normal(d).astype(np.float32)
    X = S.normalize_rows_core((base[None, :] + 2e-3 * rng.standard_normal((N, d))).astype(np.float32)).astype(np.float32)
    idx = _index_with({}, d, X)
    Q = (base[None, :] + 1e-2 * rng.standard_normal((150, d))).astype(np.float32)
    has16, redone = _check(idx, X, Q, 10)
    assert has16 and redone == 150


def test_mixed_batch_some_queries_fail_some_pass():
    """Half the rows are a tight cluster (queries aimed at it cannot be verified), half are random: the failure list holds only
    the cluster's queries, in any order, and both halves come out exact."""
    rng = np.random.default_rng(9)
    d, N = 256, 150_000
    base = rng.standard_normal(d).astype(np.float32)
    X = rng.standard_normal((N, d)).astype(np.float32)
    X[::2] = base[None, :] + 1e-3 * rng.standard_normal((N // 2, d)).astype(np.float32)
    X = S.normalize_rows_core(X).astype(np.float32)
    idx = _index_with({}, d, X)
    Q = rng.standard_normal((200, d)).astype(np.float32)
    Q[::4] = base[None, :] + 1e-2 * rng.standard_normal((50, d)).astype(np.float32)
    has16, redone = _check(idx, X, Q, 10)
    assert has16 and 50 <= redone < 200


def test_exact_duplicates_and_unnormalised_operands():
    rng = np.random.default_rng(4)
    d, N = 256, 100_000
    X = (rng.standard_normal((N, d)) * rng.uniform(0.2, 5.0, (N, 1))).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    dup = np.arange(40) * 2048 + 77
    X[dup] = q * 3.0
    idx = _index_with({}, d, X)
    Q = np.concatenate([q[None, :], -q[None, :], (rng.standard_normal((98, d)) * 3).astype(np.float32)])
    D, I = idx.search_device(Q, 10, normalize=False)
    assert I[0].cpu().tolist() == dup[:10].tolist()          # ties resolve to the lowest ids
    _check(idx, X, Q, 10, normalize_q=False)


def test_ring_overwrites_keep_the_residual_bound_valid():
    rng = np.random.default_rng(6)
    d, N = 512, 131_072
    X = rng.standard_normal((N, d)).astype(np.float32)
    idx = _index_with({}, d, X, normalize=True)
    Y = rng.standard_normal((4096, d)).astype(np.float32)
    idx.write(50_001, Y[:1000], normalize=True)
    cursor = torch.zeros(1, dtype=torch.int64, device="cuda")
    for i in range(3):
        idx.write_ring(torch.from_numpy(Y[1024 * i:1024 * (i + 1)]).cuda(), cursor, normalize=True)
    Xn = idx.reconstruct_n(0, N)
    Q = np.concatenate([Y[5:60], Y[1500:1560], rng.standard_normal((85, d)).astype(np.float32)])
    _check(idx, Xn, Q, 10)


def test_small_index_and_k_beyond_the_large_batch_range_take_the_chunked_path():
    rng = np.random.default_rng(11)
    X = S.normalize_rows_core(rng.standard_normal((3_000, 128)).astype(np.float32)).astype(np.float32)
    idx = _index_with({}, 128, X)
    Q = rng.standard_normal((100, 128)).astype(np.float32)
    _check(idx, X, Q, 10)
    X2 = S.normalize_rows_core(rng.standard_normal((200_000, 128)).astype(np.float32)).astype(np.float32)
    idx2 = _index_with({}, 128, X2)
    _check(idx2, X2, Q, 200, oracle_rows=np.arange(0, 100, 7))


def test_more_than_one_chunk_of_1024_queries():
    rng = np.random.default_rng(12)
    X = S.normalize_rows_core(rng.standard_normal((90_000, 256)).astype(np.float32)).astype(np.float32)
    idx = _index_with({}, 256, X)
    Q = rng.standard_normal((2500, 256)).astype(np.float32)
    _check(idx, X, Q, 5, oracle_rows=np.arange(0, 2500, 41))


def test_baseline_config3_shard_full_size():
    """BASELINE configs[2] per-GPU shard at full size: 1.25M x 512 rows, 1,000 queries, top-10; 64 sampled queries against the
    float64 oracle, all of them against the 64-query chunks of the same build."""
    n, d, nq = 1_250_000, 512, 1000
    from ivr_amd.index import FlatIPIndex
    g = torch.Generator(device="cuda").manual_seed(5678)
    idx = FlatIPIndex(d, capacity=n)
    old = os.environ.get("IVR_SCAN_BIGQ")
    os.environ["IVR_SCAN_BIGQ"] = "0"
    try:
        chunks = FlatIPIndex(d, capacity=n)
    finally:
        if old is None:
            del os.environ["IVR_SCAN_BIGQ"]
        else:
            os.environ["IVR_SCAN_BIGQ"] = old
    Xs = []
    for i in range(0, n, 250_000):
        x = torch.randn((250_000, d), generator=g, device="cuda", dtype=torch.float32)
        idx.add(x, normalize=True)
        chunks.add(x, normalize=True)
        Xs.append(x.cpu().numpy())
    X = S.normalize_rows_core(np.concatenate(Xs)).astype(np.float32)
    Q = np.random.default_rng(91011).standard_normal((nq, d), dtype=np.float32)
    D, I = idx.search_device(Q, 10, normalize=True)
    has16, redone = idx.scan_stats()
    Dc, Ic = chunks.search_device(Q, 10, normalize=True)
    assert torch.equal(I, Ic) and torch.equal(D, Dc)
    sample = np.arange(0, nq, nq // 64)[:64]
    Dr, Ir = S.flat_ip_search(X, S.normalize_rows_core(Q[sample]).astype(np.float32), 10, dtype=np.float64)
    assert np.array_equal(I.cpu().numpy()[sample], Ir)
    assert np.abs(D.cpu().numpy()[sample] - Dr).max() < 1e-5
    print(f"configs[2] shard: {redone} of {nq} queries redone exactly")
    assert redone <= 20


def test_pruned_rescore_equals_the_full_rescore():
    """The large-batch re-score skips selected tiles whose approximate maximum is more than twice the error bound below the k-th
    selected tile's (they cannot hold a top-k row): same D and I as with every selected tile fetched (IVR_SCAN_PRUNE=0), on rows
    where most tiles are pruned (Gaussian) and on rows where none can be (a tight cluster)."""
    rng = np.random.default_rng(21)
    d, N = 384, 160_000
    base = rng.standard_normal(d).astype(np.float32)
    for X in (rng.standard_normal((N, d)).astype(np.float32), (base[None, :] + 3e-3 * rng.standard_normal((N, d))).astype(np.float32)):
        X = S.normalize_rows_core(X).astype(np.float32)
        Q = np.concatenate([rng.standard_normal((150, d)).astype(np.float32), X[:50] + 1e-3 * rng.standard_normal((50, d)).astype(np.float32)])
        a, b = _index_with({}, d, X), _index_with({"IVR_SCAN_PRUNE": "0"}, d, X)
        for k in (1, 10, 40):
            Da, Ia = a.search_device(Q, k, normalize=True)
            Db, Ib = b.search_device(Q, k, normalize=True)
            assert torch.equal(Ia, Ib) and torch.equal(Da, Db)
