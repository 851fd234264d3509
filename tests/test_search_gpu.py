"""GPU parity: HIP flat inner-product index (through the C ABI) vs the CPU oracle.
Bar: ids bit-exact on tie-free data and under the documented tie rule; scores within 1e-5 (the
north-star tolerance is 1e-3)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import search_ref as S

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "golden.json")))["search"]


def _index(X, **kw):
    from ivr_amd.index import FlatIPIndex
    idx = FlatIPIndex(X.shape[1], **kw)
    if len(X):
        idx.add(X)
    return idx


def _check(X, Q, k, atol=1e-5):
    idx = _index(X)
    D, I = idx.search(Q, k)
    Dr, Ir = S.flat_ip_search(X, Q, k, dtype=np.float64)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (len(Q), k)
    assert np.array_equal(I, Ir)
    valid = Ir >= 0
    assert np.abs(D[valid] - Dr[valid]).max(initial=0) <= atol
    assert (D[~valid] == S.NEG_FLT_MAX).all()
    return idx


def test_golden_fixture(golden):
    g = golden("search")
    X = S.normalize_rows_core(np.random.default_rng(META["index_seed"]).standard_normal((META["n"], META["d"]), dtype=np.float32))
    Q = np.random.default_rng(META["query_seed"]).standard_normal((META["q"], META["d"]), dtype=np.float32)
    from ivr_amd.index import FlatIPIndex
    idx = FlatIPIndex(META["d"])
    idx.add(X.astype(np.float32))
    # raw queries, normalised on the device (N2 on the query side, core.py:875)
    D, I = idx.search_device(Q, META["k"], normalize=True)
    assert np.array_equal(I.cpu().numpy(), g["I"])
    assert np.abs(D.cpu().numpy() - g["D"]).max() < 1e-5
    assert idx.ntotal == META["n"] and idx.d == META["d"] and idx.is_trained


@pytest.mark.parametrize("n,d,nq,k", [(1, 16, 1, 1), (15, 16, 3, 5), (63, 32, 2, 10), (64, 512, 10, 10), (65, 512, 1, 50),
                                      (1000, 100, 5, 10), (4097, 384, 17, 10), (5000, 768, 33, 7), (3000, 64, 70, 10),
                                      (20000, 512, 1, 1000), (300, 1024, 4, 300), (2000, 2048, 2, 3), (777, 7, 3, 4)])
def test_shapes_vs_oracle(n, d, nq, k):
    rng = np.random.default_rng(n * 7 + d)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    _check(X, Q, k, atol=1e-4 * np.sqrt(d))


def test_k_larger_than_ntotal_and_empty():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((5, 32), dtype=np.float32)
    _check(X, X[:2], 8)
    from ivr_amd.index import FlatIPIndex
    idx = FlatIPIndex(32)
    D, I = idx.search(X[:3], 4)
    assert (I == -1).all() and (D == S.NEG_FLT_MAX).all()


def test_ties_resolve_to_lower_id():
    X = np.ones((200, 16), np.float32)
    X[37] *= 2                     # unique best
    X[150:] *= 0.5
    idx = _index(X)
    D, I = idx.search(np.ones((1, 16), np.float32), 12)
    assert I[0].tolist() == [37] + [i for i in range(12) if i != 37][:11]
    Z = np.zeros((130, 16), np.float32)   # every score is +0.0 / -0.0
    Z[5, 0] = -0.0
    idx = _index(Z)
    D, I = idx.search(np.ones((2, 16), np.float32), 5)
    assert I.tolist() == [[0, 1, 2, 3, 4]] * 2 and (D == 0).all()


def test_incremental_add_matches_one_shot():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((1234, 96), dtype=np.float32)
    Q = rng.standard_normal((4, 96), dtype=np.float32)
    from ivr_amd.index import FlatIPIndex
    idx = FlatIPIndex(96, capacity=10)            # forces reallocation
    for a, b in [(0, 1), (1, 17), (17, 600), (600, 601), (601, 1234)]:
        idx.add(X[a:b])
    assert idx.ntotal == 1234
    D, I = idx.search(Q, 10)
    Dr, Ir = S.flat_ip_search(X, Q, 10, dtype=np.float64)
    assert np.array_equal(I, Ir)
    assert np.array_equal(idx.reconstruct_n(0, 1234), X)      # rows round-trip bit-exactly through the tiled layout
    assert np.array_equal(idx.reconstruct_n(100, 33), X[100:133])
    idx.reset()
    assert idx.ntotal == 0 and (idx.search(Q, 3)[1] == -1).all()


def test_add_with_normalize_and_ring_write():
    rng = np.random.default_rng(6)
    X = rng.standard_normal((500, 512), dtype=np.float32) * 3
    X[7] = 0                                         # zero row stays zero (core.py:1194-1196 / normalize_L2)
    Q = rng.standard_normal((3, 512), dtype=np.float32)
    from ivr_amd.index import FlatIPIndex
    idx = FlatIPIndex(512)
    idx.add(X, normalize=True)
    Xn = S.normalize_rows_core(X).astype(np.float32)
    assert np.abs(idx.reconstruct_n() - Xn).max() < 2e-7
    assert (idx.reconstruct_n(7, 1) == 0).all()
    Y = rng.standard_normal((40, 512), dtype=np.float32)
    idx.write(123, Y, normalize=True)                 # rolling-window overwrite
    Xn[123:163] = S.normalize_rows_core(Y)
    D, I = idx.search(Q, 10)
    Dr, Ir = S.flat_ip_search(Xn, Q, 10, dtype=np.float64)
    assert np.array_equal(I, Ir) and np.abs(D - Dr).max() < 1e-5
    with pytest.raises(ValueError):
        idx.write(490, Y)                             # past ntotal


def test_normalize_L2_and_nonfinite_count():
    from ivr_amd.index import count_nonfinite_and_normalize, normalize_L2
    rng = np.random.default_rng(8)
    x = rng.standard_normal((1000, 512), dtype=np.float32)
    x[3] = 0
    ref = x.copy()
    S.normalize_rows_faiss(ref)
    normalize_L2(x)                                   # in place, numpy in / numpy out like faiss
    assert np.abs(x - ref).max() < 2e-7 and (x[3] == 0).all()
    t = torch.from_numpy(rng.standard_normal((10, 33), dtype=np.float32)).cuda()
    t[2, 5] = float("nan")
    t[4, 0] = float("inf")
    assert count_nonfinite_and_normalize(t) == 2


def test_merge_parts_equals_global():
    from ivr_amd.index import FlatIPIndex, topk_merge
    rng = np.random.default_rng(9)
    X = rng.standard_normal((3000, 128), dtype=np.float32)
    Q = rng.standard_normal((9, 128), dtype=np.float32)
    bounds = [(0, 1000), (1000, 1003), (1003, 3000)]
    Dp, Ip = [], []
    for a, b in bounds:
        idx = FlatIPIndex(128)
        idx.add(X[a:b])
        D, I = idx.search_device(Q, 10, id_base=a)
        Dp.append(D)
        Ip.append(I)
    D, I = topk_merge(torch.stack(Dp), torch.stack(Ip))
    Dr, Ir = S.flat_ip_search(X, Q, 10, dtype=np.float64)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.abs(D.cpu().numpy() / Dr - 1).max() < 1e-6
    assert (Ip[1][:, 3:] == -1).all()                 # 3-row shard pads with -1 before the merge


def test_error_mapping():
    from ivr_amd.index import FlatIPIndex
    idx = FlatIPIndex(16)
    idx.add(np.ones((4, 16), np.float32))
    with pytest.raises(ValueError):
        idx.search(np.ones((1, 8), np.float32), 3)    # wrong dimension (core.py:879-880)
    with pytest.raises(ValueError):
        idx.search(np.ones((1, 16), np.float32), 0)
    with pytest.raises(ValueError):
        idx.search(np.ones((1, 16), np.float32), 5000)
    with pytest.raises(ValueError):
        FlatIPIndex(0)


@pytest.mark.parametrize("nq", [10])
def test_baseline_config2_full_size(nq):
    """BASELINE config 2 search half at full size: 1M x 512 rows, 10 queries, top-10, ids vs the float64 oracle."""
    n, d = 1_000_000, 512
    from ivr_amd.index import FlatIPIndex
    g = torch.Generator(device="cuda").manual_seed(5678)
    idx = FlatIPIndex(d, capacity=n)
    Xs = []
    for i in range(0, n, 250_000):
        x = torch.randn((250_000, d), generator=g, device="cuda", dtype=torch.float32)
        idx.add(x, normalize=True)
        Xs.append(x.cpu().numpy())
    X = S.normalize_rows_core(np.concatenate(Xs)).astype(np.float32)
    Q = np.random.default_rng(91011).standard_normal((nq, d), dtype=np.float32)
    D, I = idx.search_device(Q, 10, normalize=True)
    Dr, Ir = S.flat_ip_search(X, S.normalize_rows_core(Q).astype(np.float32), 10, dtype=np.float64)
    assert np.array_equal(I.cpu().numpy(), Ir)        # recall@10 == 1.0 and identical order
    assert np.abs(D.cpu().numpy() - Dr).max() < 1e-5
    # size-independent property: searching with stored rows returns the row itself first with score ~1
    rows = torch.from_numpy(X[[0, 123_456, 999_999]]).cuda()
    D2, I2 = idx.search_device(rows, 1)
    assert I2.flatten().tolist() == [0, 123_456, 999_999] and (D2 - 1).abs().max() < 1e-5


def test_baseline_config3_shape_many_queries():
    """BASELINE config 3 per-GPU shape scaled down in N: a 1000-query batch, top-10, against the oracle (the index is read
    once per 64 queries)."""
    rng = np.random.default_rng(31)
    X = S.normalize_rows_core(rng.standard_normal((30_000, 512), dtype=np.float32)).astype(np.float32)
    Q = rng.standard_normal((1000, 512), dtype=np.float32)
    idx = _index(X)
    D, I = idx.search_device(Q, 10, normalize=True)
    Dr, Ir = S.flat_ip_search(X, S.normalize_rows_core(Q).astype(np.float32), 10, dtype=np.float64)
    assert np.array_equal(I.cpu().numpy(), Ir)
    assert np.abs(D.cpu().numpy() - Dr).max() < 1e-5
