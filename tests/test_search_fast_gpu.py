"""GPU parity of the bf16 candidate scan behind ivr_index_search: the result must stay bit-exact in the ids (scores within
1e-5) whatever the data does to the approximate ranking, because every query is verified on the device and redone by the
exact float32 scan when the bound cannot exclude the skipped groups.  Oracle: float64 brute force (oracle/search_ref.py)."""
import numpy as np
import pytest
import torch

from oracle import search_ref as S

pytestmark = pytest.mark.gpu


def _check(idx, X, Q, k, normalize_q=True):
    """(1) against the float64 brute force: ids equal except inside float32 near-ties (|gap| < 1e-6, where float32 scoring
    may order two rows either way - the exact float32 path does the same), scores within 1e-5;
    (2) against the exact float32 scan of this build (IVR_SCAN_BF16=0 index on the same rows): bit-identical D and I."""
    import os
    from ivr_amd.index import FlatIPIndex
    Dt, It = idx.search_device(Q, k, normalize=normalize_q)
    D, I = Dt.cpu().numpy(), It.cpu().numpy()
    qn = S.normalize_rows_core(Q.astype(np.float64)) if normalize_q else Q.astype(np.float64)
    Dr, Ir = S.flat_ip_search(X.astype(np.float64), qn, k + 1, dtype=np.float64)
    gap = np.minimum(np.abs(np.diff(Dr, axis=1, prepend=np.inf))[:, :k], np.abs(np.diff(Dr, axis=1))[:, :k])
    firm = gap > 1e-6 * max(1.0, np.abs(Dr).max())
    assert np.array_equal(I[firm], Ir[:, :k][firm]), np.argwhere((I != Ir[:, :k]) & firm)[:5]
    assert np.abs(D - Dr[:, :k]).max() < 1e-5 * max(1.0, np.abs(Dr).max())        # float32 scoring
    old = os.environ.get("IVR_SCAN_BF16")
    os.environ["IVR_SCAN_BF16"] = "0"
    try:
        ref = FlatIPIndex(idx.d, capacity=len(X))
    finally:
        if old is None:
            del os.environ["IVR_SCAN_BF16"]
        else:
            os.environ["IVR_SCAN_BF16"] = old
    ref.add(X, normalize=False)                      # X are the stored rows already
    De, Ie = ref.search_device(Q, k, normalize=normalize_q)
    assert not ref.scan_stats()[0]
    assert torch.equal(It, Ie) and torch.equal(Dt, De)
    return idx.scan_stats()


@pytest.mark.parametrize("d", [96, 512, 768])
@pytest.mark.parametrize("nq,k", [(1, 1), (10, 10), (33, 50), (100, 10)])
def test_candidate_scan_is_exact_on_random_rows(d, nq, k):
    from ivr_amd.index import FlatIPIndex
    rng = np.random.default_rng(d * 7 + nq)
    N = 150_000
    X = rng.standard_normal((N, d)).astype(np.float32)
    idx = FlatIPIndex(d, capacity=N)
    idx.add(X, normalize=True)
    Xn = idx.reconstruct_n(0, N)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    has16, redone = _check(idx, Xn, Q, k)
    assert has16
    print(f"d={d} nq={nq} k={k}: {redone} queries of the last chunk redone exactly")
    assert redone == 0           # well separated random scores: the bound must pass (else the fast path is useless)


def test_clustered_rows_fall_back_and_stay_exact():
    """Rows within 1e-4 of each other: the bf16 ranking is noise, the bound cannot exclude anything -> exact pass."""
    from ivr_amd.index import FlatIPIndex
    rng = np.random.default_rng(3)
    d, N = 512, 120_000
    base = rng.standard_normal(d).astype(np.float32)
    X = (base[None, :] + 2e-3 * rng.standard_normal((N, d))).astype(np.float32)
    idx = FlatIPIndex(d, capacity=N)
    idx.add(X, normalize=True)
    Xn = idx.reconstruct_n(0, N)
    Q = (base[None, :] + 1e-2 * rng.standard_normal((20, d))).astype(np.float32)
    has16, redone = _check(idx, Xn, Q, 10)
    assert has16 and redone > 0


def test_many_exact_duplicates_across_groups():
    """40 identical rows spread over 40 groups, k = 10: ties resolve to the lowest ids; the 33rd duplicate's group is excluded
    with an approximate maximum equal to the k-th score, so the strict bound fails and the exact pass decides."""
    from ivr_amd.index import FlatIPIndex
    rng = np.random.default_rng(4)
    d, N = 256, 100_000
    X = rng.standard_normal((N, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    dup = np.arange(40) * 2048 + 77
    X[dup] = q * 3.0
    idx = FlatIPIndex(d, capacity=N)
    idx.add(X, normalize=True)
    Xn = idx.reconstruct_n(0, N)
    D, I = idx.search_device(q[None, :], 10, normalize=True)
    assert I[0].cpu().tolist() == dup[:10].tolist()
    _check(idx, Xn, np.stack([q, -q, rng.standard_normal(d).astype(np.float32)]), 10)


def test_unnormalised_rows_and_queries():
    from ivr_amd.index import FlatIPIndex
    rng = np.random.default_rng(5)
    d, N = 512, 100_000
    X = (rng.standard_normal((N, d)) * rng.uniform(0.2, 5.0, (N, 1))).astype(np.float32)
    idx = FlatIPIndex(d, capacity=N)
    idx.add(X, normalize=False)
    Q = (rng.standard_normal((12, d)) * 3).astype(np.float32)
    _check(idx, X, Q, 10, normalize_q=False)


def test_overwrites_and_ring_keep_the_scan_copy_in_sync():
    from ivr_amd.index import FlatIPIndex
    rng = np.random.default_rng(6)
    d, N = 512, 131_072
    X = rng.standard_normal((N, d)).astype(np.float32)
    idx = FlatIPIndex(d, capacity=N)
    idx.add(X, normalize=True)
    Y = rng.standard_normal((4096, d)).astype(np.float32)
    idx.write(50_001, Y[:1000], normalize=True)                     # unaligned start
    cursor = torch.zeros(1, dtype=torch.int64, device="cuda")
    for i in range(3):
        idx.write_ring(torch.from_numpy(Y[1024 * i:1024 * (i + 1)]).cuda(), cursor, normalize=True)
    Xn = idx.reconstruct_n(0, N)
    Q = np.concatenate([Y[5:10], Y[1500:1505], rng.standard_normal((6, d)).astype(np.float32)])
    _check(idx, Xn, Q, 10)


def test_switch_off(monkeypatch):
    from ivr_amd.index import FlatIPIndex
    monkeypatch.setenv("IVR_SCAN_BF16", "0")
    rng = np.random.default_rng(8)
    d, N = 512, 100_000
    X = rng.standard_normal((N, d)).astype(np.float32)
    idx = FlatIPIndex(d, capacity=N)
    idx.add(X, normalize=True)
    has16, _ = _check(idx, idx.reconstruct_n(0, N), rng.standard_normal((10, d)).astype(np.float32), 10)
    assert not has16


def test_randomised_cross_check_against_exact_scan():
    """tools/fuzz_search.py for a few seconds: random d, N, nq, k and row distributions (clusters, duplicates, scaled, sparse,
    zero rows); candidate-scan result must equal the exact float32 scan's bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_search.py"), "8", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("d", [250, 384, 512])
def test_ring_scan_is_bit_identical_to_the_register_scan(d):
    """Round 3: for at most 16 queries the bf16 candidate scan streams the index through per-wave LDS-DMA rings (scan16_ring_kernel:
    PIECES = 8 / 12 / 16) instead of through registers - same MFMA sequence, so the same group maxima, hence the same D and I as with
    IVR_SCAN_RING=0; ragged row counts (partial last group, fewer groups than waves), 1 .. 16 queries, ring overwrites."""
    import os
    from ivr_amd.index import FlatIPIndex
    rng = np.random.default_rng(d)
    for N in (70_001, 9_000):
        X = rng.standard_normal((N, d)).astype(np.float32)
        a = FlatIPIndex(d, capacity=N)
        os.environ["IVR_SCAN_RING"] = "0"
        try:
            b = FlatIPIndex(d, capacity=N)
        finally:
            del os.environ["IVR_SCAN_RING"]
        a.add(X, normalize=True)
        b.add(X, normalize=True)
        for nq, k in ((1, 1), (10, 10), (16, 50)):
            Q = rng.standard_normal((nq, d)).astype(np.float32)
            Da, Ia = a.search_device(Q, k, normalize=True)
            Db, Ib = b.search_device(Q, k, normalize=True)
            assert torch.equal(Ia, Ib) and torch.equal(Da, Db), (N, nq, k)
        if N == 70_001:
            _check(a, a.reconstruct_n(0, N), rng.standard_normal((10, d)).astype(np.float32), 10)
