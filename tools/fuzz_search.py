#!/usr/bin/env python3
"""Randomised cross-check of ivr_index_search with the bf16 candidate scan against the exact float32 scan of the same build
(IVR_SCAN_BF16=0 index on the same stored rows): D and I must be bit-identical for every draw.

    python tools/fuzz_search.py [seconds=60] [seed=0]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd.index import FlatIPIndex  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, draws, fast_hits, redone_total, bad = time.time(), 0, 0, 0, 0
last_print = t0
while time.time() - t0 < budget:
    d = int(rng.choice([16, 48, 96, 128, 384, 512, 768, 1000]))
    N = int(rng.integers(20_000, 400_000))
    nq = int(rng.choice([1, 3, 10, 16, 17, 64, 65, 130, 300, 1100]))
    k = int(rng.choice([1, 5, 10, 50, 100]))
    kind = rng.choice(["gauss", "clustered", "dups", "scaled", "sparse", "zeros"])
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    X = torch.randn((N, d), generator=g, device="cuda")
    if kind == "clustered":
        X = X[:1].clone() + float(rng.choice([1e-3, 1e-2, 1e-1])) * X
    elif kind == "dups":
        X[torch.randint(0, N, (N // 3,), generator=g, device="cuda")] = X[:N // 3].clone()
    elif kind == "scaled":
        X = X * torch.exp(torch.randn((N, 1), generator=g, device="cuda") * 2)
    elif kind == "sparse":
        X = X * (torch.rand((N, d), generator=g, device="cuda") < 0.05)
    elif kind == "zeros":
        X[torch.randint(0, N, (N // 10,), generator=g, device="cuda")] = 0
    normalize = bool(rng.integers(2)) or kind == "scaled"
    Q = torch.randn((nq, d), generator=g, device="cuda")
    if rng.integers(3) == 0:
        Q[: max(1, nq // 2)] = X[torch.randint(0, N, (max(1, nq // 2),), generator=g, device="cuda")]     # queries that ARE rows
    a = FlatIPIndex(d, capacity=N)
    a.add(X, normalize=normalize)
    os.environ["IVR_SCAN_BF16"] = "0"
    b = FlatIPIndex(d, capacity=N)
    del os.environ["IVR_SCAN_BF16"]
    b.add(X, normalize=normalize)
    Da, Ia = a.search_device(Q, k, normalize=bool(rng.integers(2)) if False else True)
    Db, Ib = b.search_device(Q, k, normalize=True)
    has16, redone = a.scan_stats()
    same = torch.equal(Ia, Ib) and torch.equal(Da, Db)
    draws += 1
    fast_hits += int(has16)
    redone_total += redone
    if not same:
        bad += 1
        print(f"MISMATCH d={d} N={N} nq={nq} k={k} kind={kind} normalize={normalize}: ids differ at {(Ia != Ib).nonzero()[:4].tolist()}")
    del a, b, X, Q
    if time.time() - last_print > 30:
        last_print = time.time()
        print(f"  ... {draws} draws, {bad} mismatches after {last_print - t0:.0f} s", flush=True)
print(f"fuzz: {draws} draws, {bad} mismatches; queries redone exactly in the last chunks: {redone_total}")
sys.exit(1 if bad else 0)
