#!/usr/bin/env python3
"""scanq_kernel rate against the K depth of a work item (d = 256 ... 2048, rows scaled to the same index bytes): how much of an item
is per-item overhead (epilogue, the vmcnt(0) after its stores) rather than K-loop stages.   python tools/bench_scanq_d.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "intelligent-video-analysis-retrieval-system_amd"))
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402
from ivr_amd.index import FlatIPIndex  # noqa: E402

for d in (256, 512, 1024, 2048):
    N = 1_250_000 * 512 // d
    idx = FlatIPIndex(d, capacity=N)
    for i in range(0, N, 250_000):
        idx.add(torch.randn((min(250_000, N - i), d), device="cuda"), normalize=True)
    q = torch.randn((1000, d), device="cuda")
    idx.reserve_search(1000, 10)
    for _ in range(3):
        idx.search_device(q, 10, normalize=True)
    torch.cuda.synchronize()
    _ffi.profile_reset()
    _ffi.profile_enable(1)
    for _ in range(5):
        idx.search_device(q, 10, normalize=True)
    torch.cuda.synchronize()
    _ffi.profile_enable(False)
    p = _ffi.profile_read()["scanq"]
    ms = p["ms"] / p["launches"]
    items = (N + 255) // 256 * 4 / 256
    print(f"d={d:5d} N={N:8d}: scanq {ms:.3f} ms = {p['work'] / p['launches'] / ms / 1e9:7.1f} TFLOP/s; {d // 64} stages per item, "
          f"{items:.1f} items per CU -> {ms * 1e3 / items:.2f} us per item, {ms * 1e3 / items / (d // 64):.3f} us per stage")
    idx.close()
