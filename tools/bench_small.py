#!/usr/bin/env python3
"""Microbenchmarks of the two HBM-bound kernels: preprocess emit and index scan (HIP-event timed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402
from ivr_amd.index import FlatIPIndex  # noqa: E402
from ivr_amd.preprocess import preprocess_frames  # noqa: E402

n = 4096
frames = torch.randint(0, 256, (n, 224, 224, 3), device="cuda", dtype=torch.uint8)
out = torch.empty((n * 49, 3072), dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    preprocess_frames(frames, "identity", patch=32, out=out)
torch.cuda.synchronize()
_ffi.profile_reset(); _ffi.profile_enable(True)
for _ in range(10):
    preprocess_frames(frames, "identity", patch=32, out=out)
torch.cuda.synchronize(); _ffi.profile_enable(False)
p = _ffi.profile_read()["preprocess_emit"]
print(f"preprocess_emit  {p['ms'] / p['launches']:.3f} ms  {p['work'] / (p['ms'] * 1e-3) / 1e9:.0f} GB/s")
N = 1_000_000
idx = FlatIPIndex(512, capacity=N)
for i in range(0, N, 250_000):
    idx.add(torch.randn((250_000, 512), device="cuda"), normalize=True)
for nq in (1, 10, 16, 32, 64):
    q = torch.randn((nq, 512), device="cuda")
    for _ in range(3):
        idx.search_device(q, 10, normalize=True)
    torch.cuda.synchronize()
    _ffi.profile_reset(); _ffi.profile_enable(2)
    for _ in range(10):
        idx.search_device(q, 10, normalize=True)
    torch.cuda.synchronize(); _ffi.profile_enable(False)
    pr = _ffi.profile_read()
    p = pr["scan16_groupmax"] if "scan16_groupmax" in pr else pr["scan_groupmax"]
    tot = sum(v["ms"] for k, v in pr.items()) / 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        idx.search_device(q, 10, normalize=True)
    e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / 50
    if nq == 10:
        print("   ", {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in pr.items()}, "us")
    print(f"scan nq={nq:3d}  {p['ms'] / p['launches']:.3f} ms  {p['work'] / (p['ms'] * 1e-3) / 1e9:.0f} GB/s   sum of kernel events {tot:.3f} ms   "
          f"whole search, back to back on the stream {wall:.3f} ms = {N * nq / wall / 1e6:.1f} G pairs/s")
