#!/usr/bin/env python3
"""Attention kernel time vs sequence length for both query-tile widths of the head-resident kernel (IVR_ATTN_QC=3|4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402
from ivr_amd.config import TowerConfig  # noqa: E402
from ivr_amd.tower import Tower  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402

for grid, patch in ((8, 4), (10, 8), (12, 4), (14, 8), (16, 4), (20, 4), (24, 4)):
    T = grid * grid + 1
    n = max(8, int(2_000_000 / (T * T)) * 8)
    cfg = TowerConfig(f"t{T}", "vision", 768, 1, 12, 3072, T, 64, image=patch * grid, patch=patch)
    tw = Tower(cfg, make_weights(cfg, 1), max_batch=n)
    frames = torch.randint(0, 256, (n, cfg.image, cfg.image, 3), device="cuda", dtype=torch.uint8)
    res = []
    for qc in ("3", "4", "5"):
        os.environ["IVR_ATTN_QC"] = qc
        for _ in range(2):
            tw.encode_frames(frames)
        torch.cuda.synchronize()
        _ffi.profile_reset()
        _ffi.profile_enable(True)
        for _ in range(5):
            tw.encode_frames(frames)
        torch.cuda.synchronize()
        _ffi.profile_enable(False)
        p = _ffi.profile_read()["attention"]
        res.append(p["ms"] / p["launches"])
    print(f"T={T:4d} tiles={-(-T // 16):3d} n={n:5d}: QC=3 {res[0]:7.3f} ms  QC=4 {res[1]:7.3f} ms  QC=5 {res[2]:7.3f} ms  -> {3 + res.index(min(res))}")
