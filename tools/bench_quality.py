#!/usr/bin/env python3
"""Frame-quality chain (ivr_frame_quality: gray + Laplacian sums + Sobel + NMS in one tiled kernel, hysteresis as reconstruction sweeps
over bit planes) on batches of decoded frames resident in HBM: ms per batch and algorithmic GB/s (3 B read + 2 bits written per pixel) against
the 8 TB/s HBM peak.

    python tools/bench_quality.py [frames=64]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _ffi.load()
for h, w, kind in ((1080, 1920, "smooth"), (1080, 1920, "noise"), (720, 1280, "smooth"), (224, 224, "smooth")):
    g = torch.Generator(device="cuda").manual_seed(1)
    if kind == "noise":
        frames = torch.randint(0, 256, (n, h, w, 3), generator=g, device="cuda", dtype=torch.uint8)
    else:      # smooth gradients + mild noise: a realistic share of edge pixels (a few percent)
        yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
        base = 127 + 80 * torch.sin(xx / 37.0) * torch.cos(yy / 23.0) + 30 * torch.sin((xx + yy) / 11.0)
        frames = (base[None, :, :, None] + 6 * torch.randn((n, h, w, 3), generator=g, device="cuda")).clamp(0, 255).to(torch.uint8)
    lap = torch.empty((n, 2), dtype=torch.int64, device="cuda")
    cnt = torch.empty(n, dtype=torch.int64, device="cuda")

    def run():
        _ffi.check(lib.ivr_frame_quality(_ffi.context(0), C.c_void_p(frames.data_ptr()), n, h, w, 0, 20, 80, C.c_void_p(lap.data_ptr()),
                                         C.c_void_p(cnt.data_ptr()), _ffi.stream_ptr()), "ivr_frame_quality")
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    _ffi.profile_reset()
    _ffi.profile_enable(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    _ffi.profile_enable(False)
    ms = e0.elapsed_time(e1) / 5
    pr = {k: round(v["ms"] / 5, 3) for k, v in _ffi.profile_read().items()}
    npix = n * h * w
    tile_ms = pr.get("quality_tile", ms)
    print(f"{n} x {h}x{w} {kind:6s}: {ms:7.3f} ms per batch ({npix * 3.25 / ms / 1e6:7.1f} GB/s algorithmic over the whole chain); tile kernel "
          f"{tile_ms:.3f} ms = {npix * 3.25 / tile_ms / 1e6:7.1f} GB/s = {npix * 3.25 / tile_ms / 1e6 / 8000:.3f} of 8 TB/s; edges "
          f"{float(cnt.float().mean()) / (h * w) * 100:.1f} % of the pixels", pr)
