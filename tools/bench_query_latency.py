#!/usr/bin/env python3
"""Interactive path of the reference, one query at a time (system.py:733 -> core.py:1504 encode_text -> unified_index.py:480
search_vectors with k = 50): wall-clock latency per call from Python, host buffers in and out, median / p90 of 200 calls.
The reference logged 38 - 273 ms per text query for the encoder alone on its CUDA box (logs/performance.log:2-7).

    python tools/bench_query_latency.py [rows=1000000]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd.compat import CLIPFeatureExtractor  # noqa: E402
from ivr_amd.index import FlatIPIndex  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def timed(fn, n=200, warm=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[int(len(ts) * 0.9)]


for name, d in (("openai/clip-vit-base-patch32", 512), ("openai/clip-vit-large-patch14", 768)):
    g = torch.Generator(device="cuda").manual_seed(5678)
    idx = FlatIPIndex(d, capacity=rows)
    for i in range(0, rows, 250_000):
        idx.add(torch.randn((min(250_000, rows - i), d), generator=g, device="cuda"), normalize=True)
    for text_compute in ("f32", "bf16"):
        ex = CLIPFeatureExtractor(name, allow_random_init=True, max_batch=8, text_compute=text_compute)
        qv = ex.encode_text("a person riding a bicycle at night")
        enc = timed(lambda: ex.encode_text("a person riding a bicycle at night"))
        sea = timed(lambda: idx.search(qv, 50))
        both = timed(lambda: idx.search(ex.encode_text("a person riding a bicycle at night"), 50))
        if text_compute == "f32":      # search_by_image (system.py:828): one decoded 224 x 224 frame through the bf16 vision tower
            frame = torch.randint(0, 256, (1, 224, 224, 3), device="cuda", dtype=torch.uint8)
            img = timed(lambda: ex.encode_frames(frame).cpu())
            print(f"{name.split('/')[-1]:24s} one image query (bf16 vision tower, frame resident): {img[0]:6.3f} ms (p90 {img[1]:6.3f})", flush=True)
        print(f"{name.split('/')[-1]:24s} text tower {text_compute:4s}: encode_text {enc[0]:6.3f} ms (p90 {enc[1]:6.3f})   search 1 x {rows} x {d}, k=50 "
              f"{sea[0]:6.3f} ms (p90 {sea[1]:6.3f})   text -> top-50 {both[0]:6.3f} ms (p90 {both[1]:6.3f})", flush=True)
        del ex
    idx.close()
