#!/usr/bin/env python3
"""BASELINE configs[3] at full size on one GPU: 8 feeds x 30 fps of 1280x720 BGR frames -> HIP resize+normalise ->
ViT-B/32 -> rolling 5M-row x 512-d index -> top-10 of a query batch; the step is captured once as a HIP graph.

    python tools/bench_stream.py [rows=5000000] [feeds=8] [frames_per_feed_per_step=1] [queries=10] [steps=200]

Reports the step latency (graph replay, frames already in HBM) next to the un-captured launch sequence, and the real-time
margin: a step must finish within frames_per_feed_per_step / 30 s.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd import config as C  # noqa: E402
from ivr_amd.index import FlatIPIndex  # noqa: E402
from ivr_amd.streaming import StreamingSession  # noqa: E402
from ivr_amd.tower import Tower  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402

arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d   # noqa: E731
rows, feeds, fpf, nq, steps = arg(1, 5_000_000), arg(2, 8), arg(3, 1), arg(4, 10), arg(5, 200)
n = feeds * fpf
rows -= rows % n
dev = torch.device("cuda", 0)
cfg = C.CLIP_VIT_B32
tower = Tower(cfg, make_weights(cfg, 12), max_batch=n)
index = FlatIPIndex(512, capacity=rows)
g = torch.Generator(device=dev).manual_seed(5678)
for i in range(0, rows, 250_000):
    index.add(torch.randn((min(250_000, rows - i), 512), generator=g, device=dev), normalize=True)
queries = torch.from_numpy(np.random.default_rng(91011).standard_normal((nq, 512), dtype=np.float32))
frames = torch.randint(0, 256, (n, 720, 1280, 3), generator=g, device=dev, dtype=torch.uint8)
for use_graph in (True, False):
    sess = StreamingSession(tower, index, n, 720, 1280, queries, k=10, mode="stretch", bgr=True, use_graph=use_graph)
    for _ in range(5):
        sess.step(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sess.step(frames)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    budget = fpf / 30.0 * 1e3
    print(f"{'hipGraph replay' if use_graph else 'plain launches '}: {ms:7.3f} ms per step of {n} frames ({feeds} feeds x {fpf}) + top-10 of {nq} queries "
          f"over {rows} rows; real-time budget {budget:.1f} ms -> {budget / ms:.1f}x margin; {rows * nq / (ms * 1e-3) / 1e9:.1f} G pairs/s incl. embed; "
          f"bf16 scan copy / queries redone exactly in the last step: {index.scan_stats()}")
