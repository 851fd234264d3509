#!/usr/bin/env python3
"""Encoder microbenchmark: frames/s and per-kernel HIP-event times of one tower config at one batch.

    python tools/bench_tower.py [b32|l14|dino] [batch] [steps] [bf16|fp8|f32]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import torch  # noqa: E402

from ivr_amd import _ffi, config  # noqa: E402
from ivr_amd.tower import Tower  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "b32"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
compute = sys.argv[4] if len(sys.argv) > 4 else "bf16"
cfg = {"b32": config.CLIP_VIT_B32, "l14": config.CLIP_VIT_L14, "dino": config.DINO_VIT_S16}[name]
tower = Tower(cfg, make_weights(cfg, seed=1), max_batch=batch, compute=compute)
frames = torch.randint(0, 256, (batch, cfg.image, cfg.image, 3), device="cuda", dtype=torch.uint8)
for _ in range(2):
    tower.encode_frames(frames)
torch.cuda.synchronize()
_ffi.profile_reset()
_ffi.profile_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    tower.encode_frames(frames)
e1.record()
torch.cuda.synchronize()
_ffi.profile_enable(False)
ms = e0.elapsed_time(e1) / steps
prof = _ffi.profile_read()
print(f"{name} {compute} batch {batch}: {ms:.2f} ms/step  {batch / ms * 1e3:.0f} frames/s  (profiled: per-kernel events serialise nothing, same stream)")
tot = sum(v["ms"] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    extra = f"  {v['work'] / (v['ms'] * 1e-3) / 1e12:7.1f} T(FLOP|B)/s" if v.get("work") else ""
    print(f"  {k:18s} {v['ms'] / steps:8.3f} ms/step  {100 * v['ms'] / tot:5.1f} %  launches/step {v['launches'] // steps}{extra}")
