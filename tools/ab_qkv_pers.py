#!/usr/bin/env python3
"""A/B in one process, interleaved rounds: the fused QKV + attention kernel, one tile per workgroup vs its persistent form (next
item's first K stage fetched during the attention phase), inside the ViT-B/32 tower at 4,096 frames.   python tools/ab_qkv_pers.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402
from ivr_amd import config as C  # noqa: E402
from ivr_amd.tower import Tower  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = C.CLIP_VIT_B32
tw = Tower(cfg, make_weights(cfg, 12), max_batch=B)
px = (torch.randn((B * 49, 3072), device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty((B, 512), device="cuda")
res = {"0": [], "1": []}
for rd in range(7):
    for mode in ("0", "1"):
        os.environ["IVR_QKV_PERS"] = mode
        for _ in range(2):
            tw.encode_patches(px, B, out=out)
        torch.cuda.synchronize()
        _ffi.profile_reset()
        _ffi.profile_enable(1)
        for _ in range(4):
            tw.encode_patches(px, B, out=out)
        torch.cuda.synchronize()
        _ffi.profile_enable(False)
        p = _ffi.profile_read()["gemm_qkv_attention"]
        if rd:
            res[mode].append(p["ms"] / p["launches"])
for mode, t in res.items():
    print(f"IVR_QKV_PERS={mode}: fused QKV + attention median {np.median(t):.4f} ms per launch, min {min(t):.4f}")
