#!/usr/bin/env python3
"""A/B in ONE process, interleaved rounds (guide rule 24): the tile-per-workgroup 256 x 256 GEMM against the persistent kernel of
round 3, with and without the staggered start, on the four tower shapes of ViT-B/32 at 4,096 frames.

    python tools/ab_gemm_pers.py [frames=4096] [rounds=5]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd.linear import EPI_RESID, EPI_STORE, linear  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M = B * 50
shapes = [("qkv", M, 2304, 768, EPI_STORE, -1), ("attn_out", M, 768, 768, EPI_RESID, -1), ("fc1", M, 3072, 768, EPI_STORE, 0),
          ("fc2", M, 768, 3072, EPI_RESID, -1), ("patch", B * 49, 768, 3072, EPI_STORE, -1)]
# (the "tile loop" columns of profiles/r03i_ab_gemm_persistent.log were an experiment kernel - gemm_big_kernel's tile body in a persistent
# loop - that is no longer in the library: it lost to the plain launch on every shape)
variants = [("tile/wg", {"IVR_GEMM_PERS": "0"}),
            ("persistent", {"IVR_GEMM_PERS": "3", "IVR_GEMM_STAGGER": "0"}),
            ("persistent+stagger", {"IVR_GEMM_PERS": "3", "IVR_GEMM_STAGGER": "-1"})]
for name, m, n, k, epi, act in shapes:
    x = (torch.randn((m, k), device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device="cuda")
    r = torch.zeros((m, n), device="cuda") if epi == EPI_RESID else None
    times = {v: [] for v, _ in variants}
    for rd in range(rounds + 1):
        for v, env in variants:
            os.environ.update(env)
            for _ in range(2):
                linear(x, w, b, act=act, epilogue=epi, resid=r)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                linear(x, w, b, act=act, epilogue=epi, resid=r)
            e1.record()
            torch.cuda.synchronize()
            if rd:
                times[v].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * m * n * k
    print(f"{name:9s} M={m} N={n} K={k}: " + "  |  ".join(f"{v}: median {np.median(t):.3f} ms min {min(t):.3f} ({fl / np.median(t) / 1e9:.0f} TF/s)" for v, t in times.items()),
          flush=True)
