#!/usr/bin/env python3
"""Diagnostic: where a 256x256 GEMM workgroup spends its cycles (needs a library built with -DIVR_GEMM_STAMPS)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402
from ivr_amd.linear import EPI_RESID, EPI_STORE, linear  # noqa: E402

lib = _ffi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = B * 50
shapes = [("qkv", M, 2304, 768, EPI_STORE, -1), ("attn_out", M, 768, 768, EPI_RESID, -1), ("fc1", M, 3072, 768, EPI_STORE, 0),
          ("fc2", M, 768, 3072, EPI_RESID, -1), ("sq4096", 4096, 4096, 4096, EPI_STORE, -1)]
for name, m, n, k, epi, act in shapes:
    x = (torch.randn((m, k), device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device="cuda")
    r = torch.zeros((m, n), device="cuda") if epi == EPI_RESID else None
    for _ in range(5):
        linear(x, w, b, act=act, epilogue=epi, resid=r)
    torch.cuda.synchronize()
    nb = min(16384, ((m + 255) // 256) * ((n + 255) // 256))
    st = np.zeros((nb, 6), dtype=np.uint64)
    assert lib.ivr_debug_gemm_stamps(st.ctypes.data_as(C.c_void_p), nb) == 0
    st = st.astype(np.int64)
    pro, loop, epi_c = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    kt = k // 64
    # the counters are per XCD (blockIdx % 8), not synchronised across the chip: take spans per XCD
    spans, reals = [], []
    for xcd in range(8):
        sx = st[xcd::8]
        sx = sx[sx[:, 0] > 0]
        spans.append(sx[:, 3].max() - sx[:, 0].min())
        reals.append((sx[:, 5].max() - sx[:, 4].min()) / 100e6)     # seconds (s_memrealtime ticks at 100 MHz)
    span, real = float(np.median(spans)), float(np.median(reals))
    clk = span / real / 1e9
    print(f"{name:9s} KT={kt:3d} blocks={nb:6d}  prologue {np.median(pro):8.0f}  loop {np.median(loop):8.0f} ({np.median(loop) / kt:6.0f}/stage, ideal 2048)"
          f"  epilogue {np.median(epi_c):8.0f}  total/block {np.median(st[:, 3] - st[:, 0]):8.0f}  span {span} cyc = {real * 1e3:.3f} ms -> in-kernel clock {clk:.2f} GHz")
