#!/usr/bin/env python3
"""Diagnostic: where a 256x256 GEMM workgroup (and the fused QKV + attention workgroup) spends its cycles.  Needs a library built
with -DIVR_GEMM_STAMPS:  make -C .../csrc stamps  ->  lib/libivr_hip_stamps.so, then  IVR_LIB=.../libivr_hip_stamps.so python tools/gemm_stamps.py"""
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
    st = np.zeros((nb, 8), dtype=np.uint64)
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

# fused QKV projection + attention (ViT-B/32 shape: T = 50, D = 768, 12 heads): one tower layer through the debug entry point
from ivr_amd import config as Cfg  # noqa: E402
from ivr_amd.tower import Tower  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402

os.environ["IVR_FUSED_QKV"] = "1"
cfg = Cfg.TowerConfig("stamps-b32-1layer", "vision", 768, 1, 12, 3072, 50, 512, image=224, patch=32)
tw = Tower(cfg, make_weights(cfg, 1), max_batch=B)
px = (torch.randn((B * 49, 3072), device="cuda") * 0.5).to(torch.bfloat16)
for _ in range(3):
    tw.encode_patches(px, B)
torch.cuda.synchronize()
# the last launches of a layer overwrite the stamps: run the fused kernel last by reading right after a dedicated call
lib.ivr_debug_last_qkv_attention.restype = C.c_int
nb = min(16384, 8 * (((B * 50 + 249) // 250 + 7) // 8) * 12)
st = np.zeros((nb, 8), dtype=np.uint64)
assert lib.ivr_debug_last_qkv_attention(tw._h, B, None) == 0
torch.cuda.synchronize()
assert lib.ivr_debug_gemm_stamps(st.ctypes.data_as(C.c_void_p), nb) == 0
st = st.astype(np.int64)
st = st[st[:, 0] > 0]
print(f"qkv_attn  blocks={len(st)}  prologue {np.median(st[:, 1] - st[:, 0]):8.0f}  K loop {np.median(st[:, 2] - st[:, 1]):8.0f} "
      f"({np.median(st[:, 2] - st[:, 1]) / 12:6.0f}/stage, 1536 = MFMA-bound)  acc->LDS {np.median(st[:, 6] - st[:, 2]):8.0f}  "
      f"attention {np.median(st[:, 3] - st[:, 6]):8.0f}  total/block {np.median(st[:, 3] - st[:, 0]):8.0f}")
