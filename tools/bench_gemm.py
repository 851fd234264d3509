#!/usr/bin/env python3
"""Kernel microbenchmark: the tower GEMM shapes of ViT-B/32 at a given batch (HIP-event timed through ivr_profile_*)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import torch  # noqa: E402

from ivr_amd import _ffi  # noqa: E402
from ivr_amd.linear import EPI_RESID, EPI_STORE, linear  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = B * 50
shapes = [("qkv", M, 2304, 768, EPI_STORE, -1), ("attn_out", M, 768, 768, EPI_RESID, -1), ("fc1", M, 3072, 768, EPI_STORE, 0),
          ("fc2", M, 768, 3072, EPI_RESID, -1), ("patch", B * 49, 768, 3072, EPI_STORE, -1), ("sq4096", 4096, 4096, 4096, EPI_STORE, -1)]
for name, m, n, k, epi, act in shapes:
    x = (torch.randn((m, k), device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda") * 0.05).to(torch.bfloat16)
    if os.environ.get("IVR_BENCH_ZERO"):       # DVFS probe: all-zero operands draw less power, so the clock stays up
        x.zero_()
        w.zero_()
    b = torch.randn(n, device="cuda")
    r = torch.zeros((m, n), device="cuda") if epi == EPI_RESID else None
    for _ in range(3):
        linear(x, w, b, act=act, epilogue=epi, resid=r)
    torch.cuda.synchronize()
    _ffi.profile_reset()
    _ffi.profile_enable(True)
    for _ in range(10):
        linear(x, w, b, act=act, epilogue=epi, resid=r)
    torch.cuda.synchronize()
    _ffi.profile_enable(False)
    p = _ffi.profile_read()["linear"]
    line = f"{name:9s} M={m:7d} N={n:5d} K={k:5d}  {p['ms'] / p['launches']:8.3f} ms  {p['work'] / (p['ms'] * 1e-3) / 1e12:8.1f} TFLOP/s"
    if os.environ.get("IVR_BENCH_VENDOR"):
        # the vendor library (hipBLASLt through torch) on the bare product, no bias / activation / residual: a practical ceiling
        wt = w.t()
        for _ in range(3):
            torch.matmul(x, wt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(x, wt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"   | hipBLASLt bare matmul {ms:8.3f} ms {2.0 * m * n * k / (ms * 1e-3) / 1e12:8.1f} TFLOP/s"
    print(line)
