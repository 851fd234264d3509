#!/usr/bin/env python3
"""Experiment (stamps build): is the f32 read-modify-write epilogue of the residual GEMMs bound by the chip's HBM rate (all CUs in
their epilogue together) or per CU?  Runs the attn-out / fc2 shapes with 64, 128, 255 and 2400 tiles in flight and prints the median
epilogue cycles per workgroup.   make -C .../csrc stamps && IVR_LIB=.../libivr_hip_stamps.so IVR_GEMM=4 python tools/exp_epilogue_contention.py"""
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
for name, n, k, epi, act in (("attn_out", 768, 768, EPI_RESID, -1), ("fc2", 768, 3072, EPI_RESID, -1), ("fc1", 3072, 768, EPI_STORE, 0)):
    for panels in (21, 43, 85, 800):
        m = panels * 256
        x = (torch.randn((m, k), device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(n, device="cuda")
        r = torch.zeros((m, n), device="cuda") if epi == EPI_RESID else None
        for _ in range(5):
            linear(x, w, b, act=act, epilogue=epi, resid=r)
        torch.cuda.synchronize()
        nb = min(16384, 8 * ((panels + 7) // 8) * ((n + 255) // 256))
        st = np.zeros((nb, 8), dtype=np.uint64)
        assert lib.ivr_debug_gemm_stamps(st.ctypes.data_as(C.c_void_p), nb) == 0
        st = st.astype(np.int64)
        st = st[st[:, 0] > 0]
        pro, loop, epi_c = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
        print(f"{name:9s} tiles {panels * ((n + 255) // 256):5d}: prologue {np.median(pro):7.0f}  loop {np.median(loop):8.0f} ({np.median(loop) / (k // 64):5.0f}/stage)  "
              f"epilogue median {np.median(epi_c):7.0f}  p10 {np.percentile(epi_c, 10):7.0f}  p90 {np.percentile(epi_c, 90):7.0f}", flush=True)
