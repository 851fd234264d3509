"""Host wrapper of the HIP frame-preprocessing kernels (ivr_preprocess).

Mirrors what `HFCLIPProcessor(images=...)` (core.py:1613) and
`Image.resize` + the ViT processor (video_frame_filter.py:58-59,29) compute,
but takes a batch of same-sized uint8 NHWC frames already in HBM.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .config import CLIP_MEAN, CLIP_STD


def preprocess_frames(frames, mode="identity", mean=CLIP_MEAN, std=CLIP_STD, bgr=False, size=224, patch=None,
                      out_dtype=torch.bfloat16, bilinear=False, out=None):
    """frames: uint8 [n,h,w,3] (numpy or CUDA tensor).  Returns a CUDA tensor:
    patch=None -> [n,3,size,size] (NCHW);  patch=P -> patch-major [n*(size/P)^2, Kpad], Kpad = 3*P*P rounded up to 64.
    """
    lib = _ffi.load()
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames))
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be uint8 [n,h,w,3]")
    if not frames.is_cuda:
        frames = frames.cuda()
    frames = frames.contiguous()
    n, h, w, _ = frames.shape
    if mode not in _ffi.PP_MODE:
        raise ValueError(f"unknown mode {mode}")
    flags = _ffi.PP_MODE[mode]
    if bgr:
        flags |= _ffi.PP_BGR
    if out_dtype == torch.float32:
        flags |= _ffi.PP_OUT_F32
    elif out_dtype != torch.bfloat16:
        raise ValueError("out_dtype must be torch.bfloat16 or torch.float32")
    if bilinear:
        flags |= _ffi.PP_BILINEAR
    if patch is not None:
        flags |= _ffi.PP_OUT_PATCH_MAJOR
        g = size // patch
        kpad = -(-3 * patch * patch // 64) * 64
        shape = (n * g * g, kpad)
    else:
        shape = (n, 3, size, size)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=frames.device)
    elif tuple(out.shape) != shape or out.dtype != out_dtype or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous {out_dtype} tensor of shape {shape}")
    with torch.cuda.device(frames.device):
        _ffi.check(lib.ivr_preprocess(_ffi.context(frames.device.index), C.c_void_p(frames.data_ptr()), n, h, w, flags,
                                      _ffi.f3(mean), _ffi.f3(std), int(size), int(patch or size),
                                      C.c_void_p(out.data_ptr()), _ffi.stream_ptr()), "ivr_preprocess")
    return out
