"""y = x W^T + b on the tower GEMM kernel (ivr_linear): parity tests and kernel benchmarks."""
import ctypes as C

import torch

from . import _ffi

EPI_STORE, EPI_RESID, EPI_F32 = 0, 1, 3


def linear(x, w, bias=None, act=-1, epilogue=EPI_STORE, resid=None):
    """x [M,K], w [N,K] CUDA tensors, both bf16 or both float32; bias float32 [N] or None."""
    lib = _ffi.load()
    if x.dtype != w.dtype or x.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("x and w must both be bf16 or both float32")
    x, w = x.contiguous(), w.contiguous()
    M, K = x.shape
    N = w.shape[0]
    f32 = x.dtype == torch.float32
    out = None
    if epilogue == EPI_STORE:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    elif epilogue == EPI_F32:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    elif resid is None or resid.dtype != torch.float32 or tuple(resid.shape) != (M, N):
        raise ValueError("EPI_RESID needs a float32 [M,N] residual tensor")
    with torch.cuda.device(x.device):
        _ffi.check(lib.ivr_linear(_ffi.context(x.device.index), int(f32), int(epilogue), C.c_void_p(x.data_ptr()),
                                  C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                  M, N, K, int(act), C.c_void_p(out.data_ptr()) if out is not None else None,
                                  C.c_void_p(resid.data_ptr()) if resid is not None else None, _ffi.stream_ptr()),
                   "ivr_linear")
    return resid if epilogue == EPI_RESID else out
