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


def quantize_rows_e4m3(w):
    """float32 [N,K] -> (e4m3 bytes as uint8 [N,K], float32 scale [N]) with one scale per row (absmax / 448): the weight
    format of the fp8 tower mode, restated with torch's float8_e4m3fn conversion (round to nearest even, saturating)."""
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).to(torch.float32)
    q = (w / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def linear_fp8(x8, w8, colscale=None, bias=None, act=-1, epilogue=EPI_STORE, out_fp8=False, resid=None):
    """x8 [M,K], w8 [N,K]: e4m3 bytes (uint8 or float8_e4m3fn CUDA tensors); y = (x8 w8^T) * colscale + bias."""
    lib = _ffi.load()
    x8, w8 = x8.contiguous(), w8.contiguous()
    M, K = x8.shape
    N = w8.shape[0]
    out = None
    if epilogue == EPI_STORE:
        out = torch.empty((M, N), dtype=torch.uint8 if out_fp8 else torch.bfloat16, device=x8.device)
    elif resid is None or resid.dtype != torch.float32 or tuple(resid.shape) != (M, N):
        raise ValueError("EPI_RESID needs a float32 [M,N] residual tensor")
    ptr = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None   # noqa: E731
    with torch.cuda.device(x8.device):
        _ffi.check(lib.ivr_linear_fp8(_ffi.context(x8.device.index), int(epilogue), ptr(x8), ptr(w8), ptr(colscale), ptr(bias),
                                      M, N, K, int(act), ptr(out), int(bool(out_fp8)), ptr(resid), _ffi.stream_ptr()),
                   "ivr_linear_fp8")
    if epilogue == EPI_RESID:
        return resid
    return out.view(torch.float8_e4m3fn) if out_fp8 else out
