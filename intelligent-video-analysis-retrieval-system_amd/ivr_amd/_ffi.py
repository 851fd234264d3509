"""ctypes binding of libivr_hip.so (include/ivr_api.h).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, this module raises.  Status codes map to the exception types the
reference raises at the same places (IVR_ERR_INVALID -> ValueError as in
core.py:1178-1191, everything else -> RuntimeError as in core.py:894-896).
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# IVR_LIB: diagnostics only (e.g. a -DIVR_GEMM_STAMPS build for tools/gemm_stamps.py)
LIB_PATH = os.environ.get("IVR_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libivr_hip.so")

IVR_MAX_K = 2048
# flags of ivr_preprocess
PP_MODE = {"identity": 0, "shortest_edge_crop": 1, "stretch": 2, "letterbox": 3}
PP_BGR, PP_OUT_F32, PP_OUT_PATCH_MAJOR, PP_BILINEAR = 1 << 4, 1 << 5, 1 << 6, 1 << 7


class TowerDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("kind", "width", "layers", "heads", "mlp", "tokens", "out_dim", "act", "pool",
                                       "image", "patch", "pre_ln", "patch_bias", "vocab", "eos_id", "causal",
                                       "compute")] + [("ln_eps", C.c_float), ("fp8_sites", C.c_int), ("fp8_mlp_cls_bf16", C.c_int),
                                                                      ("fp8_first_layer", C.c_int)]


API_VERSION = 4
FP8_SITE = {"qkv": 1, "o": 2, "fc1": 4, "fc2": 8}


_p = C.c_void_p
_i, _i64, _f = C.c_int, C.c_int64, C.c_float
_SIGS = {
    "ivr_api_version": (_i, []),
    "ivr_init": (_i, [_i, C.POINTER(_p)]),
    "ivr_destroy": (_i, [_p]),
    "ivr_last_error": (C.c_char_p, [_p]),
    "ivr_device_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i64), C.c_char_p, _i]),
    "ivr_release_stream_scratch": (_i, [_p, _p]),
    "ivr_profile_enable": (_i, [_p, _i]),
    "ivr_profile_reset": (_i, [_p]),
    "ivr_profile_json": (_i, [_p, C.c_char_p, _i]),
    "ivr_preprocess": (_i, [_p, _p, _i, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _i, _i, _p, _p]),
    "ivr_preprocess_scratch_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "ivr_tower_create": (_i, [_p, C.POINTER(TowerDesc), C.POINTER(_p)]),
    "ivr_tower_set_weight": (_i, [_p, C.c_char_p, _p, _i64]),
    "ivr_tower_finalize": (_i, [_p, _i]),
    "ivr_tower_destroy": (_i, [_p]),
    "ivr_tower_encode_image": (_i, [_p, _p, _i, _i, _p, _p]),
    "ivr_tower_encode_text": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "ivr_tower_debug_hidden": (_i, [_p, _i, _i, _p, _p]),
    "ivr_tower_workspace_bytes": (_i64, [_p]),
    "ivr_linear": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "ivr_quantize_e4m3_host": (_i, [_p, _p, _i64]),
    "ivr_linear_fp8": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "ivr_l2_normalize": (_i, [_p, _p, _i64, _i, _p, _p]),
    "ivr_index_create": (_i, [_p, _i, _i64, C.POINTER(_p)]),
    "ivr_index_destroy": (_i, [_p]),
    "ivr_index_scan_stats": (_i, [_p, _p]),
    "ivr_index_reset": (_i, [_p]),
    "ivr_index_ntotal": (_i64, [_p]),
    "ivr_index_dim": (_i, [_p]),
    "ivr_index_capacity": (_i64, [_p]),
    "ivr_index_add": (_i, [_p, _p, _i64, _i, _p]),
    "ivr_index_write": (_i, [_p, _i64, _p, _i64, _i, _p]),
    "ivr_index_write_ring": (_i, [_p, _p, _i64, _i, _p, _p]),
    "ivr_index_reconstruct": (_i, [_p, _i64, _i64, _p, _p]),
    "ivr_index_reserve_search": (_i, [_p, _i, _i]),
    "ivr_index_search": (_i, [_p, _p, _i, _i, _i, _i64, _p, _p, _p]),
    "ivr_topk_merge": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "ivr_topk_pack": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "ivr_topk_merge_packed": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "ivr_rowwise_cosine": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "ivr_dedup_keep_mask": (_i, [_p, _p, _i, _i, _f, _p, _p, _p]),
    "ivr_scene_keep_mask": (_i, [_p, _p, _i, _i, _f, _i, _p, _p]),
    "ivr_scene_keep_mask_window": (_i, [_p, _p, _i, _i, _f, _i, _p, _p]),
    "ivr_frame_quality_scratch_bytes": (_i64, [_i, _i, _i]),
    "ivr_frame_quality": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
}
EXPORTS = tuple(_SIGS)

_lib = None
_lock = threading.Lock()


class IvrError(RuntimeError):
    pass


def load():
    """Load libivr_hip.so; raises ImportError (never falls back) when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback")
            # torch must load its HIP runtime first: the process has to end up with ONE libamdhip64, the one
            # torch's allocator and streams live in (loading ours first leaves two runtimes and no visible device)
            import torch  # noqa: F401
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc, what=""):
    if rc == 0:
        return
    msg = (load().ivr_last_error(None) or b"").decode("utf-8", "replace")
    text = f"{what}: {msg}" if what else msg
    if rc == -1:
        raise ValueError(text)
    if rc == -3:
        raise MemoryError(text)
    raise IvrError(text)


_ctxs = {}


def context(device=0):
    """One ivr_ctx per device per process."""
    with _lock:
        ctx = _ctxs.get(device)
    if ctx is None:
        lib = load()
        h = _p()
        check(lib.ivr_init(int(device), C.byref(h)), "ivr_init")
        with _lock:
            ctx = _ctxs.setdefault(device, h)
    return ctx


def device_info(device=0):
    lib = load()
    cu, hbm = _i(), _i64()
    arch = C.create_string_buffer(64)
    check(lib.ivr_device_info(context(device), C.byref(cu), C.byref(hbm), arch, 64))
    return {"cu_count": cu.value, "hbm_bytes": hbm.value, "arch": arch.value.decode()}


def stream_ptr(stream=None):
    """hipStream_t of a torch stream (default: torch's current stream) as an integer for ctypes."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def f3(v):
    return (_f * 3)(*[float(x) for x in v])


def profile_enable(on=True, device=0):
    """on: False / True (level 1: the kernels that carry a step) / 2 (also the short search-tail and append launches)."""
    check(load().ivr_profile_enable(context(device), int(on)))


def profile_reset(device=0):
    check(load().ivr_profile_reset(context(device)))


def profile_read(device=0):
    """{"kernel": {"launches", "ms", "work"}} measured with HIP events on the launch stream."""
    import json
    buf = C.create_string_buffer(1 << 16)
    check(load().ivr_profile_json(context(device), buf, len(buf)))
    return json.loads(buf.value.decode())
