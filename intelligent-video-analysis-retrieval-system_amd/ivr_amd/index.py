"""FAISS-shaped flat inner-product index resident in MI355X HBM.

Stands in for the `faiss.IndexFlatIP` objects the reference creates at
`unified_index.py:1767` and `core.py:1208-1219`, and for `faiss.normalize_L2`
(`unified_index.py:1776`): same method names, same (D, I) contract -
float32 scores descending, int64 labels, -1 labels for unused slots - so the
call sites `index.add(x)`, `index.search(q, k)`, `index.ntotal`, `index.d`,
`index.is_trained`, `index.train(x)` run unchanged on this object.
All arithmetic happens in libivr_hip.so; numpy arrays are staged through
torch CUDA tensors, torch CUDA tensors are used in place.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi


def _dev_f32(x, device):
    """numpy / torch -> contiguous float32 CUDA tensor on `device` (a view when already there)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if not isinstance(x, torch.Tensor):
        raise ValueError("expected a numpy array or a torch tensor")
    return x.to(device=device, dtype=torch.float32, non_blocking=False).contiguous()


def normalize_L2(x):
    """faiss.normalize_L2(x): in-place row L2 normalisation of a float32 [n,d] array (zero rows stay zero)."""
    if isinstance(x, np.ndarray):
        if x.dtype != np.float32 or x.ndim != 2:
            raise ValueError("normalize_L2 expects a float32 [n,d] array")
        t = _dev_f32(x, torch.device("cuda", torch.cuda.current_device()))
        normalize_L2(t)
        x[...] = t.cpu().numpy()
        return
    if x.dtype != torch.float32 or x.dim() != 2 or not x.is_cuda or not x.is_contiguous():
        raise ValueError("normalize_L2 expects a contiguous float32 [n,d] CUDA tensor")
    lib = _ffi.load()
    with torch.cuda.device(x.device):
        _ffi.check(lib.ivr_l2_normalize(_ffi.context(x.device.index), C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1],
                                        None, _ffi.stream_ptr()), "ivr_l2_normalize")


def count_nonfinite_and_normalize(t):
    """In-place normalise a CUDA tensor and return how many input elements were NaN/Inf (N2 validation)."""
    lib = _ffi.load()
    flag = torch.zeros(1, dtype=torch.int32, device=t.device)
    with torch.cuda.device(t.device):
        _ffi.check(lib.ivr_l2_normalize(_ffi.context(t.device.index), C.c_void_p(t.data_ptr()), t.shape[0], t.shape[1],
                                        C.c_void_p(flag.data_ptr()), _ffi.stream_ptr()), "ivr_l2_normalize")
    return int(flag.item())


class FlatIPIndex:
    """Exact inner-product index (FAISS IndexFlatIP contract) on one GPU."""

    def __init__(self, d, capacity=0, device=None):
        self._lib = _ffi.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.d = int(d)
        self.is_trained = True
        self.metric_type = 0  # faiss.METRIC_INNER_PRODUCT
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_create(_ffi.context(self.device.index), self.d, int(capacity), C.byref(h)),
                       "ivr_index_create")
        self._h = h

    # -- FAISS surface ---------------------------------------------------------------------------
    @property
    def ntotal(self):
        return int(self._lib.ivr_index_ntotal(self._h))

    def scan_stats(self):
        """(has_bf16_scan_copy, queries of the last scan chunk that were redone by the exact float32 scan)."""
        out = (C.c_int * 2)()
        _ffi.check(self._lib.ivr_index_scan_stats(self._h, out), "ivr_index_scan_stats")
        return bool(out[0]), int(out[1])

    def train(self, x):  # core.py:817-820 calls train() when is_trained is False; flat indexes never need it
        return None

    def add(self, x, normalize=False, chunk_rows=1 << 20):
        """Append rows.  Host arrays are staged to HBM in chunks of `chunk_rows` so that a build of tens of millions of
        rows never needs a second full copy on either side (the reference adds 10k-row slices, unified_index.py:1770)."""
        if isinstance(x, np.ndarray) or (isinstance(x, torch.Tensor) and not x.is_cuda):
            n = len(x)
            if x.ndim != 2 or x.shape[1] != self.d:
                raise ValueError(f"add expects [n,{self.d}], got {tuple(x.shape)}")
            for i in range(0, n, chunk_rows):
                self._add_device(_dev_f32(x[i:i + chunk_rows], self.device), normalize)
            return
        self._add_device(_dev_f32(x, self.device), normalize)

    def _add_device(self, t, normalize):
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"add expects [n,{self.d}], got {tuple(t.shape)}")
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_add(self._h, C.c_void_p(t.data_ptr()), t.shape[0], int(bool(normalize)),
                                               _ffi.stream_ptr()), "ivr_index_add")
            torch.cuda.current_stream().synchronize()  # `t` may be a temporary staging copy

    def write(self, start, x, normalize=False):
        """Overwrite rows [start, start+n): ring-buffer maintenance for rolling indexes."""
        t = _dev_f32(x, self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_write(self._h, int(start), C.c_void_p(t.data_ptr()), t.shape[0],
                                                 int(bool(normalize)), _ffi.stream_ptr()), "ivr_index_write")
            torch.cuda.current_stream().synchronize()

    def write_device(self, start, rows, normalize=False):
        """Stream-ordered overwrite from a float32 CUDA tensor already on this device (no host sync)."""
        if not (isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous()
                and rows.dim() == 2 and rows.shape[1] == self.d):
            raise ValueError(f"write_device expects a contiguous float32 CUDA tensor [n,{self.d}]")
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_write(self._h, int(start), C.c_void_p(rows.data_ptr()), rows.shape[0],
                                                 int(bool(normalize)), _ffi.stream_ptr()), "ivr_index_write")

    def write_ring(self, rows, cursor, normalize=False):
        """Overwrite the rows at *cursor (int64 CUDA scalar tensor) and advance it, all stream-ordered (graph-capturable)."""
        if cursor.dtype != torch.int64 or not cursor.is_cuda or cursor.numel() != 1:
            raise ValueError("cursor must be a 1-element int64 CUDA tensor")
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_write_ring(self._h, C.c_void_p(rows.data_ptr()), rows.shape[0], int(bool(normalize)),
                                                      C.c_void_p(cursor.data_ptr()), _ffi.stream_ptr()), "ivr_index_write_ring")

    def search(self, x, k):
        """(D, I) numpy arrays, exactly like faiss: D float32 [nq,k] descending, I int64 [nq,k], -1 padded."""
        q = np.asarray(x) if not isinstance(x, torch.Tensor) else x
        if isinstance(q, np.ndarray) and q.ndim == 1:
            q = q.reshape(1, -1)
        D, I = self.search_device(q, k)
        return D.cpu().numpy(), I.cpu().numpy()

    def search_device(self, x, k, normalize=False, id_base=0, out=None):
        """Device-resident variant: returns CUDA tensors and does not synchronise."""
        t = _dev_f32(x, self.device)
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"Query dimension ({tuple(t.shape)}) != index dimension ({self.d})")
        k = int(k)
        if k < 1 or k > _ffi.IVR_MAX_K:
            raise ValueError(f"k={k} outside [1,{_ffi.IVR_MAX_K}]")
        nq = t.shape[0]
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            D, I = out
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_search(self._h, C.c_void_p(t.data_ptr()), nq, k, int(bool(normalize)),
                                                  int(id_base), C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()),
                                                  _ffi.stream_ptr()), "ivr_index_search")
            if t.data_ptr() != (x.data_ptr() if isinstance(x, torch.Tensor) else 0):
                torch.cuda.current_stream().synchronize()  # staging copy must outlive the kernels
        return D, I

    def reserve_search(self, max_nq, max_k):
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_reserve_search(self._h, int(max_nq), int(max_k)), "ivr_index_reserve_search")

    def reconstruct_n(self, start=0, n=None):
        n = self.ntotal - start if n is None else n
        out = torch.empty((n, self.d), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_reconstruct(self._h, int(start), int(n), C.c_void_p(out.data_ptr()),
                                                       _ffi.stream_ptr()), "ivr_index_reconstruct")
        return out.cpu().numpy()

    def reset(self):
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_index_reset(self._h), "ivr_index_reset")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ivr_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def IndexFlatIP(d):
    """faiss.IndexFlatIP(d) drop-in constructor."""
    return FlatIPIndex(d)


def topk_merge(D_parts, I_parts, k=None):
    """Merge per-shard candidates [parts,nq,k] (CUDA tensors, global ids, parts in ascending id order)."""
    lib = _ffi.load()
    parts, nq, kk = D_parts.shape
    k = kk if k is None else k
    D = torch.empty((nq, k), dtype=torch.float32, device=D_parts.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=D_parts.device)
    if k != kk:
        raise ValueError("merge k must equal the per-shard k")
    with torch.cuda.device(D_parts.device):
        _ffi.check(lib.ivr_topk_merge(_ffi.context(D_parts.device.index), C.c_void_p(D_parts.contiguous().data_ptr()),
                                      C.c_void_p(I_parts.contiguous().data_ptr()), parts, nq, k,
                                      C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()), _ffi.stream_ptr()),
                   "ivr_topk_merge")
    return D, I


def topk_pack(D, I):
    """(D float32 [nq,k], I int64 [nq,k]) CUDA -> int32 [nq,k,3] (score bits, id lo, id hi): the wire format of the one all-gather."""
    lib = _ffi.load()
    nq, k = D.shape
    out = torch.empty((nq, k, 3), dtype=torch.int32, device=D.device)
    with torch.cuda.device(D.device):
        _ffi.check(lib.ivr_topk_pack(_ffi.context(D.device.index), C.c_void_p(D.contiguous().data_ptr()), C.c_void_p(I.contiguous().data_ptr()),
                                     nq, k, C.c_void_p(out.data_ptr()), _ffi.stream_ptr()), "ivr_topk_pack")
    return out


def topk_merge_packed(packed_parts):
    """Merge gathered candidates int32 [parts,nq,k,3] (parts in ascending id order) -> (D [nq,k], I [nq,k])."""
    lib = _ffi.load()
    parts, nq, k, _ = packed_parts.shape
    D = torch.empty((nq, k), dtype=torch.float32, device=packed_parts.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=packed_parts.device)
    with torch.cuda.device(packed_parts.device):
        _ffi.check(lib.ivr_topk_merge_packed(_ffi.context(packed_parts.device.index), C.c_void_p(packed_parts.data_ptr()), parts, nq, k,
                                             C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()), _ffi.stream_ptr()), "ivr_topk_merge_packed")
    return D, I
