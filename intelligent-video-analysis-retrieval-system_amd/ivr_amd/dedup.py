"""Device-side near-duplicate filter: the keep/drop rule of video_frame_filter.py:63-70."""
import ctypes as C

import torch

from . import _ffi


class DedupState:
    """Carries the embedding of the last KEPT frame across batches of one video (prev_embedding,
    video_frame_filter.py:39,70)."""

    def __init__(self, d, device=None):
        self._lib = _ffi.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.d = int(d)
        self.state = torch.zeros(self.d + 1, dtype=torch.float32, device=self.device)

    def reset(self):
        self.state.zero_()

    def keep_mask(self, emb, threshold=0.98):
        """emb float32 CUDA [n,d] in frame order -> uint8 CUDA [n]; 1 = unique frame (cos < threshold)."""
        if emb.dim() != 2 or emb.shape[1] != self.d or emb.dtype != torch.float32 or not emb.is_cuda:
            raise ValueError(f"emb must be a float32 CUDA tensor [n,{self.d}]")
        emb = emb.contiguous()
        keep = torch.empty(emb.shape[0], dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.ivr_dedup_keep_mask(_ffi.context(self.device.index), C.c_void_p(emb.data_ptr()),
                                                     emb.shape[0], self.d, float(threshold),
                                                     C.c_void_p(self.state.data_ptr()), C.c_void_p(keep.data_ptr()),
                                                     _ffi.stream_ptr()), "ivr_dedup_keep_mask")
        return keep
