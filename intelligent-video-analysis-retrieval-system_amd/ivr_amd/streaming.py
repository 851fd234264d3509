"""Streaming step for live feeds (BASELINE config 4): F feeds x fps frames per step go through
HIP resize+normalise -> tower -> rolling-window index overwrite -> cosine top-k of a fixed query batch,
captured once as a HIP graph and replayed per step.

Everything the step touches is a static device buffer (frames in, (D, I) out) and the ring cursor lives in HBM, so
the captured graph contains no host-dependent argument.  The reference has no streaming mode; this is the batched,
graph-launched form of its per-frame loop (`video_frame_filter.py:53-85`) feeding `search_vectors` (`unified_index.py:480`).

Several GPUs: every feed is pinned to one GPU.  Rank r embeds its own feeds' frames, overwrites its OWN ring
(the rolling window is row-sharded: rank r holds global ids [r*W, (r+1)*W), W = window rows per rank), searches it with the
replicated query batch, and the step ends with the same single all-gather + merge as the static sharded search
(`sharded.ShardedIndex.exchange`).  The captured graph holds this rank's kernels only; the collective and the merge kernel are
issued on the same stream right after the replay - RCCL calls are stream-ordered but are not captured here (a captured collective
pins communicator state into the graph; keeping it outside lets one graph serve any world size and the gloo rehearsal).
"""
import torch

from .config import CLIP_MEAN, CLIP_STD
from .preprocess import preprocess_frames


class StreamingSession:
    def __init__(self, tower, index, frames_per_step, height, width, queries, k=10, mode="stretch", bgr=True,
                 mean=CLIP_MEAN, std=CLIP_STD, normalize_queries=True, use_graph=True, sharded=None, preprocess=preprocess_frames):
        """sharded: a `sharded.ShardedIndex` wrapping `index` (this rank's ring) - the step then returns the merged result over
        every rank's ring, global ids = sharded.id_base + ring position.  preprocess: the resize + normalise entry point (the HIP
        kernel; the world_size-2 CPU test plugs in a stand-in together with tower / index doubles)."""
        if index.ntotal % frames_per_step:
            raise ValueError("the rolling index size must be a multiple of frames_per_step")
        if frames_per_step > tower.max_batch:
            raise ValueError("frames_per_step exceeds the tower's max_batch")
        if sharded is not None and sharded.local is not index:
            raise ValueError("sharded must wrap this session's own index")
        self.tower, self.index, self.n = tower, index, int(frames_per_step)
        self.sharded, self._preprocess = sharded, preprocess
        dev = tower.device
        cfg = tower.cfg
        self.mode, self.bgr, self.mean, self.std, self.k = mode, bgr, mean, std, int(k)
        self.normalize_queries = normalize_queries
        self.frames = torch.zeros((self.n, height, width, 3), dtype=torch.uint8, device=dev)
        g = cfg.image // cfg.patch
        kpad = -(-3 * cfg.patch * cfg.patch // 64) * 64
        self.patches = torch.empty((self.n * g * g, kpad), dtype=tower.act_dtype, device=dev)
        self.emb = torch.empty((self.n, tower.embed_dim), dtype=torch.float32, device=dev)
        self.queries = queries.to(device=dev, dtype=torch.float32).contiguous()
        self.D = torch.empty((self.queries.shape[0], self.k), dtype=torch.float32, device=dev)
        self.I = torch.empty((self.queries.shape[0], self.k), dtype=torch.int64, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self.index.reserve_search(self.queries.shape[0], self.k)
        self.graph = None
        if use_graph:
            # warm-up on the session's own stream (lazy allocations, function attributes, lookup tables, and the per-stream
            # preprocess scratch, which cannot be allocated during capture), then capture on that SAME stream
            cur0 = self.cursor.clone()
            s = self.stream = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                self._enqueue()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.cursor.copy_(cur0)          # the warm-up wrote zero frames into slot 0; the first real step overwrites it
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self._enqueue()
            self.cursor.copy_(cur0)

    def _enqueue(self):
        cfg = self.tower.cfg
        self._preprocess(self.frames, self.mode, self.mean, self.std, bgr=self.bgr, size=cfg.image, patch=cfg.patch,
                         out_dtype=self.tower.act_dtype, out=self.patches)
        self.tower.encode_patches(self.patches, self.n, normalize=True, out=self.emb)
        self.index.write_ring(self.emb, self.cursor)
        self.index.search_device(self.queries, self.k, normalize=self.normalize_queries, out=(self.D, self.I),
                                 id_base=self.sharded.id_base if self.sharded is not None else 0)

    def close(self):
        """Drop the captured graph and hand the session stream's scratch block back to the library (ivr_release_stream_scratch):
        scratch is per stream and grow-only, so short-lived sessions would otherwise each leave a block behind until ivr_destroy."""
        stream = getattr(self, "stream", None)
        self.graph = None
        if stream is not None and self.tower.device.type == "cuda":
            from . import _ffi
            import ctypes as C
            _ffi.check(_ffi.load().ivr_release_stream_scratch(_ffi.context(self.tower.device.index), C.c_void_p(stream.cuda_stream)),
                       "ivr_release_stream_scratch")
            self.stream = None

    def step(self, frames=None):
        """frames: uint8 [n,h,w,3] (CUDA or pinned host) for this step, or None to reuse the buffer.  Returns (D, I)
        device tensors that the next step overwrites."""
        if frames is not None:
            self.frames.copy_(frames, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
        if self.sharded is not None and self.sharded.world > 1:
            # the one exchange step: all-gather of every rank's (score, global id) candidates + merge, on the current stream
            return self.sharded.exchange(self.D, self.I)
        return self.D, self.I
