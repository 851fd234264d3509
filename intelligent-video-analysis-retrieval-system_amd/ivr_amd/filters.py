"""Adjacent consumers of the embed + cosine kernels.

  * filter.py:142-222  calculate_similarities / detect_scene_transitions / group_into_scenes /
                       filter_similar_frames_in_scene  (scene cuts at cosine < 0.75, in-scene dedup at >= 0.95)
  * core.py:3493-3531  MetadataManager._build_similarity_relationships: per folder, each frame's 10 most similar
                       frames (excluding itself) with cosine > 0.7

The cosines come from the HIP kernels (ivr_rowwise_cosine, the flat index self-search); the list bookkeeping is the
reference's, kept on the host.  Blur / edge-density gating (filter.py:63-92): ivr_amd/quality.py.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .index import FlatIPIndex


def _dev(x):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.to(device="cuda", dtype=torch.float32).contiguous()


def rowwise_cosine(a, b):
    """cos(a[i], b[i]) -> float32 CUDA [n]."""
    a, b = _dev(a), _dev(b)
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError("rowwise_cosine expects two [n,d] arrays of the same shape")
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _ffi.check(_ffi.load().ivr_rowwise_cosine(_ffi.context(a.device.index), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                                  a.shape[0], a.shape[1], C.c_void_p(out.data_ptr()), _ffi.stream_ptr()),
                   "ivr_rowwise_cosine")
    return out


def calculate_similarities(embeddings):
    """filter.py:142: cosine between consecutive frames (list of length n-1)."""
    e = _dev(np.asarray(embeddings) if not isinstance(embeddings, torch.Tensor) else embeddings)
    if e.shape[0] < 2:
        return []
    return rowwise_cosine(e[:-1], e[1:]).cpu().numpy().tolist()


def detect_scene_transitions(similarities, threshold):
    """filter.py:152."""
    return [i + 1 for i, s in enumerate(similarities) if s < threshold]


def group_into_scenes(transition_points, total_frames, min_length):
    """filter.py:160."""
    scenes, start = [], 0
    for t in transition_points:
        if t - start >= min_length:
            scenes.append((start, t - 1))
        start = t
    if total_frames - start >= min_length:
        scenes.append((start, total_frames - 1))
    return scenes


def filter_similar_frames_in_scene(scene_embeddings, scene_indices, config):
    """filter.py:178: keep the first frame, then a frame at least `min_frame_distance` after the last kept one whose
    cosine to it is below `similarity_threshold`; always keep the scene's last frame.  The whole keep / drop chain of the
    scene is ONE kernel launch (ivr_scene_keep_mask: the sequential chain of dedup.hip with the distance rule added) and
    one copy of the mask back - not a launch and a sync per kept frame."""
    n = len(scene_embeddings)
    if not config["enable_similarity_filtering"] or n <= 1:
        return scene_indices
    thr, min_dist = float(config["similarity_threshold"]), max(1, int(config["min_frame_distance"]))
    e = _dev(np.asarray(scene_embeddings) if not isinstance(scene_embeddings, torch.Tensor) else scene_embeddings)
    keep = torch.empty(n, dtype=torch.uint8, device=e.device)
    with torch.cuda.device(e.device):
        _ffi.check(_ffi.load().ivr_scene_keep_mask(_ffi.context(e.device.index), C.c_void_p(e.data_ptr()), n, e.shape[1], thr, min_dist,
                                                   C.c_void_p(keep.data_ptr()), _ffi.stream_ptr()), "ivr_scene_keep_mask")
    kept = np.nonzero(keep.cpu().numpy())[0].tolist()
    if kept[-1] != n - 1:
        kept.append(n - 1)
    return [scene_indices[j] for j in kept]


def similarity_graph(features, keys, top=10, threshold=0.7):
    """core.py:3493-3531 for one folder: {key: [keys of the `top` most similar other frames with cosine > threshold]}.
    The reference sorts a full N x N matrix; here the folder is its own flat index and every row is a query."""
    f = np.asarray(features, dtype=np.float32)
    n = len(f)
    if n < 2:
        return {}
    idx = FlatIPIndex(f.shape[1], capacity=n)
    idx.add(f, normalize=True)
    k = min(top + 1, n)
    D, I = idx.search_device(_dev(f), k, normalize=True)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    idx.close()
    out = {}
    for i, key in enumerate(keys):
        # np.argsort(sim[i])[::-1][1:11]: drop the first entry of the descending order (the frame itself)
        out[key] = [keys[j] for d, j in zip(D[i][1:], I[i][1:]) if j >= 0 and d > threshold]
    return out
