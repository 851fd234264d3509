"""Adjacent consumers of the embed + cosine kernels.

  * filter.py:142-222  calculate_similarities / detect_scene_transitions / group_into_scenes /
                       filter_similar_frames_in_scene  (scene cuts at cosine < 0.75, in-scene dedup at >= 0.95)
  * filter.py:224-258  filter_similar_frames_advanced (window variant), :260-315 apply_similarity_filtering_to_scenes
  * filter.py:317-470  filter_transition_frames_for_video -> filter_keyframes (quality for all frames in batches -> thresholds ->
                       only the accepted frames embedded, batched -> scene split -> per-scene filter)
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


def filter_similar_frames_advanced(scene_embeddings, scene_indices, config):
    """filter.py:224: keep a frame unless a KEPT frame among its `similarity_window_size` predecessors has cosine >= threshold.
    Two launches per scene (ivr_scene_keep_mask_window: banded cosines in parallel, then the decision chain) and one copy of the
    mask back."""
    n = len(scene_embeddings)
    if not config["enable_similarity_filtering"] or n <= 1:
        return scene_indices
    thr, window = float(config["similarity_threshold"]), max(1, int(config["similarity_window_size"]))
    e = _dev(np.asarray(scene_embeddings) if not isinstance(scene_embeddings, torch.Tensor) else scene_embeddings)
    keep = torch.empty(n, dtype=torch.uint8, device=e.device)
    with torch.cuda.device(e.device):
        _ffi.check(_ffi.load().ivr_scene_keep_mask_window(_ffi.context(e.device.index), C.c_void_p(e.data_ptr()), n, e.shape[1], thr, window,
                                                          C.c_void_p(keep.data_ptr()), _ffi.stream_ptr()), "ivr_scene_keep_mask_window")
    return [scene_indices[j] for j in np.nonzero(keep.cpu().numpy())[0].tolist()]


def apply_similarity_filtering_to_scenes(embeddings, valid_rows, scenes, config):
    """filter.py:260-315: the in-scene filter over every scene; embeddings may stay a CUDA tensor (scenes are slices of it)."""
    if not config["enable_similarity_filtering"]:
        idx = [i for a, b in scenes for i in range(a, b + 1)]
        return [embeddings[i] for i in idx], [valid_rows[i] for i in idx], {"original": len(idx), "filtered": len(idx), "removed": 0}
    kept_all, stats = [], {"original": 0, "filtered": 0, "removed": 0}
    advanced = config.get("use_advanced_similarity_filtering", False)
    for a, b in scenes:
        idx = list(range(a, b + 1))
        stats["original"] += len(idx)
        kept = (filter_similar_frames_advanced if advanced else filter_similar_frames_in_scene)(embeddings[a:b + 1], idx, config)
        kept_all.extend(kept)
        stats["filtered"] += len(kept)
    stats["removed"] = stats["original"] - stats["filtered"]
    return [embeddings[i] for i in kept_all], [valid_rows[i] for i in kept_all], stats


DEFAULT_FILTER_CONFIG = {                       # filter.py:13-36
    "transition_threshold": 0.75, "min_scene_length": 2, "blur_percentile": 10.0, "edge_percentile": 10.0,
    "enable_adaptive_filtering": True, "blur_threshold": 10.0, "edge_threshold": 5.0, "enable_blur_detection": True,
    "enable_edge_detection": True, "enable_similarity_filtering": True, "similarity_threshold": 0.95, "min_frame_distance": 1,
    "similarity_window_size": 5, "use_advanced_similarity_filtering": False,
}


def filter_keyframes(frames, embed_batch, config=None, rows=None, quality_batch=64):
    """The keyframe filter of filter_transition_frames_for_video (filter.py:317-470) as batched GPU passes, without its file I/O:
      phase 1  quality scores of ALL frames (ivr_frame_quality over batches of same-sized frames)            filter.py:352-364
      phase 2  thresholds: percentiles of the batch (adaptive) or the fixed ones                               filter.py:381-394
      phase 3  acceptance; ONLY the accepted frames are embedded, in batches (embed_batch(list of frames) ->   filter.py:396-418
               [m, d] array / CUDA tensor, one call per quality_batch accepted frames)
      phase 4  consecutive cosines (one launch), scene cuts, scenes                                            filter.py:435-450
      phase 5  in-scene similarity filter per scene, window variant when use_advanced_similarity_filtering     filter.py:452-455
    frames: list of uint8 RGB arrays [h,w,3] (None = unreadable: scores 0.0 / 0.0 as the reference's try / except).  rows: the
    per-frame records to carry through (default: the positions).  Returns None where the reference returns None, else a dict with
    'kept' (positions in input order), 'rows', 'embeddings' (CUDA tensor of the kept frames) and the reference's statistics."""
    from . import quality as Q
    cfg = dict(DEFAULT_FILTER_CONFIG)
    cfg.update(config or {})
    n = len(frames)
    rows = list(range(n)) if rows is None else list(rows)
    if n == 0:
        return None
    scores = [None] * n
    by_shape = {}
    for i, f in enumerate(frames):
        if f is None:
            scores[i] = {"blur_score": 0.0, "edge_density": 0.0}
        else:
            by_shape.setdefault(tuple(f.shape), []).append(i)
    for idxs in by_shape.values():
        for j in range(0, len(idxs), quality_batch):
            chunk = idxs[j:j + quality_batch]
            for i, sc in zip(chunk, Q.frame_quality_scores(np.stack([frames[i] for i in chunk]))):
                scores[i] = sc
    if cfg["enable_adaptive_filtering"]:
        bt, et = Q.determine_adaptive_thresholds(scores, cfg)
    else:
        bt, et = cfg["blur_threshold"], cfg["edge_threshold"]
    qstats = {"blur": 0, "low_edge": 0, "acceptable": 0, "embedding_error": 0}
    accepted = []
    for i, sc in enumerate(scores):
        ok, reason = (Q.is_frame_acceptable_adaptive(sc, bt, et, cfg) if cfg["enable_adaptive_filtering"]
                      else Q.is_frame_acceptable_fixed(sc, cfg))
        if ok and frames[i] is None:
            ok, reason = False, "embedding_error"              # extract_embedding's try / except returns None (filter.py:50-61)
        if ok:
            accepted.append(i)
        else:
            qstats[reason] += 1
    qstats["acceptable"] = len(accepted)
    if len(accepted) < cfg["min_scene_length"]:
        return None
    parts = [_dev(embed_batch([frames[i] for i in accepted[j:j + quality_batch]])) for j in range(0, len(accepted), quality_batch)]
    emb = torch.cat(parts)
    sims = calculate_similarities(emb)
    transitions = detect_scene_transitions(sims, cfg["transition_threshold"])
    scenes = group_into_scenes(transitions, len(accepted), cfg["min_scene_length"])
    if not scenes:
        return None
    local = list(range(len(accepted)))
    _, kept_local, sstats = apply_similarity_filtering_to_scenes(emb, local, scenes, cfg)
    return {"kept": [accepted[i] for i in kept_local], "rows": [rows[accepted[i]] for i in kept_local],
            "embeddings": emb[torch.tensor(kept_local, dtype=torch.long, device=emb.device)], "quality_scores": scores,
            "quality_stats": qstats, "similarity_stats": sstats, "scenes": scenes, "transitions": transitions,
            "avg_similarity": float(np.mean(sims)) if sims else float("nan"), "blur_threshold": bt, "edge_threshold": et}


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
