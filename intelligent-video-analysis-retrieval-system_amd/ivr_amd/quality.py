"""Frame quality gating of the keyframe filter (filter.py:63-140) on the GPU.

  reference (filter.py)                         here
  --------------------------------------------  -----------------------------------------------------------
  calculate_blur_score(image_path)     :63      calculate_blur_score(image_path) / frame_quality_scores(frames)
  calculate_edge_density(image_path)   :76      calculate_edge_density(image_path)
  calculate_frame_quality_scores       :92      calculate_frame_quality_scores
  determine_adaptive_thresholds        :102     same name (np.percentile over the batch, host)
  is_frame_acceptable_adaptive / _fixed :117,:130   same names (host)

The two measures are OpenCV operators - Laplacian variance and the share of Canny edge pixels - computed per decoded frame by
ivr_frame_quality (csrc/quality.hip) for a whole batch of same-sized frames in HBM; decode stays on the CPU (PIL here, the
reference uses cv2.imread).  OpenCV is not installed in this environment: the operators follow their published definitions
and are checked against oracle/quality_ref.py, which is "parity unpinned" for the same reason.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi

CANNY_LOW, CANNY_HIGH = 20, 80            # filter.py:85


def frame_quality_scores(frames, bgr=False, canny_low=CANNY_LOW, canny_high=CANNY_HIGH):
    """frames: uint8 [n,h,w,3] (numpy or CUDA tensor; bgr=True for cv2-ordered pixels) ->
    [{'blur_score': float, 'edge_density': float}] per frame (filter.py:92-100)."""
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames))
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be uint8 [n,h,w,3]")
    frames = frames.cuda().contiguous() if not frames.is_cuda else frames.contiguous()
    if frames.data_ptr() % 16:                      # the tile kernel fetches 16-byte chunks: a view into a larger batch may start anywhere
        frames = frames.clone()
    n, h, w, _ = frames.shape
    if n == 0:
        return []
    lap = torch.empty((n, 2), dtype=torch.int64, device=frames.device)
    cnt = torch.empty(n, dtype=torch.int64, device=frames.device)
    with torch.cuda.device(frames.device):
        _ffi.check(_ffi.load().ivr_frame_quality(_ffi.context(frames.device.index), C.c_void_p(frames.data_ptr()), n, h, w, int(bool(bgr)),
                                                 int(canny_low), int(canny_high), C.c_void_p(lap.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                 _ffi.stream_ptr()), "ivr_frame_quality")
    lap, cnt = lap.cpu().numpy(), cnt.cpu().numpy()
    N = float(h * w)
    out = []
    for i in range(n):
        s1, s2 = int(lap[i, 0]), int(lap[i, 1])
        # exact integer sums -> population variance in float64 (numpy's .var() of the CV_64F response, ddof = 0)
        var = (s2 - s1 * s1 / N) / N
        out.append({"blur_score": float(var), "edge_density": float(cnt[i] / N * 100.0)})
    return out


def _decode(image_path):
    from PIL import Image
    return np.asarray(Image.open(image_path).convert("RGB"))


def calculate_frame_quality_scores(image_path):
    """filter.py:92; unreadable images score 0.0 / 0.0 as in the reference's try / except."""
    try:
        return frame_quality_scores(_decode(image_path)[None], bgr=False)[0]
    except Exception:
        return {"blur_score": 0.0, "edge_density": 0.0}


def calculate_blur_score(image_path):
    return calculate_frame_quality_scores(image_path)["blur_score"]


def calculate_edge_density(image_path):
    return calculate_frame_quality_scores(image_path)["edge_density"]


def quality_scores_for_paths(image_paths, batch=64):
    """The batched form the GPU wants: decode on the host, group by frame size, one launch chain per group."""
    scores = [None] * len(image_paths)
    by_shape = {}
    for i, p in enumerate(image_paths):
        try:
            a = _decode(p)
            by_shape.setdefault(a.shape, []).append((i, a))
        except Exception:
            scores[i] = {"blur_score": 0.0, "edge_density": 0.0}
    for items in by_shape.values():
        for j in range(0, len(items), batch):
            chunk = items[j:j + batch]
            for (i, _), sc in zip(chunk, frame_quality_scores(np.stack([a for _, a in chunk]))):
                scores[i] = sc
    return scores


def determine_adaptive_thresholds(all_quality_scores, config):
    """filter.py:102-115."""
    if not all_quality_scores:
        return None, None
    blur = [q["blur_score"] for q in all_quality_scores]
    edge = [q["edge_density"] for q in all_quality_scores]
    return np.percentile(blur, config["blur_percentile"]), np.percentile(edge, config["edge_percentile"])


def is_frame_acceptable_adaptive(quality_scores, blur_threshold, edge_threshold, config):
    """filter.py:117-128."""
    if config["enable_blur_detection"] and blur_threshold is not None and quality_scores["blur_score"] < blur_threshold:
        return False, "blur"
    if config["enable_edge_detection"] and edge_threshold is not None and quality_scores["edge_density"] < edge_threshold:
        return False, "low_edge"
    return True, "acceptable"


def is_frame_acceptable_fixed(quality_scores, config):
    """filter.py:130-140."""
    if config["enable_blur_detection"] and quality_scores["blur_score"] < config["blur_threshold"]:
        return False, "blur"
    if config["enable_edge_detection"] and quality_scores["edge_density"] < config["edge_threshold"]:
        return False, "low_edge"
    return True, "acceptable"
