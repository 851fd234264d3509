"""ORACLE (test infrastructure, never shipped, never imported by the product path).  PARITY UNPINNED.

CPU restatement (NumPy + scipy.ndimage) of the two OpenCV measures the reference's keyframe filter computes per frame at
filter.py:63-92:  cv2.Laplacian(gray, cv2.CV_64F).var()  and the share of  cv2.Canny(gray, 20, 80)  edge pixels.
The arithmetic lives in the un-vendored, un-pinned dependency opencv-python, which is NOT installed in this environment and
of which the reference holds no fixture: this file restates the operators from OpenCV's documented definitions
(imgproc color conversions: Y = 0.299 R + 0.587 G + 0.114 B in 14-bit fixed point for 8-bit images; Laplacian ksize = 1 =
the 3x3 aperture [0 1 0; 1 -4 1; 0 1 0] with BORDER_REFLECT_101; Canny = Sobel 3x3 (BORDER_REPLICATE), L1 gradient
magnitude, non-maximum suppression by direction sector with the fixed-point tan(22.5 deg) test, double threshold
(candidate: m > low, strong: m > high) and 8-connected hysteresis) and nothing here has been compared with cv2 output.
"""
import numpy as np
from scipy import ndimage

TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)


def to_gray(img, bgr=False):
    a = np.asarray(img, dtype=np.int64)
    r, g, b = (a[..., 2], a[..., 1], a[..., 0]) if bgr else (a[..., 0], a[..., 1], a[..., 2])
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def laplacian_var(gray):
    """filter.py:71: cv2.Laplacian(gray, cv2.CV_64F).var()."""
    g = np.pad(gray.astype(np.float64), 1, mode="reflect")            # numpy 'reflect' == BORDER_REFLECT_101
    lap = g[:-2, 1:-1] + g[2:, 1:-1] + g[1:-1, :-2] + g[1:-1, 2:] - 4.0 * g[1:-1, 1:-1]
    return float(lap.var())


def canny(gray, low=20, high=80):
    """filter.py:85: cv2.Canny(gray, 20, 80) (apertureSize 3, L2gradient False) -> uint8 edge map (0 / 255)."""
    g = np.pad(gray.astype(np.int64), 1, mode="edge")                 # BORDER_REPLICATE
    dx = (g[:-2, 2:] + 2 * g[1:-1, 2:] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[1:-1, :-2] + g[2:, :-2])
    dy = (g[2:, :-2] + 2 * g[2:, 1:-1] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[:-2, 1:-1] + g[:-2, 2:])
    ax, ay = np.abs(dx), np.abs(dy)
    mag = ax + ay
    mp = np.pad(mag, 1, mode="constant")                              # magnitudes outside the image count as 0
    c = mp[1:-1, 1:-1]
    left, right = mp[1:-1, :-2], mp[1:-1, 2:]
    up, down = mp[:-2, 1:-1], mp[2:, 1:-1]
    ul, ur, dl, dr = mp[:-2, :-2], mp[:-2, 2:], mp[2:, :-2], mp[2:, 2:]
    y = ay << 15
    tg22x = ax * TG22
    horiz = y < tg22x
    vert = ~horiz & (y > tg22x + (ax << 16))
    diag = ~horiz & ~vert
    neg = (dx ^ dy) < 0                                               # s = -1: compare with (y-1, x+1) and (y+1, x-1)
    peak = (horiz & (c > left) & (c >= right)) | (vert & (c > up) & (c >= down)) | \
           (diag & ~neg & (c > ul) & (c > dr)) | (diag & neg & (c > ur) & (c > dl))
    cand = peak & (mag > low)
    strong = cand & (mag > high)
    labels, n = ndimage.label(cand, structure=np.ones((3, 3), dtype=int))
    keep = np.zeros(n + 1, dtype=bool)
    keep[np.unique(labels[strong])] = True
    keep[0] = False
    return (keep[labels].astype(np.uint8)) * 255


def quality_scores(img, bgr=False, low=20, high=80):
    """filter.py:92-100 for one decoded frame."""
    gray = to_gray(img, bgr)
    edges = canny(gray, low, high)
    return {"blur_score": laplacian_var(gray), "edge_density": float(np.sum(edges > 0) / (edges.shape[0] * edges.shape[1]) * 100)}
