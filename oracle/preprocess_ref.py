"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU (NumPy, integer + float) restatement of the frame preprocessing the
reference delegates to PIL and the HuggingFace image processors:

  * `core.py:1613`  HFCLIPProcessor(images=...)  = convert-RGB -> resize shortest edge 224
    BICUBIC -> centre-crop 224x224 -> float32(float64(u8) * (1/255)) -> (v - mean) / std
    (transformers/image_processing_backends.py:521-640, image_transforms.py:89-124,
     :296-310, :408-442, :493-501; transformers 5.15.0, un-pinned by the reference)
  * `video_frame_filter.py:58-59`  cv2 BGR->RGB, Image.fromarray(img).resize((224, 224))
    = STRETCH resize with PIL's default filter BICUBIC, then the ViT processor's
    rescale + normalise (its own resize is then an identity).

PIL's resampler (Pillow 12.2 here; src/libImaging/Resample.c, an un-vendored
dependency) is restated from its published algorithm: separable two-pass
convolution, horizontal first, kernel support scaled by the downscale factor
(antialiasing), bicubic a = -0.5, coefficients normalised to sum 1 and quantised
to 22-bit fixed point, accumulation in int32 starting from 1 << 21, result
shifted right by 22 and clamped to [0, 255], with a uint8 intermediate image
between the passes.

Pinned by tests/golden/preprocess_*.npz, generated in the build container by
running PIL / CLIPImageProcessorPil themselves (tests/golden/make_golden.py).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}


def precompute_coeffs(in_size, out_size, filt="bicubic"):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box.

    Returns (xmin int32[out], xcnt int32[out], kk int32[out, ksize])."""
    fn, fsupport = FILTERS[filt]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int32)
    xcnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        wts = [fn((x + lo - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for w in wts:
            ww += w
        if ww != 0.0:
            wts = [w / ww for w in wts]
        for x, w in enumerate(wts):
            v = w * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
        xmin[xx] = lo
        xcnt[xx] = n
    return xmin, xcnt, kk


def _resample_axis(img, out_size, axis, filt):
    """One pass over `axis` of a uint8 [H,W,C] image: int32 accumulate, >>22, clamp."""
    in_size = img.shape[axis]
    xmin, xcnt, kk = precompute_coeffs(in_size, out_size, filt)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        n = int(xcnt[xx])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        acc += np.tensordot(kk[xx, :n].astype(np.int64), src[xmin[xx]:xmin[xx] + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize(img, out_w, out_h, filt="bicubic"):
    """PIL Image.resize((out_w,out_h), filt) on a uint8 HWC image (horizontal pass, then vertical)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        out = _resample_axis(out, out_w, 1, filt)
    if out_h != h:
        out = _resample_axis(out, out_h, 0, filt)
    return np.ascontiguousarray(out)


def shortest_edge_size(h, w, size=224):
    """image_transforms.py:296-310 (default_to_square=False): returns (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop(img, ch, cw):
    """image_transforms.py:493-501 (image at least as large as the crop)."""
    h, w = img.shape[:2]
    top = (h - ch) // 2
    left = (w - cw) // 2
    assert top >= 0 and left >= 0
    return img[top:top + ch, left:left + cw]


def letterbox(img, size=224, fill=0, filt="bicubic"):
    """Extra mode named by BASELINE.json (the reference itself never letterboxes):
    aspect-preserving resize of the LONG edge to `size`, centred on a `fill` canvas."""
    h, w = img.shape[:2]
    if h >= w:
        nh, nw = size, max(1, int(size * w / h))
    else:
        nh, nw = max(1, int(size * h / w)), size
    r = pil_resize(img, nw, nh, filt)
    out = np.full((size, size, img.shape[2]), fill, dtype=np.uint8)
    top, left = (size - nh) // 2, (size - nw) // 2
    out[top:top + nh, left:left + nw] = r
    return out


def geometry(img, mode, size=224, filt="bicubic"):
    """uint8 HWC -> uint8 [size,size,C] under one of the build's geometry modes."""
    if mode == "identity":
        assert img.shape[0] == size and img.shape[1] == size
        return img
    if mode == "stretch":                 # video_frame_filter.py:59
        return pil_resize(img, size, size, filt)
    if mode == "shortest_edge_crop":      # core.py:1613 (CLIP processor defaults)
        nh, nw = shortest_edge_size(img.shape[0], img.shape[1], size)
        return center_crop(pil_resize(img, nw, nh, filt), size, size)
    if mode == "letterbox":
        return letterbox(img, size, 0, filt)
    raise ValueError(mode)


def rescale_normalize(u8_hwc, mean, std, rescale=1.0 / 255.0):
    """image_transforms.py:118-122 then :417-439: float32(float64(u8)*rescale), (v-mean)/std in float32.
    Returns float32 CHW."""
    v = (u8_hwc.astype(np.float64) * rescale).astype(np.float32)
    m = np.array(mean, dtype=np.float32)
    s = np.array(std, dtype=np.float32)
    v = (v - m) / s
    return np.ascontiguousarray(v.transpose(2, 0, 1))


def value_lut(mean, std, rescale=1.0 / 255.0):
    """The 3x256 table of every value rescale_normalize can emit (float32)."""
    u = np.arange(256, dtype=np.float64)
    v = (u * rescale).astype(np.float32)
    m = np.array(mean, dtype=np.float32)
    s = np.array(std, dtype=np.float32)
    return np.ascontiguousarray(((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32))


def preprocess(frames_u8, mode, mean, std, bgr=False, size=224, filt="bicubic"):
    """List/array of uint8 HWC frames -> float32 [n,3,size,size]."""
    out = []
    for f in frames_u8:
        if bgr:                            # video_frame_filter.py:58 cv2.COLOR_BGR2RGB
            f = f[:, :, ::-1]
        out.append(rescale_normalize(geometry(np.ascontiguousarray(f), mode, size, filt), mean, std))
    return np.stack(out)


def to_bf16_bits(x):
    """float32 -> bf16 bit pattern (uint16), round-to-nearest-even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def patch_major(pix_nchw, patch):
    """[n,3,S,S] -> [n*g*g, 3*P*P] rows in conv-weight order (c, py, px): the im2col matrix of the
    stride-P patch convolution (modeling_clip.py:170-180)."""
    n, c, s, _ = pix_nchw.shape
    g = s // patch
    x = pix_nchw.reshape(n, c, g, patch, g, patch).transpose(0, 2, 4, 1, 3, 5)
    return np.ascontiguousarray(x.reshape(n * g * g, c * patch * patch))
