"""GPU parity of the fused QKV-projection + attention kernel (T <= 64, bf16): it must reproduce the unfused path
(QKV GEMM -> bf16 buffer -> attention kernel) BIT FOR BIT - same MFMA sequences, same roundings - and both must sit within the
bf16 tolerance of the float32 oracle.  Sequence lengths 10 .. 50 (2 .. 15 whole images per 256-row tile), ragged last tiles,
e4m3 attention output."""
import numpy as np
import pytest
import torch

from conftest import synth_frames
from ivr_amd import config as C
from ivr_amd.config import TowerConfig
from ivr_amd.weights import make_weights
from oracle import preprocess_ref as P
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu


def _run(cfg, w, frames, monkeypatch, fused, pers="0", **kw):
    """fused: the QKV + attention kernel instead of GEMM + attention; pers: "2" = its persistent form (round 3: one workgroup per CU,
    the next item's first K stage fetched during the attention phase) whatever the size, "0" = one tile per workgroup."""
    from ivr_amd.tower import Tower
    monkeypatch.setenv("IVR_FUSED_QKV", "1" if fused else "0")
    monkeypatch.setenv("IVR_QKV_PERS", pers)
    out = Tower(cfg, w, max_batch=len(frames), **kw).encode_frames(frames, "identity", C.CLIP_MEAN, C.CLIP_STD)
    return out.cpu().numpy()


@pytest.mark.parametrize("grid,n", [(3, 29), (4, 31), (5, 21), (6, 13), (7, 37), (7, 5), (7, 1)], ids=lambda v: str(v))
def test_fused_equals_unfused_bitwise(grid, n, monkeypatch):
    patch = 16
    cfg = TowerConfig(f"fused-g{grid}", "vision", 192, 2, 3, 384, grid * grid + 1, 64, image=patch * grid, patch=patch)
    w = make_weights(cfg, 40 + grid)
    frames = synth_frames(500 + grid, n, cfg.image, cfg.image)
    a = _run(cfg, w, frames, monkeypatch, True)
    b = _run(cfg, w, frames, monkeypatch, False)
    assert np.array_equal(a, b), np.abs(a - b).max()
    p = _run(cfg, w, frames, monkeypatch, True, pers="2")
    assert np.array_equal(p, b), np.abs(p - b).max()
    ref = V.vision_forward(cfg, w, P.preprocess(frames, "identity", C.CLIP_MEAN, C.CLIP_STD, size=cfg.image))
    cos = (a * ref).sum(1)
    assert cos.min() > 1 - 1e-4, cos.min()


def test_fused_vit_b32_and_e4m3_attention_output(monkeypatch):
    cfg = C.CLIP_VIT_B32
    w = make_weights(cfg, 12)
    frames = synth_frames(4321, 23, 224, 224)            # 4 full tiles of 5 images + 3
    a = _run(cfg, w, frames, monkeypatch, True)
    b = _run(cfg, w, frames, monkeypatch, False)
    assert np.array_equal(a, b)
    ref = V.vision_forward(cfg, w, P.preprocess(frames[:6], "identity", C.CLIP_MEAN, C.CLIP_STD))
    assert ((a[:6] * ref).sum(1) > 1 - 1e-4).all()
    assert np.array_equal(_run(cfg, w, frames, monkeypatch, True, pers="2"), b)
    a8 = _run(cfg, w, frames, monkeypatch, True, compute="fp8_all", fp8_sites=("o",))
    b8 = _run(cfg, w, frames, monkeypatch, False, compute="fp8_all", fp8_sites=("o",))
    assert np.array_equal(a8, b8)
    assert np.array_equal(_run(cfg, w, frames, monkeypatch, True, pers="2", compute="fp8_all", fp8_sites=("o",)), b8)
    assert ((a8[:6] * ref).sum(1) > 0.995).all()


def test_default_switches_on_for_large_batches(monkeypatch):
    """Without the override the fused kernel takes over once the grid fills the chip (>= 256 workgroups): same bits."""
    from ivr_amd import _ffi
    cfg = C.CLIP_VIT_B32
    w = make_weights(cfg, 12)
    frames = synth_frames(99, 128, 224, 224)             # 26 tiles x 12 heads = 312 workgroups
    monkeypatch.delenv("IVR_FUSED_QKV", raising=False)
    from ivr_amd.tower import Tower
    tw = Tower(cfg, w, max_batch=128)
    _ffi.profile_reset()
    _ffi.profile_enable(True)
    a = tw.encode_frames(frames).cpu().numpy()
    _ffi.profile_enable(False)
    prof = _ffi.profile_read()
    assert "gemm_qkv_attention" in prof and "attention" not in prof and "gemm_qkv" not in prof
    b = _run(cfg, w, frames, monkeypatch, False)
    assert np.array_equal(a, b)


def test_persistent_form_walks_many_items_per_workgroup(monkeypatch):
    """1,300 frames of ViT-B/32 = 260 tiles x 12 heads = 3,120 items over 256 persistent workgroups (12 - 13 items each, both zigzag
    directions inside the tower): bit-identical to the one-tile-per-workgroup kernel and to the unfused path, ragged last tile included."""
    cfg = C.CLIP_VIT_B32
    w = make_weights(cfg, 12)
    frames = synth_frames(77, 1303, 224, 224)
    monkeypatch.delenv("IVR_ZIGZAG", raising=False)
    p = _run(cfg, w, frames, monkeypatch, True, pers="1")
    t = _run(cfg, w, frames, monkeypatch, True, pers="0")
    assert np.array_equal(p, t)
    u = _run(cfg, w, frames[:300], monkeypatch, False)
    assert np.array_equal(p[:300], u)
