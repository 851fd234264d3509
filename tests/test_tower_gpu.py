"""GPU parity: HIP encoder towers (through the C ABI) vs the CPU oracle and the committed HF vectors.

Tolerances (written here, as BASELINE.json's north_star asks: cosine scores within 1e-3 of the fp32 CPU path):
  * IVR_COMPUTE_F32 (f32 MFMA, verification mode): |embedding - oracle| <= 2e-5 per component
  * IVR_COMPUTE_BF16 (production): cosine(embedding, oracle) >= 1 - 1e-3, i.e. any cosine score computed from
    these embeddings moves by < 1e-3 + second-order terms; the measured value is printed.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, synth_frames
from ivr_amd import config as C
from ivr_amd.weights import make_weights
from oracle import preprocess_ref as P
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu
META = {e["tower"]: e for e in json.load(open(os.path.join(GOLDEN, "golden.json")))["towers"]}


def _cos(a, b):
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def _vision(cfg, compute, n, max_batch=None):
    from ivr_amd.tower import Tower
    e = META[cfg.name]
    w = make_weights(cfg, e["weight_seed"])
    mean, std = (C.IMAGENET_MEAN, C.IMAGENET_STD) if cfg is C.DINO_VIT_S16 else (C.CLIP_MEAN, C.CLIP_STD)
    frames = synth_frames(e["frame_seed"], e["n"], 224, 224)[:n]
    tw = Tower(cfg, w, max_batch=max_batch or n, compute=compute)
    out = tw.encode_frames(frames, "identity", mean, std).cpu().numpy()
    return tw, w, frames, out


def test_tiny_f32_every_layer(golden):
    """Bring-up shape: residual stream after every block against the oracle dump."""
    from ivr_amd.preprocess import preprocess_frames
    g = golden("towers")
    cfg = C.TINY_VIT
    tw, w, frames, out = _vision(cfg, "f32", 4)
    assert np.abs(out - g[cfg.name + "_emb"]).max() < 2e-5
    px = preprocess_frames(frames, "identity", patch=cfg.patch, out_dtype=torch.float32)
    for layer in range(cfg.layers + 1):
        _, hid = tw.encode_patches(px, 4, capture_hidden=layer)
        ref = g[f"{cfg.name}_hidden{layer}"]
        err = np.abs(hid.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err < 1e-5, (layer, err)


@pytest.mark.parametrize("cfg,n", [(C.TINY_VIT, 4), (C.CLIP_VIT_B32, 8), (C.DINO_VIT_S16, 2), (C.CLIP_VIT_L14, 2)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_vision_f32_matches_hf_golden(cfg, n, golden):
    g = golden("towers")
    _, _, _, out = _vision(cfg, "f32", n)
    err = np.abs(out - g[cfg.name + "_emb"][:n]).max()
    print(f"{cfg.name} f32 max|d|={err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("cfg,n", [(C.TINY_VIT, 4), (C.CLIP_VIT_B32, 8), (C.DINO_VIT_S16, 2), (C.CLIP_VIT_L14, 2)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_vision_bf16_within_cosine_tolerance(cfg, n, golden):
    g = golden("towers")
    _, _, _, out = _vision(cfg, "bf16", n)
    ref = g[cfg.name + "_emb"][:n]
    cos = _cos(out, ref)
    print(f"{cfg.name} bf16 min cos={cos.min():.6f} max|d|={np.abs(out - ref).max():.2e}")
    assert cos.min() > 1 - 1e-4      # measured 1 - 1.6e-5; the north-star bound is 1e-3
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5     # F.normalize applied


@pytest.mark.parametrize("grid,patch,head_kernel", [(8, 4, 1), (10, 8, 1), (14, 8, 1), (16, 4, 1), (24, 4, 1), (28, 4, 1), (16, 4, 0)])
def test_attention_sequence_lengths(grid, patch, head_kernel, monkeypatch):
    """T = grid^2 + 1 tokens: 65 (first length past the one-block kernel), 101, 197 (DINO), 257 (ViT-L/14), 577 (ViT-L/14@336:
    query split over 3 workgroups, 148 KB of LDS), 785 (past the LDS limit: generic flash kernel), and 257 again with the
    head-resident kernel switched off.  A two-layer width-128 tower on small patches, bf16 vs the float32 oracle."""
    from ivr_amd.config import TowerConfig
    from ivr_amd.tower import Tower
    monkeypatch.setenv("IVR_ATTN_HEAD", str(head_kernel))
    cfg = TowerConfig(f"tiny-g{grid}", "vision", 128, 2, 2, 256, grid * grid + 1, 64, image=patch * grid, patch=patch)
    w = make_weights(cfg, 21)
    frames = synth_frames(77, 3, cfg.image, cfg.image)
    out = Tower(cfg, w, max_batch=3).encode_frames(frames, "identity", C.CLIP_MEAN, C.CLIP_STD).cpu().numpy()
    ref = np.asarray(V.vision_forward(cfg, w, P.preprocess(frames, "identity", C.CLIP_MEAN, C.CLIP_STD, size=cfg.image)))
    cos = _cos(out, ref)
    print(f"T={cfg.tokens} head_kernel={head_kernel} min cos={cos.min():.6f}")
    assert cos.min() > 1 - 1e-4


def test_batching_is_row_independent():
    """Ragged batch sizes and max_batch chunking do not change any row (bit-exact)."""
    cfg = C.CLIP_VIT_B32
    from ivr_amd.tower import Tower
    w = make_weights(cfg, 12)
    frames = synth_frames(4321, 37, 224, 224)
    a = Tower(cfg, w, max_batch=37).encode_frames(frames).cpu().numpy()
    b = Tower(cfg, w, max_batch=16).encode_frames(frames).cpu().numpy()     # 16 + 16 + 5
    assert np.array_equal(a, b)
    ref = V.vision_forward(cfg, w, P.preprocess(frames[:3], "identity", C.CLIP_MEAN, C.CLIP_STD))
    assert _cos(a[:3], ref).min() > 1 - 1e-3


def test_unnormalised_output_and_resize_path():
    cfg = C.TINY_VIT
    from ivr_amd.tower import Tower
    w = make_weights(cfg, 11)
    frames = synth_frames(9, 3, 300, 400)
    tw = Tower(cfg, w, max_batch=8, compute="f32")
    out = tw.encode_frames(frames, "shortest_edge_crop", normalize=False).cpu().numpy()
    ref = V.vision_forward(cfg, w, P.preprocess(frames, "shortest_edge_crop", C.CLIP_MEAN, C.CLIP_STD), normalize=False)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.parametrize("cfg", [C.TINY_TEXT, C.CLIP_TEXT_B32], ids=lambda c: c.name)
@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_text_tower(cfg, compute, golden):
    from ivr_amd.tower import Tower
    g = golden("towers")
    w = make_weights(cfg, META[cfg.name]["weight_seed"])
    ids = g[cfg.name + "_ids"]
    out = Tower(cfg, w, max_batch=8, compute=compute).encode_ids(ids).cpu().numpy()
    ref = g[cfg.name + "_emb"]
    if compute == "f32":
        assert np.abs(out - ref).max() < 2e-5
    else:
        assert _cos(out, ref).min() > 1 - 1e-3
    # shorter padded length: same rows, truncated after the EOS
    T = int((ids == cfg.eos_id).argmax(1).max()) + 1
    out2 = Tower(cfg, w, max_batch=8, compute="f32").encode_ids(ids[:, :T]).cpu().numpy()
    assert np.abs(out2 - ref).max() < 2e-5


def test_tower_errors():
    from ivr_amd.tower import Tower
    cfg = C.TINY_VIT
    w = make_weights(cfg, 11)
    bad = dict(w)
    bad.pop("cls")
    with pytest.raises(RuntimeError):
        Tower(cfg, bad)                      # missing tensor -> IVR_ERR_STATE
    bad = dict(w)
    bad["cls"] = np.zeros(3, np.float32)
    with pytest.raises(ValueError):
        Tower(cfg, bad)
    tw = Tower(cfg, w, max_batch=2)
    with pytest.raises(ValueError):
        tw.encode_patches(torch.zeros((3 * 49, 3072), dtype=torch.bfloat16, device="cuda"), 3)


def test_config5_shape_mixed_text_and_image_queries():
    """BASELINE config 5 at test scale (bf16 instead of fp8): ViT-L/14 image embeddings (768-d) as index rows, a mixed
    batch of text-tower and image-tower queries, exact top-5 against the oracle on the same rows."""
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.tower import Tower
    from oracle import search_ref as S
    vis, txt = C.CLIP_VIT_L14, C.CLIP_TEXT_L14
    wv, wt = make_weights(vis, 13), make_weights(txt, 23)
    frames = synth_frames(555, 6, 224, 224)
    img = Tower(vis, wv, max_batch=6).encode_frames(frames)
    rng = np.random.default_rng(4)
    ids = np.full((3, 20), txt.eos_id, dtype=np.int64)
    ids[:, 0] = txt.eos_id - 1
    ids[:, 1:9] = rng.integers(1, 40000, (3, 8))
    text = Tower(txt, wt, max_batch=4).encode_ids(ids)
    assert img.shape == (6, 768) and text.shape == (3, 768)
    ref_text = V.text_forward(txt, wt, ids)
    assert _cos(text.cpu().numpy(), ref_text).min() > 1 - 1e-4
    rows = S.normalize_rows_core(rng.standard_normal((5000, 768), dtype=np.float32)).astype(np.float32)
    rows[:6] = img.cpu().numpy()
    idx = FlatIPIndex(768)
    idx.add(rows)
    queries = torch.cat([text, img[:2]])                          # mixed text + image query batch, already on the device
    D, I = idx.search_device(queries, 5)
    Dr, Ir = S.flat_ip_search(rows, queries.cpu().numpy(), 5, dtype=np.float64)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.abs(D.cpu().numpy() - Dr).max() < 1e-5
    assert I[3, 0].item() == 0 and I[4, 0].item() == 1             # an image query finds its own row first
