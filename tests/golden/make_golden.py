#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Runs ONLY in the build container.

The reference (`/root/reference`) cannot be imported here (faiss/h5py/lz4/cv2 are
absent and its model loaders fetch by hub name - SURVEY.md §8c), and it holds no
golden vectors of its own.  What it delegates the hot path to IS installed, so
this script runs those libraries directly - PIL's resampler, HuggingFace's
CLIPImageProcessorPil / CLIPVisionModelWithProjection / CLIPTextModelWithProjection
/ ViTModel built from local configs (no hub access), sklearn's cosine_similarity -
on seeded inputs, first asserts that oracle/ reproduces them, then stores the
expected outputs.  Nothing from transformers travels: only inputs (as seeds) and
expected numbers are written.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import torch  # noqa: E402

from ivr_amd import config as C  # noqa: E402
from ivr_amd.weights import make_weights, to_hf_state_dict  # noqa: E402
from oracle import preprocess_ref as P  # noqa: E402
from oracle import search_ref as S  # noqa: E402
from oracle import vit_ref as V  # noqa: E402


def synth_frames(seed, n, h, w):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def smooth_frames(seed, n, h, w):
    """Low-frequency content so resampling is exercised on something image-like too."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    for i in range(n):
        for c in range(3):
            f = rng.uniform(0.5, 6.0, 4)
            ph = rng.uniform(0, 6.28, 4)
            z = (np.sin(f[0] * xx / w * 6.28 + ph[0]) * np.cos(f[1] * yy / h * 6.28 + ph[1])
                 + 0.5 * np.sin(f[2] * (xx + yy) / (w + h) * 6.28 + ph[2]) + 0.2 * rng.standard_normal((h, w)))
            out[i, :, :, c] = np.clip(127.5 + 80 * z, 0, 255).astype(np.uint8)
    return out


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def golden_preprocess():
    from PIL import Image
    from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    proc = CLIPImageProcessorPil()
    cases = [("identity", 224, 224), ("shortest_edge_crop", 256, 341), ("shortest_edge_crop", 341, 256),
             ("shortest_edge_crop", 720, 1280), ("shortest_edge_crop", 100, 180), ("stretch", 360, 640),
             ("stretch", 1080, 1920), ("stretch", 120, 90)]
    out = {}
    meta = []
    for ci, (mode, h, w) in enumerate(cases):
        frames = np.concatenate([synth_frames(100 + ci, 1, h, w), smooth_frames(200 + ci, 1, h, w)])
        for fi, f in enumerate(frames):
            if mode == "stretch":
                # video_frame_filter.py:58-59 on a BGR frame, then rescale/normalise with ImageNet constants
                rgb = np.ascontiguousarray(f[:, :, ::-1])
                u8 = np.asarray(Image.fromarray(rgb).resize((224, 224)))
                mean, std = C.IMAGENET_MEAN, C.IMAGENET_STD
                exp = ((u8.astype(np.float64) * (1 / 255)).astype(np.float32) - np.array(mean, np.float32)) \
                    / np.array(std, np.float32)
                exp = np.ascontiguousarray(exp.transpose(2, 0, 1))
                got = P.preprocess([f], "stretch", mean, std, bgr=True)[0]
                got_u8 = P.geometry(rgb, "stretch")
                assert np.array_equal(got_u8, u8), (mode, h, w, "u8 geometry differs from PIL")
            else:
                exp = proc(images=[Image.fromarray(f)], return_tensors="np")["pixel_values"][0]
                got = P.preprocess([f], mode, C.CLIP_MEAN, C.CLIP_STD)[0]
                got_u8 = P.geometry(f, mode)
            assert exp.shape == got.shape == (3, 224, 224)
            assert np.array_equal(exp.view(np.uint32), got.view(np.uint32)), (mode, h, w, np.abs(exp - got).max())
            key = f"c{ci}_f{fi}"
            out[key + "_u8crc"] = crc(got_u8)
            out[key + "_f32crc"] = crc(exp)
            rs = np.random.default_rng(7).integers(0, exp.size, 256)
            out[key + "_sample_idx"] = rs.astype(np.int32)
            out[key + "_sample_val"] = exp.reshape(-1)[rs]
        meta.append({"case": ci, "mode": mode, "h": h, "w": w, "seeds": [100 + ci, 200 + ci],
                     "bgr": mode == "stretch"})
    out["lut_clip"] = P.value_lut(C.CLIP_MEAN, C.CLIP_STD)
    out["lut_imagenet"] = P.value_lut(C.IMAGENET_MEAN, C.IMAGENET_STD)
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    return meta


def hf_vision(cfg, w):
    if cfg.pool == C.POOL_LN_ALL_CLS:
        from transformers import ViTConfig, ViTModel
        hc = ViTConfig(hidden_size=cfg.width, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                       intermediate_size=cfg.mlp, image_size=cfg.image, patch_size=cfg.patch,
                       hidden_act="gelu", layer_norm_eps=cfg.ln_eps, qkv_bias=True)
        m = ViTModel(hc, add_pooling_layer=False).eval()
        sd = {k: torch.from_numpy(v) for k, v in to_hf_state_dict(cfg, w).items()}
        missing = m.load_state_dict(sd, strict=True)
        return lambda px: m(pixel_values=torch.from_numpy(px)).last_hidden_state[:, 0, :]
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    hc = CLIPVisionConfig(hidden_size=cfg.width, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers,
                          num_attention_heads=cfg.heads, image_size=cfg.image, patch_size=cfg.patch,
                          projection_dim=cfg.out_dim, hidden_act="quick_gelu", layer_norm_eps=cfg.ln_eps)
    m = CLIPVisionModelWithProjection(hc).eval()
    sd = {k: torch.from_numpy(v) for k, v in to_hf_state_dict(cfg, w).items()}
    sd["vision_model.embeddings.position_ids"] = torch.arange(cfg.tokens).unsqueeze(0)
    m.load_state_dict(sd, strict=False)
    return lambda px: m(pixel_values=torch.from_numpy(px)).image_embeds


def hf_text(cfg, w):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    hc = CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.width, intermediate_size=cfg.mlp,
                        num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                        max_position_embeddings=cfg.tokens, projection_dim=cfg.out_dim, hidden_act="quick_gelu",
                        layer_norm_eps=cfg.ln_eps, eos_token_id=cfg.eos_id, bos_token_id=cfg.eos_id - 1,
                        pad_token_id=0)
    m = CLIPTextModelWithProjection(hc).eval()
    sd = {k: torch.from_numpy(v) for k, v in to_hf_state_dict(cfg, w).items()}
    m.load_state_dict(sd, strict=False)
    return lambda ids: m(input_ids=torch.from_numpy(ids)).text_embeds


def synth_token_ids(cfg, seed, q):
    """Rows of [BOS, words..., EOS, pad...] like the CLIP tokenizer emits (core.py:1532-1538)."""
    rng = np.random.default_rng(seed)
    ids = np.zeros((q, cfg.tokens), dtype=np.int64)
    for r in range(q):
        n = int(rng.integers(1, cfg.tokens - 2))
        ids[r, 0] = cfg.eos_id - 1
        ids[r, 1:1 + n] = rng.integers(1, cfg.eos_id - 1, n)
        ids[r, 1 + n] = cfg.eos_id
        ids[r, 2 + n:] = cfg.eos_id          # HF CLIP tokenizer pads with EOS: pooling must take the FIRST one
    return ids


def golden_towers():
    meta = []
    out = {}
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vis = [(C.TINY_VIT, 11, 4), (C.CLIP_VIT_B32, 12, 8), (C.CLIP_VIT_L14, 13, 2), (C.DINO_VIT_S16, 14, 2)]
    for cfg, seed, n in vis:
        w = make_weights(cfg, seed)
        mean, std = (C.IMAGENET_MEAN, C.IMAGENET_STD) if cfg is C.DINO_VIT_S16 else (C.CLIP_MEAN, C.CLIP_STD)
        px = P.preprocess(synth_frames(1234, n, 224, 224), "identity", mean, std)
        with torch.no_grad():
            raw = hf_vision(cfg, w)(px).numpy()
        hf = raw / np.maximum(np.linalg.norm(raw, axis=1, keepdims=True), 1e-12)
        ours_raw = V.vision_forward(cfg, w, px, normalize=False)
        ours = V.vision_forward(cfg, w, px)
        err_raw = float(np.abs(ours_raw - raw).max() / np.abs(raw).max())
        err = float(np.abs(ours - hf).max())
        print(f"{cfg.name}: oracle vs HF  rel raw {err_raw:.2e}  normalised abs {err:.2e}")
        assert err < 2e-5 and err_raw < 2e-5, cfg.name
        out[cfg.name + "_emb"] = hf.astype(np.float32)
        out[cfg.name + "_raw"] = raw.astype(np.float32)
        meta.append({"tower": cfg.name, "weight_seed": seed, "frame_seed": 1234, "n": n, "oracle_vs_hf": err})
        if cfg is C.TINY_VIT:   # full intermediate dump for kernel bring-up
            _, hidden = V.vision_forward(cfg, w, px, return_hidden=True)
            for li, h in enumerate(hidden):
                out[f"{cfg.name}_hidden{li}"] = h.astype(np.float32)
    for cfg, seed, q in [(C.TINY_TEXT, 21, 4), (C.CLIP_TEXT_B32, 22, 4)]:
        w = make_weights(cfg, seed)
        ids = synth_token_ids(cfg, 777, q)
        with torch.no_grad():
            raw = hf_text(cfg, w)(ids).numpy()
        hf = raw / np.maximum(np.linalg.norm(raw, axis=1, keepdims=True), 1e-12)
        ours = V.text_forward(cfg, w, ids)
        err = float(np.abs(ours - hf).max())
        print(f"{cfg.name}: oracle vs HF  normalised abs {err:.2e}")
        assert err < 2e-5, cfg.name
        out[cfg.name + "_ids"] = ids
        out[cfg.name + "_emb"] = hf.astype(np.float32)
        meta.append({"tower": cfg.name, "weight_seed": seed, "ids_seed": 777, "q": q, "oracle_vs_hf": err})
    np.savez_compressed(os.path.join(HERE, "towers.npz"), **out)
    return meta


def golden_search():
    from sklearn.metrics.pairwise import cosine_similarity
    out = {}
    rng = np.random.default_rng(5678)
    X = rng.standard_normal((4096, 512), dtype=np.float32)
    X = S.normalize_rows_core(X).astype(np.float32)
    Q = np.random.default_rng(91011).standard_normal((10, 512), dtype=np.float32)
    Qn = S.normalize_rows_core(Q).astype(np.float32)
    D64, I64 = S.flat_ip_search(X, Qn, 10, dtype=np.float64)
    D32, I32 = S.flat_ip_search(X, Qn, 10)
    # independent brute force (no shared code with the oracle)
    s = Qn.astype(np.float64) @ X.astype(np.float64).T
    Ib = np.argsort(-s, axis=1, kind="stable")[:, :10]
    assert np.array_equal(Ib, I64) and np.array_equal(I32, I64)
    gaps = np.diff(-np.sort(-s, axis=1)[:, :11], axis=1)
    assert np.abs(gaps).min() > 1e-5, "seeded data must be tie-free"
    out.update(I=I64, D=D64, min_gap=np.float64(np.abs(gaps).min()))
    # the two score conventions of the reference
    out["unified_scores"] = np.array([[r[1] for r in S.search_vectors_rows(D32[q], I32[q])] for q in range(10)])
    stored = {i: X[i] for i in range(len(X))}
    out["legacy_scores"] = np.array([r[1] for r in S.legacy_search_rows(Qn, D32, I32, stored)]).reshape(10, 10)
    # dedup rule against sklearn itself
    E = np.cumsum(np.random.default_rng(42).standard_normal((64, 384)) * 0.35, axis=0) + 3.0
    keep = S.dedup_keep_mask(E, 0.98)
    prev, kk = None, []
    for e in E:
        u = True
        if prev is not None and cosine_similarity([e], [prev])[0][0] >= 0.98:
            u = False
        if u:
            prev = e
        kk.append(u)
    assert np.array_equal(keep, np.array(kk)) and 4 < keep.sum() < 60, keep.sum()
    out["dedup_emb"] = E.astype(np.float32)
    keep32 = S.dedup_keep_mask(E.astype(np.float32), 0.98)
    out["dedup_keep"] = keep32
    np.savez_compressed(os.path.join(HERE, "search.npz"), **out)
    return {"index_seed": 5678, "query_seed": 91011, "n": 4096, "d": 512, "q": 10, "k": 10,
            "dedup_seed": 42, "dedup_kept": int(keep32.sum())}


if __name__ == "__main__":
    import PIL
    import sklearn
    import transformers
    side = {"versions": {"numpy": np.__version__, "torch": torch.__version__, "transformers": transformers.__version__,
                         "PIL": PIL.__version__, "sklearn": sklearn.__version__},
            "preprocess": golden_preprocess(), "search": golden_search(), "towers": golden_towers()}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(side, f, indent=1)
    print("golden fixtures written")
