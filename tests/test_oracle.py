"""CPU suite: the oracle against the committed golden vectors (generated from PIL / HF / sklearn by
tests/golden/make_golden.py in the build container)."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, smooth_frames, synth_frames
from ivr_amd import config as C
from ivr_amd.weights import make_weights
from oracle import preprocess_ref as P
from oracle import search_ref as S
from oracle import vit_ref as V

META = json.load(open(os.path.join(GOLDEN, "golden.json")))


def _crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


@pytest.mark.parametrize("case", META["preprocess"], ids=lambda c: f"{c['mode']}-{c['h']}x{c['w']}")
def test_preprocess_oracle_matches_pil_hf(case, golden):
    g = golden("preprocess")
    ci, h, w = case["case"], case["h"], case["w"]
    frames = np.concatenate([synth_frames(case["seeds"][0], 1, h, w), smooth_frames(case["seeds"][1], 1, h, w)])
    mean, std = (C.IMAGENET_MEAN, C.IMAGENET_STD) if case["bgr"] else (C.CLIP_MEAN, C.CLIP_STD)
    for fi, f in enumerate(frames):
        rgb = f[:, :, ::-1] if case["bgr"] else f
        u8 = P.geometry(np.ascontiguousarray(rgb), case["mode"])
        out = P.preprocess([f], case["mode"], mean, std, bgr=case["bgr"])[0]
        key = f"c{ci}_f{fi}"
        assert _crc(u8) == g[key + "_u8crc"]          # bit-exact uint8 geometry (PIL)
        assert _crc(out) == g[key + "_f32crc"]        # bit-exact float32 pixel_values (HF processor)
        assert np.array_equal(out.reshape(-1)[g[key + "_sample_idx"]], g[key + "_sample_val"])


def test_value_lut_matches_golden(golden):
    g = golden("preprocess")
    assert np.array_equal(P.value_lut(C.CLIP_MEAN, C.CLIP_STD), g["lut_clip"])
    assert np.array_equal(P.value_lut(C.IMAGENET_MEAN, C.IMAGENET_STD), g["lut_imagenet"])


@pytest.mark.parametrize("entry", [e for e in META["towers"] if "frame_seed" in e], ids=lambda e: e["tower"])
def test_vision_oracle_matches_hf(entry, golden):
    g = golden("towers")
    cfg = C.BY_NAME[entry["tower"]]
    w = make_weights(cfg, entry["weight_seed"])
    mean, std = (C.IMAGENET_MEAN, C.IMAGENET_STD) if cfg is C.DINO_VIT_S16 else (C.CLIP_MEAN, C.CLIP_STD)
    n = min(entry["n"], 4)                       # keep the CPU suite in minutes
    px = P.preprocess(synth_frames(entry["frame_seed"], entry["n"], 224, 224)[:n], "identity", mean, std)
    emb = V.vision_forward(cfg, w, px)
    assert np.abs(emb - g[cfg.name + "_emb"][:n]).max() < 2e-5   # fp32 tolerance vs HF
    raw = V.vision_forward(cfg, w, px, normalize=False)
    ref = g[cfg.name + "_raw"][:n]
    assert np.abs(raw - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize("entry", [e for e in META["towers"] if "ids_seed" in e], ids=lambda e: e["tower"])
def test_text_oracle_matches_hf(entry, golden):
    g = golden("towers")
    cfg = C.BY_NAME[entry["tower"]]
    w = make_weights(cfg, entry["weight_seed"])
    emb = V.text_forward(cfg, w, g[cfg.name + "_ids"])
    assert np.abs(emb - g[cfg.name + "_emb"]).max() < 2e-5


def test_search_oracle_matches_golden(golden):
    g = golden("search")
    m = META["search"]
    X = S.normalize_rows_core(np.random.default_rng(m["index_seed"]).standard_normal((m["n"], m["d"]), dtype=np.float32))
    X = X.astype(np.float32)
    Q = np.random.default_rng(m["query_seed"]).standard_normal((m["q"], m["d"]), dtype=np.float32)
    Qn = S.normalize_rows_core(Q).astype(np.float32)
    D, I = S.flat_ip_search(X, Qn, m["k"])
    assert np.array_equal(I, g["I"])                       # bit-exact ids
    assert np.abs(D - g["D"]).max() < 1e-5                 # fp32 scores vs fp64 brute force
    uni = np.array([[r[1] for r in S.search_vectors_rows(D[q], I[q])] for q in range(m["q"])])
    assert np.allclose(uni, g["unified_scores"], atol=1e-6) and (np.diff(uni, axis=1) >= 0).all()  # 1 - ip rises with rank
    stored = {i: X[i] for i in range(len(X))}
    leg = np.array([r[1] for r in S.legacy_search_rows(Qn, D, I, stored)]).reshape(m["q"], m["k"])
    assert np.allclose(leg, g["legacy_scores"], atol=1e-6) and leg.min() >= 0 and leg.max() <= 1


def test_search_oracle_edge_cases():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((5, 8), dtype=np.float32)
    D, I = S.flat_ip_search(X, X[:2], 8)                   # k > ntotal: -1 / -FLT_MAX padding
    assert (I[:, 5:] == -1).all() and (D[:, 5:] == S.NEG_FLT_MAX).all() and (I[:, :5] >= 0).all()
    D, I = S.flat_ip_search(np.zeros((0, 8), np.float32), X[:1], 3)
    assert (I == -1).all()
    T = np.ones((6, 4), np.float32)                        # all scores tie: ascending ids
    D, I = S.flat_ip_search(T, T[:1], 4)
    assert I.tolist() == [[0, 1, 2, 3]]
    with pytest.raises(ValueError):
        S.normalize_rows_core(np.array([[1.0, np.nan]], np.float32))
    with pytest.raises(ValueError):
        S.normalize_rows_core(np.zeros((0, 4), np.float32))
    z = S.normalize_rows_core(np.zeros((2, 4), np.float32))
    assert (z == 0).all()                                   # zero-norm rows divided by 1
    f = np.zeros((2, 4), np.float32)
    f[1] = [3, 0, 4, 0]
    S.normalize_rows_faiss(f)
    assert (f[0] == 0).all() and np.allclose(f[1], [0.6, 0, 0.8, 0])


def test_dedup_oracle_matches_sklearn_golden(golden):
    g = golden("search")
    assert np.array_equal(S.dedup_keep_mask(g["dedup_emb"], 0.98), g["dedup_keep"])
    assert S.dedup_keep_mask(g["dedup_emb"][:1]).tolist() == [True]
    assert S.dedup_keep_mask(np.zeros((0, 4))).tolist() == []


def test_merge_shards_equals_global_search():
    rng = np.random.default_rng(3)
    X = rng.standard_normal((1000, 32), dtype=np.float32)
    Q = rng.standard_normal((7, 32), dtype=np.float32)
    D, I = S.flat_ip_search(X, Q, 10)
    parts = [(0, 300), (300, 301), (301, 1000)]
    Dp = np.stack([S.flat_ip_search(X[a:b], Q, 10)[0] for a, b in parts])
    Ip = np.stack([np.where(S.flat_ip_search(X[a:b], Q, 10)[1] >= 0, S.flat_ip_search(X[a:b], Q, 10)[1] + a, -1)
                   for a, b in parts])
    Dm, Im = S.merge_shards(Dp, Ip, 10)
    assert np.array_equal(Im, I) and np.array_equal(Dm, D)


def test_faiss_flat_container_round_trip(tmp_path):
    """Layout restated from faiss's serialiser (UNPINNED: faiss is absent); the header must be the 45 packed bytes the
    docstring lists and the payload must round-trip bit-exactly."""
    from ivr_amd import faiss_io
    x = np.random.default_rng(0).standard_normal((37, 24)).astype(np.float32)
    p = tmp_path / "index.faiss"
    faiss_io.write_flat_index(str(p), x)
    raw = p.read_bytes()
    assert raw[:4] == b"IxFI" and len(raw) == 45 + x.nbytes
    assert int.from_bytes(raw[4:8], "little") == 24 and int.from_bytes(raw[8:16], "little") == 37
    y, metric = faiss_io.read_flat_index(str(p))
    assert metric == "ip" and np.array_equal(x, y)
    p.write_bytes(b"IwFl" + raw[4:])
    with pytest.raises(ValueError):
        faiss_io.read_flat_index(str(p))
    p.write_bytes(raw[:-8])
    with pytest.raises(ValueError):
        faiss_io.read_flat_index(str(p))
