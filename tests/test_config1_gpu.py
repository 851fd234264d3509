"""BASELINE config 1 end to end on the GPU against the CPU path: synthetic 224x224 frames -> 512-d embeddings ->
index -> 10 Gaussian queries, top-10.  (256 of the 1,000 frames keep the CPU oracle's ViT-B/32 pass to a few seconds.)"""
import numpy as np
import pytest
import torch

from conftest import synth_frames
from ivr_amd import config as C
from ivr_amd.weights import make_weights
from oracle import preprocess_ref as P
from oracle import search_ref as S
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu


def test_embed_index_search_against_cpu_path():
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.tower import Tower
    cfg = C.CLIP_VIT_B32
    w = make_weights(cfg, 12)
    frames = synth_frames(1234, 256, 224, 224)
    emb = Tower(cfg, w, max_batch=256).encode_frames(frames)                  # bf16 tower, rows L2-normalised
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = np.concatenate([V.vision_forward(cfg, w, P.preprocess(frames[i:i + 32], "identity", C.CLIP_MEAN, C.CLIP_STD))
                          for i in range(0, 256, 32)])                       # batches of 32: core.py:1558
    got = emb.cpu().numpy()
    cos = (got * ref).sum(1)
    assert cos.min() > 1 - 1e-4                                               # north-star bound on cosine scores: 1e-3
    Q = np.random.default_rng(91011).standard_normal((10, 512), dtype=np.float32)
    idx = FlatIPIndex(512)
    idx.add(emb)
    D, I = idx.search_device(Q, 10, normalize=True)
    qn = S.normalize_rows_core(Q).astype(np.float32)
    Dr, Ir = S.flat_ip_search(got, qn, 10, dtype=np.float64)                  # same rows, CPU search: ids bit-exact
    assert np.array_equal(I.cpu().numpy(), Ir) and np.abs(D.cpu().numpy() - Dr).max() < 1e-5
    # scores against the CPU-embedded index stay within the north-star tolerance (random-weight embeddings of random
    # frames are nearly collinear, so id equality across the two embedders is ill-conditioned and not asserted)
    Dc, _ = S.flat_ip_search(ref, qn, 10, dtype=np.float64)
    assert np.abs(D.cpu().numpy() - Dc).max() < 1e-3
