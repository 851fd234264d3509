"""GPU: the hipGraph-captured streaming step (BASELINE config 4, scaled down) against eager execution and the oracle."""
import numpy as np
import pytest
import torch

from conftest import smooth_frames
from ivr_amd import config as C
from ivr_amd.weights import make_weights
from oracle import preprocess_ref as P
from oracle import search_ref as S
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_graph", [True, False])
def test_rolling_window_step(use_graph):
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.streaming import StreamingSession
    from ivr_amd.tower import Tower
    cfg = C.TINY_VIT
    w = make_weights(cfg, 11)
    tower = Tower(cfg, w, max_batch=16, compute="f32")
    rng = np.random.default_rng(3)
    window = 48                                              # rolling index of 48 rows, 16 frames per step
    X0 = S.normalize_rows_core(rng.standard_normal((window, cfg.embed_dim), dtype=np.float32)).astype(np.float32)
    index = FlatIPIndex(cfg.embed_dim)
    index.add(X0)
    Q = rng.standard_normal((5, cfg.embed_dim), dtype=np.float32)
    sess = StreamingSession(tower, index, 16, 120, 160, torch.from_numpy(Q), k=7, mode="stretch", bgr=True, use_graph=use_graph)
    ref_rows = X0.copy()
    qn = S.normalize_rows_core(Q).astype(np.float32)
    for step in range(5):                                    # wraps around the window once
        frames = smooth_frames(100 + step, 16, 120, 160)
        D, I = sess.step(torch.from_numpy(frames).cuda())
        emb = V.vision_forward(cfg, w, P.preprocess(frames, "stretch", C.CLIP_MEAN, C.CLIP_STD, bgr=True))
        pos = (step * 16) % window
        ref_rows[pos:pos + 16] = emb
        Dr, Ir = S.flat_ip_search(ref_rows, qn, 7, dtype=np.float64)
        assert np.array_equal(I.cpu().numpy(), Ir), step
        assert np.abs(D.cpu().numpy() - Dr).max() < 1e-4
    assert int(sess.cursor.item()) == (5 * 16) % window
    assert np.abs(index.reconstruct_n() - ref_rows).max() < 1e-5
