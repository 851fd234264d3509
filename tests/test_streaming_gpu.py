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


def test_streaming_step_vit_b32_bf16_under_graph_capture():
    """BASELINE configs[3] with the production tower: ViT-B/32 in bf16, a 65,536-row rolling window, 8 feeds per step of 360x640
    BGR frames, hipGraph replay vs plain launches (bit-identical) and vs the oracle (ids exact over the rows the device holds,
    embeddings within the bf16 bound)."""
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.streaming import StreamingSession
    from ivr_amd.tower import Tower
    cfg = C.CLIP_VIT_B32
    w = make_weights(cfg, 12)
    window, n, steps = 65536, 8, 3
    g = torch.Generator(device="cuda").manual_seed(7)
    X0 = torch.nn.functional.normalize(torch.randn((window, 512), generator=g, device="cuda"), dim=1)
    Q = np.random.default_rng(8).standard_normal((10, 512), dtype=np.float32)
    res = {}
    for use_graph in (True, False):
        tower = Tower(cfg, w, max_batch=n)
        index = FlatIPIndex(512, capacity=window)
        index.add(X0)
        sess = StreamingSession(tower, index, n, 360, 640, torch.from_numpy(Q), k=10, mode="stretch", bgr=True, use_graph=use_graph)
        outs = []
        for step in range(steps):
            frames = smooth_frames(300 + step, n, 360, 640)
            D, I = sess.step(torch.from_numpy(frames).cuda())
            outs.append((D.cpu().numpy().copy(), I.cpu().numpy().copy()))
        res[use_graph] = (outs, index.reconstruct_n(0, steps * n), int(sess.cursor.item()))
    for (Da, Ia), (Db, Ib) in zip(res[True][0], res[False][0]):
        assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)            # graph replay == plain launches, bit for bit
    assert np.array_equal(res[True][1], res[False][1]) and res[True][2] == steps * n
    frames = np.concatenate([smooth_frames(300 + s, n, 360, 640) for s in range(steps)])
    emb = V.vision_forward(cfg, w, P.preprocess(frames, "stretch", C.CLIP_MEAN, C.CLIP_STD, bgr=True))
    assert ((res[True][1] * emb).sum(1) > 1 - 1e-4).all()                    # the rows written into the ring
    rows = X0.cpu().numpy()
    rows[:steps * n] = res[True][1]
    Dr, Ir = S.flat_ip_search(rows, S.normalize_rows_core(Q).astype(np.float32), 10, dtype=np.float64)
    D, I = res[True][0][-1]
    assert np.array_equal(I, Ir) and np.abs(D - Dr).max() < 1e-5
