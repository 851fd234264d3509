"""GPU: the N-rank streaming step with the real kernels.  Two ranks share the one GPU of the box and
rendezvous over gloo (RCCL needs one device per rank): each rank owns its ring, replays ITS captured HIP graph (resize ->
tower -> ring overwrite -> local search with global ids) and then issues the single all-gather + merge on the same stream.
Merged ids on both ranks == the oracle over the union of both rings, across a wrap-around."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, use_graph, ret):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import smooth_frames
    from ivr_amd import config as C
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.sharded import ShardedIndex
    from ivr_amd.streaming import StreamingSession
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    from oracle import preprocess_ref as P
    from oracle import search_ref as S
    from oracle import vit_ref as V
    torch.cuda.set_device(0)
    cfg = C.TINY_VIT
    w = make_weights(cfg, 11)
    tower = Tower(cfg, w, max_batch=8, compute="f32")
    window, n, k, steps = 32, 8, 7, 6                        # 4 steps fill a ring, steps 5-6 wrap
    rings0 = [S.normalize_rows_core(np.random.default_rng(60 + r).standard_normal((window, cfg.embed_dim), dtype=np.float32)).astype(np.float32)
              for r in range(world)]
    index = FlatIPIndex(cfg.embed_dim)
    index.add(rings0[rank])
    Q = np.random.default_rng(3).standard_normal((5, cfg.embed_dim), dtype=np.float32)
    sh = ShardedIndex(index, cfg.embed_dim, merge="device")
    sh.sync_counts()
    sess = StreamingSession(tower, index, n, 120, 160, torch.from_numpy(Q), k=k, mode="stretch", bgr=True, use_graph=use_graph, sharded=sh)
    union = [r.copy() for r in rings0]
    qn = S.normalize_rows_core(Q).astype(np.float32)
    ok = sh.id_base == rank * window
    for step in range(steps):
        frames = [smooth_frames(500 + 10 * r + step, n, 120, 160) for r in range(world)]
        D, I = sess.step(torch.from_numpy(frames[rank]).cuda())
        pos = (step * n) % window
        for r in range(world):
            union[r][pos:pos + n] = V.vision_forward(cfg, w, P.preprocess(frames[r], "stretch", C.CLIP_MEAN, C.CLIP_STD, bgr=True))
        Dr, Ir = S.flat_ip_search(np.concatenate(union), qn, k, dtype=np.float64)
        ok = ok and np.array_equal(I.cpu().numpy(), Ir) and float(np.abs(D.cpu().numpy() - Dr).max()) < 1e-4
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [True, False])
def test_two_rank_streaming_step_on_one_gpu(use_graph):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_graph, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_single_rank_sharded_session_equals_the_plain_session():
    from conftest import smooth_frames
    from ivr_amd import config as C
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.sharded import ShardedIndex
    from ivr_amd.streaming import StreamingSession
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    cfg = C.TINY_VIT
    tower = Tower(cfg, make_weights(cfg, 11), max_batch=8, compute="f32")
    rng = np.random.default_rng(2)
    X0 = rng.standard_normal((32, cfg.embed_dim), dtype=np.float32)
    Q = torch.from_numpy(rng.standard_normal((4, cfg.embed_dim), dtype=np.float32))
    res = []
    for wrap in (False, True):
        index = FlatIPIndex(cfg.embed_dim)
        index.add(X0, normalize=True)
        sh = ShardedIndex(index, cfg.embed_dim) if wrap else None
        sess = StreamingSession(tower, index, 8, 120, 160, Q, k=5, use_graph=True, sharded=sh)
        for step in range(3):
            D, I = sess.step(torch.from_numpy(smooth_frames(700 + step, 8, 120, 160)).cuda())
        res.append((D.clone(), I.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
