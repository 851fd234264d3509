"""CPU suite: the N-rank streaming step (every feed pinned to one GPU, each rank overwrites its own ring,
replicated queries, the same single all-gather + merge per search step) with world_size 2 over gloo.

The real `StreamingSession` and `ShardedIndex` run; the three device pieces are stand-ins (no GPU in this container): a
deterministic embedder for the tower, a pooling stand-in for the resize kernel, and a ring backed by oracle/search_ref.py for the
HIP index.  After every step, through a wrap-around of both rings, the merged ids on every rank must equal the oracle search over
the UNION of both rings (global id = rank * window + ring position).  The GPU form of the same step is tests/test_streaming_gpu.py."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

D_EMB, G2, KPAD = 24, 4, 768          # KPAD = 3 * patch^2 rounded up to 64 (what StreamingSession allocates per patch row)


class RingDouble:
    """FlatIPIndex surface of the streaming step: reserve_search / write_ring / search_device(out=, id_base=)."""

    def __init__(self, rows):
        self.rows = np.array(rows, np.float32)
        self.d = self.rows.shape[1]

    @property
    def ntotal(self):
        return len(self.rows)

    def reserve_search(self, nq, k):
        pass

    def write_ring(self, emb, cursor):
        pos, n = int(cursor.item()), emb.shape[0]
        self.rows[pos:pos + n] = emb.numpy()
        cursor.fill_((pos + n) % len(self.rows))

    def search_device(self, q, k, normalize=False, id_base=0, out=None):
        from oracle import search_ref as S
        qn = np.asarray(q, np.float32)
        if normalize:
            qn = S.normalize_rows_core(qn).astype(np.float32)
        D, I = S.flat_ip_search(self.rows, qn, k)
        I = np.where(I >= 0, I + id_base, -1)
        if out is not None:
            out[0].copy_(torch.from_numpy(D))
            out[1].copy_(torch.from_numpy(I))
            return out
        return torch.from_numpy(D), torch.from_numpy(I)


def pool_frames(frames, mode, mean, std, bgr=False, size=0, patch=0, out_dtype=None, out=None):
    """Stand-in for ivr_preprocess: every frame -> G2 'patches' of KPAD numbers (block means of its bytes)."""
    n = frames.shape[0]
    v = frames.reshape(n, -1).to(torch.float32)
    v = v[:, : (v.shape[1] // (G2 * KPAD)) * G2 * KPAD].reshape(n, G2 * KPAD, -1).mean(2) / 255.0
    out.copy_(v.reshape(n * G2, KPAD))
    return out


class TowerDouble:
    def __init__(self, proj):
        self.device = torch.device("cpu")
        self.cfg = SimpleNamespace(image=32, patch=16)        # (32 / 16)^2 = G2 patches per frame
        self.act_dtype = torch.float32
        self.embed_dim = D_EMB
        self.max_batch = 64
        self.proj = proj

    def encode_patches(self, patches, n, normalize=True, out=None):
        e = patches.reshape(n, -1) @ self.proj
        e = e - e.mean(1, keepdim=True)
        out.copy_(torch.nn.functional.normalize(e, dim=1))
        return out


def frames_of(rank, step, n):
    return torch.from_numpy(np.random.default_rng(1000 * rank + step).integers(0, 256, (n, 32, 32, 3), dtype=np.uint8))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ivr_amd.sharded import ShardedIndex
    from ivr_amd.streaming import StreamingSession
    from oracle import search_ref as S
    window, n, k, steps = 24, 4, 6, 9                       # 6 steps fill a ring: steps 7-9 overwrite its oldest rows
    rng = np.random.default_rng(5)
    proj = torch.from_numpy(rng.standard_normal((G2 * KPAD, D_EMB)).astype(np.float32))
    rings0 = [S.normalize_rows_core(np.random.default_rng(50 + r).standard_normal((window, D_EMB)).astype(np.float32)).astype(np.float32)
              for r in range(world)]
    Q = rng.standard_normal((5, D_EMB)).astype(np.float32)
    Q[0] = rings0[1][13]                                     # a query that IS a row of rank 1's initial ring
    tower, index = TowerDouble(proj), RingDouble(rings0[rank])
    sh = ShardedIndex(index, D_EMB, merge="host")
    sh.sync_counts()
    assert sh.id_base == rank * window and sh.ntotal == world * window
    sess = StreamingSession(tower, index, n, 32, 32, torch.from_numpy(Q), k=k, use_graph=False, sharded=sh, preprocess=pool_frames)
    # the reference state: every rank replays every rank's feed with the same stand-ins
    union = [r.copy() for r in rings0]
    qn = S.normalize_rows_core(Q).astype(np.float32)
    ok = True
    for step in range(steps):
        D, I = sess.step(frames_of(rank, step, n))
        pos = (step * n) % window
        for r in range(world):
            patches = torch.empty((n * G2, KPAD))
            pool_frames(frames_of(r, step, n), None, None, None, out=patches)
            emb = torch.empty((n, D_EMB))
            tower.encode_patches(patches, n, out=emb)
            union[r][pos:pos + n] = emb.numpy()
        Dr, Ir = S.flat_ip_search(np.concatenate(union), qn, k)
        ok = ok and np.array_equal(I.numpy(), Ir) and np.allclose(D.numpy(), Dr, rtol=1e-6, atol=1e-6)
        if step == 0:
            ok = ok and int(I[0, 0]) == window + 13          # rank 1's row, reported under its global id on BOTH ranks
    ok = ok and int(sess.cursor.item()) == (steps * n) % window
    gathered = [None] * world
    dist.all_gather_object(gathered, I.numpy().tolist())
    ok = ok and all(g == gathered[0] for g in gathered)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_streaming_step_matches_the_oracle_over_both_rings():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_session_rejects_a_foreign_shard():
    import pytest
    from ivr_amd.sharded import ShardedIndex
    from ivr_amd.streaming import StreamingSession
    proj = torch.zeros((G2 * KPAD, D_EMB))
    a, b = RingDouble(np.zeros((8, D_EMB))), RingDouble(np.zeros((8, D_EMB)))
    with pytest.raises(ValueError):
        StreamingSession(TowerDouble(proj), a, 4, 32, 32, torch.zeros((1, D_EMB)), use_graph=False, sharded=ShardedIndex(b, D_EMB),
                         preprocess=pool_frames)
