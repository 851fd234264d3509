"""Row-sharded exact search across the GPUs of one node (one process per GPU).

MI355X-native replacement of the reference's only distributed pattern - scatter the
query vector to peers over HTTP, gather their top-k lists, concatenate and sort
(`system.py:1715-1757`, `api.py:877-925`): here every rank owns a contiguous row
range of the index in its own HBM, the (small) query batch is replicated, each rank
scans its shard, and ONE all-gather of the per-shard (score, global id) candidates
over RCCL/xGMI (`torch.distributed`, backend "nccl") is followed by a k-way merge.
The message is nq*k*12 bytes per rank (1000x10 -> 120 KB): latency-bound, so a single
collective is the right shape; no other exchange exists on this path.

The class is agnostic of how a shard is searched: `local` only needs
`search_device(q, k, normalize=, id_base=) -> (D, I)` tensors, `add`, `ntotal`.  On GPUs that
is `ivr_amd.index.FlatIPIndex`; the world_size-2 gloo tests on CPU plug the oracle in.
"""
import numpy as np
import torch
import torch.distributed as dist

NEG_FLT_MAX = -3.4028234663852886e38


def shard_bounds(n_rows, world_size):
    """Contiguous row ranges [lo, hi) per rank; the first n_rows % world_size ranks take one extra row."""
    base, extra = divmod(int(n_rows), int(world_size))
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def merge_host(D_parts, I_parts, k):
    """Host-side final merge (north_star): D_parts/I_parts [G,nq,k] with global ids, shards in ascending id
    order.  Sort by score descending, ties to the lower id; unused slots are (-FLT_MAX, -1)."""
    D_parts = D_parts.detach().cpu()
    I_parts = I_parts.detach().cpu()
    G, nq, kk = D_parts.shape
    d = D_parts.permute(1, 0, 2).reshape(nq, G * kk)
    i = I_parts.permute(1, 0, 2).reshape(nq, G * kk)
    d = torch.where(i >= 0, d, torch.full_like(d, float("-inf")))
    # stable sort on the score keeps candidate order among ties = ascending (shard, rank) = ascending id
    order = torch.sort(d, dim=1, descending=True, stable=True).indices[:, :k]
    D = torch.gather(d, 1, order)
    I = torch.gather(i, 1, order)
    D = torch.where(I >= 0, D, torch.full_like(D, NEG_FLT_MAX))
    if D.shape[1] < k:
        pad = k - D.shape[1]
        D = torch.cat([D, torch.full((nq, pad), NEG_FLT_MAX)], 1)
        I = torch.cat([I, torch.full((nq, pad), -1, dtype=torch.int64)], 1)
    return D, I


class ShardedIndex:
    def __init__(self, local, d, group=None, merge="device"):
        self.local = local
        self.d = int(d)
        self.group = group
        self.merge = merge
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.id_base = 0
        self._counts = [0] * self.world

    # -- build -------------------------------------------------------------------------------
    def add_local(self, rows, normalize=False):
        """Append this rank's rows, then agree on every shard's global id offset (one tiny all-gather of counts;
        build-time only, never on the search path)."""
        self.local.add(rows, normalize=normalize) if normalize else self.local.add(rows)
        self.sync_counts()

    def sync_counts(self):
        n = int(self.local.ntotal)
        if self.world > 1:
            counts = [None] * self.world
            dist.all_gather_object(counts, n, group=self.group)
        else:
            counts = [n]
        self._counts = [int(c) for c in counts]
        self.id_base = int(sum(self._counts[:self.rank]))

    @property
    def ntotal(self):
        return int(sum(self._counts))

    # -- search ------------------------------------------------------------------------------
    def search(self, q, k, normalize=False):
        """q [nq,d] replicated on every rank -> (D [nq,k], I [nq,k] global ids), identical on every rank."""
        D, I = self.local.search_device(q, k, normalize=normalize, id_base=self.id_base)
        return self.exchange(D, I)

    def exchange(self, D, I):
        """The single exchange step of the path: this rank's candidates (D [nq,k] scores, I [nq,k] GLOBAL ids, unused slots -1)
        -> the merged top-k over all shards, identical on every rank.  ONE all-gather of nq*k*12 bytes per rank - (score f32,
        id i64) packed as three int32 words per candidate - then the k-way merge.  Enqueued on the current stream (RCCL work is
        stream-ordered), so it can follow a replayed HIP graph that produced D and I (streaming.StreamingSession)."""
        if self.world == 1:
            return D, I
        nq, k = D.shape
        if self.merge == "device" and D.is_cuda:
            # pack (one launch) -> the all-gather -> merge straight from the gathered buffer (one launch)
            from .index import topk_merge_packed, topk_pack
            cand = topk_pack(D, I)
            gathered = torch.empty((self.world * nq, k, 3), dtype=torch.int32, device=D.device)     # concatenated along dim 0
            dist.all_gather_into_tensor(gathered, cand, group=self.group)
            return topk_merge_packed(gathered.view(self.world, nq, k, 3))
        cand = torch.empty((nq, k, 3), dtype=torch.int32, device=D.device)
        cand[..., 0] = D.contiguous().view(torch.int32)
        cand[..., 1:] = I.contiguous().view(torch.int32).view(nq, k, 2)
        gathered = torch.empty((self.world * nq, k, 3), dtype=torch.int32, device=D.device)
        dist.all_gather_into_tensor(gathered, cand, group=self.group)
        gathered = gathered.view(self.world, nq, k, 3)
        Dg = gathered[..., 0].contiguous().view(torch.float32)
        Ig = gathered[..., 1:].contiguous().view(torch.int64).view(self.world, nq, k)
        return merge_host(Dg, Ig, k)


def stride_frames(n_frames, rank, world):
    """Frame i is embedded on GPU i mod G: indices owned by `rank`."""
    return np.arange(rank, n_frames, world)
