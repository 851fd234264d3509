#!/usr/bin/env python3
"""bench.py - the hot path of BASELINE.json on N GPUs of one node.

Workload (config.workload): BASELINE.json configs[1] - ViT-B/32 bf16 random-init, ~100k synthetic 224x224 RGB
frames embedded to 512-d, and a 1M-row x 512-d float32 index searched with 10 queries, top-10 - per GPU.
A "step" is one pass of the whole path over one batch, inputs already resident in HBM:
    uint8 NHWC frames --HIP preprocess--> bf16 patches --HIP ViT-B/32--> L2-normalised rows
    --HIP append (ring overwrite)--> the 1M-row index --HIP cosine top-10--> (score, id) for 10 queries
    [N > 1: one RCCL all-gather of the per-shard candidates + merge]
value = frames embedded per second by the whole job (every frame also pays its share of the search);
pairs_per_s = query x index-row cosines per second of the search part alone (HIP events), also reported.
extra.configs2_sharded (every N): BASELINE configs[2] as a strong-scaling leg - 10M rows x 512-d split over the N ranks, 1,000
replicated queries, local exact top-10, one all-gather of (score, id) + merge: search_ms, allgather_merge_ms, pairs_per_s.

python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Started from a plain shell the script launches `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` itself as a child process and relays rank 0's JSON line;
started under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PMC_NAME = {"scan_groupmax": "scan_groupmax_kernel<1>", "scan16_groupmax": "scan16_ring_kernel<16>", "preprocess_emit": "emit_vec_kernel<bf16, 32>"}
PMC_FILE = "r03_pmc_traffic.json"   # the committed PMC passes `traffic` is read from (tools/pmc_aggregate.py)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA


def cpu_baseline(cfg, weights, index_rows, queries, k, budget_s=25.0):
    """The oracle (a port: the reference's own Python cannot travel) timed on this box's host
    cores.  Two parts, both reported:
      * BASELINE.json configs[0] exactly: 1,000 synthetic 224x224 frames (default_rng(1234)) through the fp32 tower in
        batches of 32 (core.py:1558) -> 1,000 x 512 rows -> 10 Gaussian queries (default_rng(91011)), exact top-10;
        cut short (and said so) only if the embedding alone would exceed the time budget;
      * the exact inner-product top-k over a 100,000-row slice of the benchmark's own index: the 1,000-row search of
        configs[0] lasts microseconds, too short to quote a pairs/s figure from."""
    from oracle import preprocess_ref as P
    from oracle import search_ref as S
    from oracle import vit_ref as V
    from ivr_amd import config as C
    # threads actually used: this process's CPU share (the box has more logical cores than a 1-GPU job may use)
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    frames = np.random.default_rng(1234).integers(0, 256, (1000, 224, 224, 3), dtype=np.uint8)
    V.vision_forward(cfg, weights, P.preprocess(frames[:8], "identity", C.CLIP_MEAN, C.CLIP_STD))      # warm-up
    embs, done, t0 = [], 0, time.perf_counter()
    while done < len(frames):
        px = P.preprocess(frames[done:done + 32], "identity", C.CLIP_MEAN, C.CLIP_STD)
        embs.append(V.vision_forward(cfg, weights, px))
        done += len(px)
        if time.perf_counter() - t0 > budget_s:
            break
    t_embed = time.perf_counter() - t0
    rows0 = np.concatenate(embs)
    q0 = S.normalize_rows_core(np.random.default_rng(91011).standard_normal((10, rows0.shape[1]), dtype=np.float32)).astype(np.float32)
    t1 = time.perf_counter()
    for _ in range(20):
        D0, I0 = S.flat_ip_search(rows0, q0, 10)
    t_s0 = (time.perf_counter() - t1) / 20
    S.flat_ip_search(index_rows[:1000], queries, k)              # warm-up
    t1 = time.perf_counter()
    reps = 0
    while True:
        S.flat_ip_search(index_rows, queries, k)
        reps += 1
        if time.perf_counter() - t1 > 5.0 or reps >= 20:
            break
    t_search = (time.perf_counter() - t1) / reps
    # per-core figure: 64 of the same frames and the same 100k-row search on ONE thread
    torch.set_num_threads(1)
    t1 = time.perf_counter()
    for i in range(0, 64, 32):
        V.vision_forward(cfg, weights, P.preprocess(frames[i:i + 32], "identity", C.CLIP_MEAN, C.CLIP_STD))
    t_embed1 = time.perf_counter() - t1
    t1 = time.perf_counter()
    S.flat_ip_search(index_rows, queries, k)
    t_search1 = time.perf_counter() - t1
    torch.set_num_threads(cores)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": done / t_embed, "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "logical_cpus": os.cpu_count(),
            "one_thread": {"frames_per_s": 64 / t_embed1, "pairs_per_s": len(index_rows) * len(queries) / t_search1,
                           "sample": f"64 frames through the fp32 oracle + one exact IP top-{k} of {len(queries)} queries over the "
                                     f"{len(index_rows)}-row slice, torch.set_num_threads(1) (NumPy's BLAS keeps its own pool for the search)"},
            "sample": f"configs[0]{' exactly' if done == len(frames) else ' cut at the time budget'}: {done} of 1000 frames of 224x224 "
                      f"through the fp32 oracle {cfg.name} in batches of 32 ({t_embed:.1f} s), 10 queries top-10 over those "
                      f"{len(rows0)} rows ({t_s0 * 1e6:.0f} us per search); pairs/s from the exact IP top-{k} of {len(queries)} queries "
                      f"over a {len(index_rows)} x {index_rows.shape[1]} slice of the benchmark's index, {reps} repeats",
            "pairs_per_s": len(index_rows) * len(queries) / t_search, "configs0_search_us": t_s0 * 1e6,
            "configs0_top1_ids": [int(v) for v in I0[:, 0]], "threads": torch.get_num_threads(), "emb_dim": int(rows0.shape[1])}


def side_configs(dev, weights_b32):
    """The other BASELINE configs that fit one GPU, timed by this same process right after the headline (outside its timed
    region): one GPU's shard of configs[2] (1.25M rows x 1000 queries), configs[3] at full size (8 feeds
    of 1280x720 BGR -> rolling 5M-row index, hipGraph replay) and the configs[4] tower (ViT-L/14 bf16 / e4m3 modes)."""
    from ivr_amd import config as C
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.streaming import StreamingSession
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    out = {}

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    g = torch.Generator(device=dev).manual_seed(5678)
    rows = 5_000_000
    index = FlatIPIndex(512, capacity=rows, device=dev.index)
    for i in range(0, rows, 250_000):
        index.add(torch.randn((250_000, 512), generator=g, device=dev), normalize=True)
        if i + 250_000 == 1_250_000:                       # configs[2]: 10M rows over 8 GPUs = 1.25M rows per GPU, 1000 queries
            q = torch.randn((1000, 512), generator=g, device=dev)
            index.reserve_search(1000, 10)
            for _ in range(2):
                index.search_device(q, 10, normalize=True)
            ms = timed(lambda: index.search_device(q, 10, normalize=True), 5)
            # parity at full size on a sample: 64 of the 1000 queries against the float64 oracle over the rows the device holds
            from oracle import search_ref as S
            Dq, Iq = index.search_device(q, 10, normalize=True)
            pick = np.arange(0, 1000, 16)[:64]
            Xh = index.reconstruct_n(0, 1_250_000)
            qn = S.normalize_rows_core(q[pick].cpu().numpy()).astype(np.float32)
            Dr, Ir = S.flat_ip_search(Xh, qn, 10, dtype=np.float64)
            out["configs2_one_shard"] = {"rows": 1_250_000, "queries": 1000, "k": 10, "search_ms": ms,
                                         "pairs_per_s": 1_250_000 * 1000 / (ms * 1e-3),
                                         "ids_exact_on_sample_of_64_queries": bool(np.array_equal(Iq[pick].cpu().numpy(), Ir)),
                                         "max_abs_score_err_on_sample": float(np.abs(Dq[pick].cpu().numpy() - Dr).max())}
            del Xh
    cfg = C.CLIP_VIT_B32
    tower = Tower(cfg, weights_b32, max_batch=8, device=dev.index)
    queries = torch.from_numpy(np.random.default_rng(91011).standard_normal((10, 512), dtype=np.float32))
    frames = torch.randint(0, 256, (8, 720, 1280, 3), generator=g, device=dev, dtype=torch.uint8)
    st = {"rows": rows, "feeds": 8, "frames_per_step": 8, "frame": "1280x720 BGR, stretch-resized on the GPU", "queries": 10, "k": 10}
    for use_graph in (True, False):
        sess = StreamingSession(tower, index, 8, 720, 1280, queries, k=10, mode="stretch", bgr=True, use_graph=use_graph)
        for _ in range(5):
            sess.step(frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            sess.step(frames)
        torch.cuda.synchronize()
        st["graph_replay_ms" if use_graph else "plain_launch_ms"] = (time.perf_counter() - t0) / 100 * 1e3
    st["real_time_margin"] = (1000.0 / 30.0) / st["graph_replay_ms"]
    out["configs3_streaming_step"] = st
    del tower, sess
    # the reference's interactive path, one text query per call (system.py:733 -> core.py:1504 encode_text -> unified_index.py:480
    # search_vectors, k = 50): wall clock from Python, host buffers in and out, against this 5M-row index
    out["interactive_text_query"] = interactive_query(dev, index)
    del index
    torch.cuda.empty_cache()
    cfg = C.CLIP_VIT_L14
    wl = make_weights(cfg, 12)
    kpad = (3 * cfg.patch * cfg.patch + 63) // 64 * 64
    patches = (torch.randn((512 * (cfg.tokens - 1), kpad), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    emb = torch.empty((512, cfg.embed_dim), dtype=torch.float32, device=dev)
    l14 = {"frames_per_step": 512}
    for compute in ("bf16", "fp8", "fp8_mlp", "fp8_all"):
        tw = Tower(cfg, wl, max_batch=512, compute=compute, device=dev.index)
        for _ in range(2):
            tw.encode_patches(patches, 512, out=emb)
        ms = timed(lambda: tw.encode_patches(patches, 512, out=emb), 4)
        l14[compute] = {"ms_per_step": ms, "frames_per_s": 512 / (ms * 1e-3)}
        tw.close()
        del tw
        torch.cuda.empty_cache()
    l14["note"] = ("encoder only (patch-major pixels resident), random-init weights; fp8 = the preset inside the north-star bound (|score - f32 score| "
                   "<= 1e-3 on every pair): fc1+fc2 in e4m3 in the last third of the blocks + bf16 token-0 rows; fp8_mlp = the same in every block "
                   "(1 - cos <= 1e-3, text-query scores ~2e-3); fp8_all = all four sites in e4m3 (1 - cos ~ 4e-3): tests/test_fp8_gpu.py, "
                   "profiles/r02_fp8_error_budget.json")
    out["configs4_tower_vit_l14"] = l14
    return out


def interactive_query(dev, index, calls=100):
    """One text query -> top-50, the way system.search() issues it: CLIPFeatureExtractor.encode_text (ViT-B/32 text tower, random-init,
    byte-level stand-in tokenizer: 77 token ids as in the reference) and FlatIPIndex.search with a host query vector, median wall-clock
    milliseconds per call.  The reference logged 38 - 273 ms for the encoder alone on its CUDA box (logs/performance.log:2-7)."""
    from ivr_amd.compat import CLIPFeatureExtractor

    def med(fn):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))
    text = "a person riding a bicycle at night"
    out = {"rows": int(index.ntotal), "d": 512, "k": 50, "calls": calls}
    for tc in ("f32", "bf16"):
        ex = CLIPFeatureExtractor("openai/clip-vit-base-patch32", allow_random_init=True, max_batch=8, text_compute=tc)
        qv = ex.encode_text(text)
        out[f"encode_text_{tc}_ms"] = med(lambda: ex.encode_text(text))
        if tc == "f32":
            out["search_ms"] = med(lambda: index.search(qv, 50))
        out[f"text_to_top50_{tc}_ms"] = med(lambda: index.search(ex.encode_text(text), 50))
        del ex
    out["note"] = ("compat.CLIPFeatureExtractor runs text queries through the float32 text tower by default (text-vs-image scores within 1e-3 of "
                   "the float32 reference); reference log: 38 - 273 ms per text query for the encoder alone (logs/performance.log:2-7)")
    return out


def configs2_sharded(dev, rank, world, total_rows=10_000_000, nq=1000, k=10, reps=5):
    """BASELINE configs[2] as a STRONG-scaling leg, run at every world size: a 10M-row x 512-d index row-sharded over the ranks
    (10M / world rows each, Gaussian rows seeded per shard), a replicated 1,000-query batch, local exact top-10 with global ids, ONE
    all-gather of the per-shard (score, id) candidates and the k-way merge on every rank (replaces the peer fan-out + concat + sort of
    system.py:1715-1757 / :1744-1746).  Times are the MAX over ranks; pairs/s = 10M x 1000 / whole search."""
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.sharded import ShardedIndex, shard_bounds
    lo, hi = shard_bounds(total_rows, world)[rank]
    n = hi - lo
    index = FlatIPIndex(512, capacity=n, device=dev.index)
    g = torch.Generator(device=dev).manual_seed(777 + rank)
    for i in range(0, n, 250_000):
        index.add(torch.randn((min(250_000, n - i), 512), generator=g, device=dev), normalize=True)
    q = torch.from_numpy(np.random.default_rng(4242).standard_normal((nq, 512), dtype=np.float32)).to(dev)
    index.reserve_search(nq, k)
    sh = ShardedIndex(index, 512, merge="device")
    sh.sync_counts()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(2):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        sync()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3, out

    local_ms, _ = timed(lambda: index.search_device(q, k, normalize=True, id_base=sh.id_base))
    total_ms, (D, I) = timed(lambda: sh.search(q, k, normalize=True))
    # size-independent checks on the merged result: scores descend, ids are unique per query and inside the global range, every
    # rank holds the same answer; and this rank's own rows come back first when they are the queries
    ok = bool((D[:, :-1] >= D[:, 1:]).all()) and bool(((I >= 0) & (I < total_rows)).all())
    ok = ok and all(len(set(r)) == k for r in I[:16].cpu().tolist())
    mine = torch.from_numpy(index.reconstruct_n(0, 4)).to(dev)
    if world > 1:
        probe = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(probe, mine)
        probe = torch.cat(probe)
        sig = I.sum().reshape(1).clone()
        sigs = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        ok = ok and all(int(x) == int(sigs[0]) for x in sigs)
    else:
        probe = mine
    Dp, Ip = sh.search(probe, 1, normalize=False)
    bases = [b for b, _ in shard_bounds(total_rows, world)]
    want = [b + j for b in bases for j in range(4)]
    ok = ok and Ip.flatten().cpu().tolist() == want and float((Dp - 1).abs().max()) < 1e-5
    _, redone = index.scan_stats()
    res = {"rows_total": total_rows, "rows_per_rank": n, "ranks": world, "queries": nq, "k": k, "search_ms": local_ms,
           "allgather_merge_ms": max(0.0, total_ms - local_ms), "total_ms": total_ms, "pairs_per_s": total_rows * nq / (total_ms * 1e-3),
           "scaling": "strong", "allgather_bytes_per_rank": nq * k * 12, "checks_ok": bool(ok),
           "queries_redone_by_the_exact_pass_on_rank0": int(redone),
           "note": "local search timed alone, then search + all-gather + merge; both = max over ranks of the mean of %d repeats" % reps}
    del index, sh
    torch.cuda.empty_cache()
    return res


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(n_ranks, script, script_args, port=None):
    """The command `python bench.py --gpus N` turns itself into when it was started from a plain shell: one rank per
    GPU under torch.distributed.run, rendezvous on 127.0.0.1 (replaces the reference's hand-rolled peer fan-out,
    system.py:1715-1757 / api.py:877-925, by one launcher + one collective)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), script, *script_args]


def self_launch(n_ranks, script, script_args, env=None, visible_gpus=None):
    """Start the N ranks as a CHILD process (never exec: nothing in this process has touched the GPU yet, and the
    children are ordinary processes), relay rank 0's JSON line to stdout and return the launcher's exit code.
    With fewer visible GPUs than ranks the ranks share devices and rendezvous over gloo (rehearsal of the
    multi-rank control flow on a smaller box); with N GPUs visible the backend stays nccl (= RCCL over xGMI)."""
    import subprocess
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if visible_gpus is None:
        visible_gpus = torch.cuda.device_count()            # counting devices does not initialise the GPU
    if visible_gpus < n_ranks and "IVR_DIST_BACKEND" not in env:
        env["IVR_DIST_BACKEND"] = "gloo"
        print(f"bench.py: {visible_gpus} GPU(s) visible for {n_ranks} ranks - ranks share devices, backend gloo (rehearsal)",
              file=sys.stderr)
    cmd = launcher_command(n_ranks, script, script_args)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                json.loads(s)
                line = s
            except ValueError:
                pass
    other = [ln for ln in proc.stdout.splitlines() if ln.strip() != line]
    if other:
        print("\n".join(other), file=sys.stderr)
    if line is not None:
        print(line)
        sys.stdout.flush()
    elif proc.returncode == 0:
        print("bench.py: the ranks finished without printing a JSON line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)           # 25 x 4096 = 102,400 frames (configs[1]: 100k)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=4096)
    ap.add_argument("--index-rows", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=10)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side configurations (configs[2] shard, configs[3], configs[4] tower)")
    ap.add_argument("--lanes", type=int, default=1, help="split each step's frames over this many HIP streams / towers")
    # the defaults are BASELINE.json configs[1] (the metric's configuration); the two flags below select the
    # configs[4]-shaped variant (ViT-L/14, fp8 GEMMs, 768-d rows) as an additional measurement, never the headline
    ap.add_argument("--tower", choices=("b32", "l14"), default="b32")
    ap.add_argument("--compute", choices=("bf16", "fp8", "fp8_mlp", "fp8_all", "fp8_strict"), default="bf16")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started from a plain shell: become the launcher (before anything initialises the GPU in this process)
        raise SystemExit(self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    local_rank = local_rank % max(1, torch.cuda.device_count())   # rehearsal: more ranks than GPUs share devices
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # "nccl" is RCCL on ROCm (xGMI between the GPUs of the node); IVR_DIST_BACKEND=gloo only for rehearsing the
        # multi-rank control flow on a box with fewer GPUs than ranks
        backend = os.environ.get("IVR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ivr_amd import _ffi
    from ivr_amd import config as C
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.preprocess import preprocess_frames
    from ivr_amd.sharded import ShardedIndex
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights

    cfg = C.CLIP_VIT_B32 if args.tower == "b32" else C.CLIP_VIT_L14
    headline = args.tower == "b32" and args.compute == "bf16"
    d_emb, G2 = cfg.embed_dim, cfg.tokens - 1
    kpad = (3 * cfg.patch * cfg.patch + 63) // 64 * 64
    B, N, Q, k = args.frames_per_step, args.index_rows, args.queries, args.k
    weights = make_weights(cfg, 12)
    L = max(1, args.lanes)
    assert B % L == 0
    towers = [Tower(cfg, weights, max_batch=B // L, compute=args.compute, device=local_rank) for _ in range(L)]
    tower = towers[0]
    lanes = [torch.cuda.Stream(device=dev) for _ in range(L)] if L > 1 else [None]

    # synthetic inputs, generated on the device (a 100k-frame host array would be 15 GB)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    frame_tiles = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8) for _ in range(2)]
    index = FlatIPIndex(d_emb, capacity=N, device=local_rank)
    gi = torch.Generator(device=dev).manual_seed(5678 + rank)
    for i in range(0, N, 250_000):
        rows = torch.randn((min(250_000, N - i), d_emb), generator=gi, device=dev, dtype=torch.float32)
        index.add(rows, normalize=True)
        del rows
    queries = torch.from_numpy(np.random.default_rng(91011).standard_normal((Q, d_emb), dtype=np.float32)).to(dev)
    index.reserve_search(Q, k)
    sharded = ShardedIndex(index, d_emb, merge="device")
    sharded.sync_counts()

    patches = torch.empty((B * G2, kpad), dtype=torch.bfloat16, device=dev)
    emb = torch.empty((B, d_emb), dtype=torch.float32, device=dev)
    ring = {"pos": 0}
    ev_s0, ev_s1 = [], []

    def embed(lane, i, pos):
        b = B // L
        fr = frame_tiles[i & 1][lane * b:(lane + 1) * b]
        pt = patches[lane * b * G2:(lane + 1) * b * G2]
        em = emb[lane * b:(lane + 1) * b]
        preprocess_frames(fr, "identity", C.CLIP_MEAN, C.CLIP_STD, size=224, patch=cfg.patch, out=pt)
        towers[lane].encode_patches(pt, b, normalize=True, out=em)
        index.write_device(pos + lane * b, em)          # rows are already L2-normalised by the tower epilogue

    def step(i, timed):
        pos = ring["pos"]
        if L == 1:
            embed(0, i, pos)
        else:
            main = torch.cuda.current_stream()
            for lane in range(L):
                lanes[lane].wait_stream(main)
                with torch.cuda.stream(lanes[lane]):
                    embed(lane, i, pos)
            for lane in range(L):
                main.wait_stream(lanes[lane])
        ring["pos"] = (pos + B) % (N - B + 1) if N > B else 0
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        D, I = sharded.search(queries, k, normalize=True)
        if timed:
            e1.record()
            ev_s0.append(e0)
            ev_s1.append(e1)
        return D, I

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    _ffi.profile_reset(local_rank)
    _ffi.profile_enable(True, local_rank)
    t0 = time.perf_counter()
    for i in range(args.steps):
        D, I = step(i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    _ffi.profile_enable(False, local_rank)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = _ffi.profile_read(local_rank)
    search_ms = sum(a.elapsed_time(b) for a, b in zip(ev_s0, ev_s1)) / max(1, len(ev_s0))
    # the search alone, 50 times back to back with no per-kernel events, then 10 times with every launch of its tail
    # bracketed (profile level 2): the breakdown costs a few microseconds per event pair, so it stays out of the timed region
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        sharded.search(queries, k, normalize=True)
    e1.record()
    torch.cuda.synchronize()
    search_alone_ms = e0.elapsed_time(e1) / 50
    _ffi.profile_reset(local_rank)
    _ffi.profile_enable(2, local_rank)
    for _ in range(10):
        sharded.search(queries, k, normalize=True)
    torch.cuda.synchronize()
    _ffi.profile_enable(False, local_rank)
    search_kernels_us = {n: v["ms"] / v["launches"] * 1e3 for n, v in sorted(_ffi.profile_read(local_rank).items())}
    if world > 1:
        dist.barrier()
    c2 = None
    if headline and not args.no_extra:
        c2 = configs2_sharded(dev, rank, world)

    if rank == 0:
        frames_total = B * args.steps * world
        # HBM bytes per launch from the committed PMC passes of this same command (tools/pmc_aggregate.py); null when the
        # profile was taken at another batch size
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
                pmc = json.load(f)
        except OSError:
            pass
        pmc_ok = headline and B == pmc.get("_frames_per_step", -1) and N == 1_000_000

        def traffic(*kernels):
            if not pmc_ok or not kernels or not all(k in pmc for k in kernels):
                return None
            tot = sum(pmc[k]["hbm_bytes"] * pmc[k]["launches"] for k in kernels)
            return tot / sum(pmc[k]["launches"] for k in kernels)
        gemm = {n: v for n, v in prof.items() if n.startswith("gemm_")}
        gemm_flop = sum(v["work"] for v in gemm.values())
        gemm_ms = sum(v["ms"] for v in gemm.values())
        gemm_launches = sum(v["launches"] for v in gemm.values())
        gemm_tf = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms else 0.0
        # the peak the GEMM launches are priced against: dense bf16, or - in an e4m3 mode - the harmonic mix of the bf16 and fp8
        # (2 x) peaks weighted by the share of the FLOPs that actually run on the fp8 MFMA (the sites and blocks of the preset)
        mfma_peak = MFMA_BF16_PEAK_TF
        if args.compute != "bf16":
            T_, D_, M_, L_ = cfg.tokens, cfg.width, cfg.mlp, cfg.layers
            site = {1: 2.0 * T_ * D_ * 3 * D_, 2: 2.0 * T_ * D_ * D_, 4: 2.0 * T_ * D_ * M_, 8: 2.0 * T_ * D_ * M_}
            total = 2.0 * G2 * kpad * D_ + L_ * sum(site.values()) + 2.0 * D_ * d_emb
            f8 = sum(v for b, v in site.items() if tower.fp8_sites & b) * (L_ - tower.fp8_first_layer)
            mfma_peak = total / ((total - f8) / MFMA_BF16_PEAK_TF + f8 / (2 * MFMA_BF16_PEAK_TF))

        def hbm(name):
            v = prof.get(name)
            if not v or not v["ms"]:
                return None
            a = v["work"] / (v["ms"] * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                    "traffic": traffic(PMC_NAME[name]), "kernel": name, "avg_launch_ms": v["ms"] / v["launches"], "launches": int(v["launches"]),
                    "bytes_per_launch": v["work"] / v["launches"]}

        out = {
            "metric": "frames/s embedded (+ query-vs-index cosine pairs/s; top-10 recall vs CPU ref)",
            "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.compute, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: ViT-B/32 bf16 random-init embed of 224x224 RGB frames + 1M-row x 512-d "
                                    "fp32 index cosine top-10 (per GPU)") if headline else
                                   (f"NOT the headline configuration: {cfg.name} {args.compute} random-init embed of 224x224 RGB frames + "
                                    f"{N}-row x {d_emb}-d fp32 index cosine top-{k} (per GPU)"),
                       "frames_per_step_per_gpu": B, "frames_total": frames_total, "index_rows_per_gpu": N, "queries": Q, "k": k,
                       "parallelism": f"{world} x (frames + index rows sharded per GPU); one all-gather of (score,id) + merge"},
            "pairs_per_s": N * world * Q / (search_ms * 1e-3), "search_ms_per_step": search_ms,
            "search_alone": {"ms": search_alone_ms, "pairs_per_s": N * world * Q / (search_alone_ms * 1e-3),
                             "note": "50 searches back to back outside the timed region, no per-kernel events",
                             "per_launch_us_with_events": search_kernels_us},
            "roofline": {"bound": "mfma", "achieved": gemm_tf, "peak": mfma_peak, "unit": "TFLOP/s", "frac": gemm_tf / mfma_peak,
                         "traffic": traffic(*[k for k in pmc if k.startswith(("gemm_big_kernel<bf16", "gemm_pers_kernel", "gemm_kernel<bf16", "qkv_attn_kernel<bf16", "qkv_attn_pers_kernel<bf16"))]),
                         "kernel": ("gemm_big_kernel<bf16> / gemm_pers_kernel (fc1) / qkv_attn_pers_kernel<bf16> (QKV projection fused with attention) / gemm_kernel<bf16>" if args.compute == "bf16" else "gemm_big8_kernel (e4m3 sites) + the bf16 GEMMs of the other sites, patch embedding and projection")
                                   + " (all tower GEMM launches of the timed region)",
                         "avg_launch_ms": gemm_ms / max(1, gemm_launches), "launches": int(gemm_launches),
                         "flop_per_launch": gemm_flop / max(1, gemm_launches),
                         "by_call_site": {n: {"TFLOP/s": v["work"] / (v["ms"] * 1e-3) / 1e12, "ms": v["ms"] / v["launches"]}
                                          for n, v in sorted(gemm.items())}},
            # the index pass that actually streams the rows: the bf16 candidate scan when the index keeps one (the exact float32
            # scan then only runs for queries whose verification failed), else the float32 scan
            "roofline_search": hbm("scan16_groupmax" if "scan16_groupmax" in prof else "scan_groupmax"),
            "roofline_preprocess": hbm("preprocess_emit"),
            "kernel_ms_per_step": {n: v["ms"] / args.steps for n, v in sorted(prof.items())},
        }
        # recall / id check of the last search against the oracle (rank 0's shard; N == 1: the whole index)
        from oracle import search_ref as S
        Xh = index.reconstruct_n(0, N)
        qn = S.normalize_rows_core(queries.cpu().numpy()).astype(np.float32)
        Dl, Il = index.search_device(queries, k, normalize=True)
        Dr, Ir = S.flat_ip_search(Xh, qn, k, dtype=np.float64)
        Il, Dl = Il.cpu().numpy(), Dl.cpu().numpy()
        out["recall_at_10"] = float(np.mean([len(set(Il[q]) & set(Ir[q])) / k for q in range(Q)]))
        out["ids_exact"] = bool(np.array_equal(Il, Ir))
        out["max_abs_score_err"] = float(np.abs(Dl - Dr).max())
        # SURVEY.md section 8 hard part 2: recall with and without the re-rank.  "Without" = the top-k of the bf16 candidate
        # ranking alone (<bf16(row), q> in f32, emulated here with torch on the same rows; the library never reports these scores)
        Xd = torch.from_numpy(Xh).to(dev)
        approx = torch.from_numpy(qn).to(dev) @ Xd.to(torch.bfloat16).to(torch.float32).T
        Ia = torch.topk(approx, k, dim=1).indices.cpu().numpy()
        out["recall_at_10_candidates_only"] = float(np.mean([len(set(Ia[q]) & set(Ir[q])) / k for q in range(Q)]))
        out["ids_exact_candidates_only"] = bool(np.array_equal(Ia, Ir))
        del Xd, approx
        for key in ("roofline", "roofline_search", "roofline_preprocess"):
            if out.get(key):
                out[key]["traffic_source"] = (f"profiles/{PMC_FILE} (rocprofv3 --pmc passes of this same command, tools/pmc_aggregate.py; "
                                              "not re-measured in this run)") if pmc_ok else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, Xh[:100_000], qn, k)
        if world == 1 and headline and not args.no_extra:
            # free the headline's buffers first: the streaming index alone is 15 GB
            del frame_tiles, patches, emb, index, sharded, towers, tower, Xh
            torch.cuda.empty_cache()
            out["extra"] = side_configs(dev, weights)
        if c2 is not None:
            out.setdefault("extra", {})["configs2_sharded"] = c2
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
