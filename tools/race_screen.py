#!/usr/bin/env python3
"""Race screen for the hand-synchronised kernels (counted vmcnt + raw barriers): every launch of the same problem must
reproduce the first result bit for bit, under memory load from a concurrent copy stream.  A DMA that is read one barrier
too early shows up here as a rare mismatch long before it shows up in a tolerance test."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd"))
import torch  # noqa: E402

from ivr_amd import config as C  # noqa: E402
from ivr_amd.linear import EPI_RESID, EPI_STORE, linear, linear_fp8, quantize_rows_e4m3  # noqa: E402
from ivr_amd.tower import Tower  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
os.environ["IVR_GEMM"] = "4"                     # the 256 x 256 kernels for every shape
side = torch.cuda.Stream()
junk_a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
junk_b = torch.empty_like(junk_a)
bad = 0
shapes = [(65536, 768, 3072), (65536, 2304, 768), (4096, 768, 768), (2500, 2304, 768), (3000, 768, 3072), (777, 1024, 4096), (5000, 3072, 768), (300, 320, 128), (256, 256, 256)]
for (m, n, k) in shapes:
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn((m, k), generator=g, device="cuda") * 0.7).to(torch.bfloat16)
    w = (torch.randn((n, k), generator=g, device="cuda") * k ** -0.5).to(torch.bfloat16)
    b = torch.randn(n, generator=g, device="cuda")
    r0 = torch.randn((m, n), generator=g, device="cuda")
    x8 = x.float().to(torch.float8_e4m3fn)
    w8, ws = quantize_rows_e4m3(w.float())
    ref = {}
    for it in range(iters if m < 20000 else max(8, iters // 8)):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a, non_blocking=True)
        os.environ["IVR_GEMM_PERS"] = "0"
        outs = {"store": linear(x, w, b, act=0), "resid": linear(x, w, b, epilogue=EPI_RESID, resid=r0.clone())}
        if n % 64 == 0:
            # the persistent kernel (round 3: flattened stage sequence across tiles, LDS-free epilogues, staggered start) must give the
            # tile-per-workgroup kernel's bits on every launch
            os.environ["IVR_GEMM_PERS"] = "2"
            os.environ["IVR_GEMM_STAGGER"] = str(it % 4)
            outs["store_persistent"] = linear(x, w, b, act=0)
            outs["resid_persistent"] = linear(x, w, b, epilogue=EPI_RESID, resid=r0.clone())
            os.environ["IVR_GEMM_PERS"] = "0"
            if it == 0:
                assert torch.equal(outs["store_persistent"], outs["store"]) and torch.equal(outs["resid_persistent"], outs["resid"])
        if k % 128 == 0 and n % 64 == 0:
            outs["fp8"] = linear_fp8(x8, w8, ws, b)
            outs["fp8_resid"] = linear_fp8(x8, w8, ws, b, epilogue=EPI_RESID, resid=r0.clone())
            outs["fp8_out8"] = linear_fp8(x8, w8, ws, b, act=0, out_fp8=True).view(torch.uint8)
        for name, o in outs.items():
            if it == 0:
                ref[name] = o.clone()
            elif not torch.equal(o, ref[name]):
                bad += 1
                print(f"MISMATCH {name} M={m} N={n} K={k} iteration {it}: {(o.float() - ref[name].float()).abs().max().item()}")
    torch.cuda.synchronize()
    print(f"gemm M={m} N={n} K={k}: {it + 1} launches x {len(outs)} variants reproduced" if not bad else f"gemm M={m} N={n} K={k}: mismatches so far {bad}")
os.environ.pop("IVR_GEMM_PERS", None)
os.environ.pop("IVR_GEMM_STAGGER", None)
# one-query products (gemm_skinny_kernel: weight panel by LDS-DMA behind one vmcnt(0) + barrier, activations by counted register loads)
os.environ.pop("IVR_GEMM", None)
for (m, n, k, dt) in ((77, 512, 2048, torch.float32), (77, 2048, 512, torch.bfloat16), (50, 768, 3072, torch.bfloat16), (128, 1024, 4096, torch.float32),
                      (1, 64, 8192, torch.bfloat16)):
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn((m, k), generator=g, device="cuda") * 0.7).to(dt)
    w = (torch.randn((n, k), generator=g, device="cuda") * k ** -0.5).to(dt)
    b = torch.randn(n, generator=g, device="cuda")
    r0 = torch.randn((m, n), generator=g, device="cuda")
    ref = None
    for it in range(iters):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a, non_blocking=True)
        outs = (linear(x, w, b, act=0), linear(x, w, b, epilogue=EPI_RESID, resid=r0.clone()))
        if ref is None:
            os.environ["IVR_GEMM_SKINNY"] = "0"
            ref = (linear(x, w, b, act=0), linear(x, w, b, epilogue=EPI_RESID, resid=r0.clone()))       # the tiled kernel's bits
            os.environ.pop("IVR_GEMM_SKINNY")
        if not (torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])):
            bad += 1
            print(f"MISMATCH skinny M={m} N={n} K={k} {dt} iteration {it}")
    torch.cuda.synchronize()
    print(f"skinny gemm M={m} N={n} K={k} {str(dt).split('.')[-1]}: {iters} launches equal to the tiled kernel")
# the large-batch candidate scan (search_scanq.hip: persistent workgroups, DMA cursors across work items, counted vmcnt)
from ivr_amd.index import FlatIPIndex  # noqa: E402
# ... and the small-batch scan through per-wave LDS-DMA rings (scan16_ring_kernel: at most 16 queries, d <= 512)
for (rows, d, nq, k) in ((300_001, 512, 300, 10), (120_000, 768, 1000, 5), (70_000, 96, 65, 50), (1_000_003, 512, 10, 10), (200_000, 384, 16, 5),
                         (90_001, 250, 1, 1)):
    g = torch.Generator(device="cuda").manual_seed(rows + nq)
    idx = FlatIPIndex(d, capacity=rows)
    idx.add(torch.randn((rows, d), generator=g, device="cuda"), normalize=True)
    q = torch.randn((nq, d), generator=g, device="cuda")
    ref = None
    for it in range(max(10, iters // 3)):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a, non_blocking=True)
        D, I = idx.search_device(q, k, normalize=True)
        if ref is None:
            ref = (D.clone(), I.clone())
        elif not (torch.equal(D, ref[0]) and torch.equal(I, ref[1])):
            bad += 1
            print(f"MISMATCH search rows={rows} d={d} nq={nq} iteration {it}")
    torch.cuda.synchronize()
    print(f"{'large-batch' if nq > 64 else 'ring-scan'} search {rows} x {d}, {nq} queries: {it + 1} launches reproduced")
    idx.close()
for name, batch in (("l14", 24), ("dino", 64), ("b32", 256)):
    cfg = {"b32": C.CLIP_VIT_B32, "l14": C.CLIP_VIT_L14, "dino": C.DINO_VIT_S16}[name]
    tw = Tower(cfg, make_weights(cfg, 3), max_batch=batch)
    frames = torch.randint(0, 256, (batch, cfg.image, cfg.image, 3), device="cuda", dtype=torch.uint8)
    ref = None
    for it in range(max(10, iters // 10)):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a, non_blocking=True)
        o = tw.encode_frames(frames)
        if ref is None:
            ref = o.clone()
        elif not torch.equal(o, ref):
            bad += 1
            print(f"MISMATCH tower {name} iteration {it}: {(o - ref).abs().max().item()}")
    print(f"tower {name} batch {batch}: reproduced")
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
