#!/usr/bin/env python3
"""Per-site error budget of the e4m3 tower mode, on the CPU with oracle/quant_ref.py.

For every assignment of the four linear sites of a block {qkv, o (attn-out), fc1, fc2} to e4m3 (the rest bf16)
and every activation scaling, embed the same synthetic frames and report, against the float32 oracle:
  1 - cos(embedding, f32 embedding)                       (max over frames)
  max |score - f32 score| over (image + text query) x row pairs  (the north-star bound is 1e-3)
Text queries always come from the bf16 text tower (queries are few; the fp8 question is about the 50M rows).

python tools/fp8_error_budget.py [--tower l14|b32] [--frames 8] [--out profiles/r02_fp8_error_budget.json]
"""
import argparse
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ivr_amd import config as C  # noqa: E402
from ivr_amd.weights import make_weights  # noqa: E402
from oracle import preprocess_ref as P  # noqa: E402
from oracle import quant_ref as QR  # noqa: E402
from oracle import vit_ref as V  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tower", choices=("l14", "b32"), default="l14")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--image-queries", type=int, default=4)
    ap.add_argument("--text-queries", type=int, default=8)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="single sites + all four only")
    ap.add_argument("--outliers", action="store_true",
                    help="budget on the outlier-stressed weights of oracle/quant_ref.add_outliers (LN-2 gains x64, fc1 rows x2000)")
    args = ap.parse_args()
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    vcfg, tcfg = (C.CLIP_VIT_L14, C.CLIP_TEXT_L14) if args.tower == "l14" else (C.CLIP_VIT_B32, C.CLIP_TEXT_B32)
    wv, wt = make_weights(vcfg, 12), make_weights(tcfg, 13)
    if args.outliers:
        wv = QR.add_outliers(vcfg, wv)
    n = args.frames + args.image_queries
    frames = np.random.default_rng(1234).integers(0, 256, (n, 224, 224, 3), dtype=np.uint8)
    px = P.preprocess(frames, "identity", C.CLIP_MEAN, C.CLIP_STD)
    rng = np.random.default_rng(77)
    ids = rng.integers(1, tcfg.vocab - 2, (args.text_queries, 16)).astype(np.int64)
    for r in range(len(ids)):
        ids[r, rng.integers(4, 16):] = tcfg.eos_id

    t0 = time.time()
    ref = V.vision_forward(vcfg, wv, px)
    tref = V.text_forward(tcfg, wt, ids)
    print(f"f32 oracle: {time.time() - t0:.1f} s for {n} frames + {len(ids)} text queries", flush=True)
    tq = QR.text_forward(tcfg, wt, ids, QR.QuantSpec((), base="bf16"))
    Sref = np.concatenate([ref[args.frames:], tref]) @ ref[:args.frames].T

    def measure(spec):
        nonlocal tq
        emb = QR.vision_forward(vcfg, wv, px, spec)
        cos = (emb * ref).sum(1)
        S = np.concatenate([emb[args.frames:], tq]) @ emb[:args.frames].T
        d = np.abs(S - Sref)
        return {"one_minus_cos_max": float(1 - cos.min()), "one_minus_cos_mean": float(1 - cos.mean()),
                "dscore_max_image_q": float(d[:args.image_queries].max()), "dscore_max_text_q": float(d[args.image_queries:].max()),
                "dscore_max": float(d.max())}

    rows = []

    def run(label, spec):
        t1 = time.time()
        m = measure(spec)
        m["assignment"] = label
        rows.append(m)
        print(f"{label:44s} 1-cos max {m['one_minus_cos_max']:.2e}  |dscore| image-q {m['dscore_max_image_q']:.2e} "
              f"text-q {m['dscore_max_text_q']:.2e}   ({time.time() - t1:.0f} s)", flush=True)

    run("bf16 (all sites)", QR.QuantSpec((), base="bf16"))
    subsets = [(s,) for s in QR.SITES] + [QR.SITES]
    if not args.quick:
        subsets = [c for r in range(1, 5) for c in itertools.combinations(QR.SITES, r)]
    for scale in ("none", "row", "block32"):
        for sub in subsets:
            run(f"e4m3 {'+'.join(sub)} / act scale {scale}", QR.QuantSpec(sub, act_scale=scale))
    # e4m3 only in a prefix / suffix of the stack (all four sites)
    if not args.quick:
        L = vcfg.layers
        for lo, hi in ((0, L // 4), (0, L // 2), (L // 2, L), (3 * L // 4, L)):
            run(f"e4m3 all sites, layers [{lo},{hi}) / act scale row", QR.QuantSpec(QR.SITES, fp8_layers=range(lo, hi), act_scale="row"))
    # the bf16 side path for the token-0 rows of the MLP sites (what the kernels implement: tower.hip run_layers)
    for sub in (("fc1",), ("fc2",), ("fc1", "fc2"), QR.SITES):
        spec = QR.QuantSpec(sub, keep_rows=(0,))
        spec.keep_sites = {"fc1", "fc2"}
        run(f"e4m3 {'+'.join(sub)} / token-0 rows of fc1, fc2 in bf16", spec)
    # ... and only from block `first` on (tower desc fp8_first_layer): compute="fp8" (old name "fp8_strict") is first = 2L/3
    if not args.quick:
        L = vcfg.layers
        for first in (L // 3, L // 2, (2 * L) // 3, (5 * L) // 6):
            spec = QR.QuantSpec(("fc1", "fc2"), fp8_layers=range(first, L), keep_rows=(0,))
            spec.keep_sites = {"fc1", "fc2"}
            run(f"e4m3 fc1+fc2 in blocks [{first},{L}) / token-0 rows in bf16", spec)
        # text queries from the float32 text tower instead of the bf16 one (queries are few)
        tq_saved = tq
        tq = tref
        for label, spec in (("bf16 (all sites), f32 text tower", QR.QuantSpec((), base="bf16")),
                            (f"e4m3 fc1+fc2 in blocks [{(2 * L) // 3},{L}) / token-0 rows in bf16, f32 text tower",
                             QR.QuantSpec(("fc1", "fc2"), fp8_layers=range((2 * L) // 3, L), keep_rows=(0,))),
                            ("e4m3 fc1+fc2 / token-0 rows in bf16, f32 text tower", QR.QuantSpec(("fc1", "fc2"), keep_rows=(0,)))):
            spec.keep_sites = {"fc1", "fc2"}
            run(label, spec)
        tq = tq_saved
    out = {"tower": vcfg.name, "frames": args.frames, "image_queries": args.image_queries, "text_queries": args.text_queries,
           "bound": "north_star: |score - f32 score| <= 1e-3", "rows": rows,
           "method": "oracle/quant_ref.py: operand rounding emulated on the CPU, float32 accumulation"}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
