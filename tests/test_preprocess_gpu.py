"""GPU parity: HIP preprocessing (through the C ABI) vs the oracle, which is pinned bit-exactly to
PIL + CLIPImageProcessorPil by tests/golden/preprocess.npz.  Bar: float32 output bit-exact, bf16 output
equal to the round-to-nearest-even of the float32 reference, uint8 geometry bit-exact."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN, smooth_frames, synth_frames
from ivr_amd import config as C
from oracle import preprocess_ref as P

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "golden.json")))["preprocess"]


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("case", META, ids=lambda c: f"{c['mode']}-{c['h']}x{c['w']}")
def test_golden_cases_bit_exact(case, golden):
    from ivr_amd.preprocess import preprocess_frames
    g = golden("preprocess")
    h, w = case["h"], case["w"]
    frames = np.concatenate([synth_frames(case["seeds"][0], 1, h, w), smooth_frames(case["seeds"][1], 1, h, w)])
    mean, std = (C.IMAGENET_MEAN, C.IMAGENET_STD) if case["bgr"] else (C.CLIP_MEAN, C.CLIP_STD)
    out = preprocess_frames(frames, case["mode"], mean, std, bgr=case["bgr"], out_dtype=torch.float32).cpu().numpy()
    ref = P.preprocess(frames, case["mode"], mean, std, bgr=case["bgr"])
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    for fi in range(2):   # and against the committed HF/PIL checksums directly
        assert np.uint32(zlib.crc32(out[fi].tobytes())) == g[f"c{case['case']}_f{fi}_f32crc"]
    b = preprocess_frames(frames, case["mode"], mean, std, bgr=case["bgr"])
    assert np.array_equal(_bits(b), P.to_bf16_bits(ref))


@pytest.mark.parametrize("mode,h,w", [("letterbox", 360, 640), ("letterbox", 640, 360), ("letterbox", 224, 224),
                                      ("shortest_edge_crop", 224, 300), ("shortest_edge_crop", 500, 224),
                                      ("stretch", 224, 500), ("stretch", 37, 224), ("shortest_edge_crop", 32, 32),
                                      ("stretch", 2160, 3840)])
def test_more_geometry_vs_oracle(mode, h, w):
    from ivr_amd.preprocess import preprocess_frames
    frames = synth_frames(h * 10000 + w, 3, h, w) if h * w < 3_000_000 else smooth_frames(5, 1, h, w)
    out = preprocess_frames(frames, mode, out_dtype=torch.float32).cpu().numpy()
    ref = P.preprocess(frames, mode, C.CLIP_MEAN, C.CLIP_STD)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_random_sizes_modes_and_output_sizes_vs_oracle():
    """24 seeded draws of input size (17..900 px per side), geometry mode, BGR flag and output size: float32 output bit-exact."""
    from ivr_amd.preprocess import preprocess_frames
    rng = np.random.default_rng(20251005)
    for i in range(24):
        h, w = int(rng.integers(17, 900)), int(rng.integers(17, 900))
        mode = str(rng.choice(["stretch", "shortest_edge_crop", "letterbox"]))
        bgr, size = bool(rng.integers(2)), int(rng.choice([112, 224, 336]))
        frames = synth_frames(1000 + i, 2, h, w) if i % 2 else smooth_frames(1000 + i, 2, h, w)
        out = preprocess_frames(frames, mode, C.CLIP_MEAN, C.CLIP_STD, bgr=bgr, size=size, out_dtype=torch.float32)      # NCHW
        ref = P.preprocess(frames, mode, C.CLIP_MEAN, C.CLIP_STD, bgr=bgr, size=size)
        got = out.cpu().numpy()
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (i, h, w, mode, bgr, size)


def test_bilinear_filter_vs_oracle():
    from ivr_amd.preprocess import preprocess_frames
    frames = synth_frames(77, 2, 300, 400)
    out = preprocess_frames(frames, "stretch", out_dtype=torch.float32, bilinear=True).cpu().numpy()
    ref = P.preprocess(frames, "stretch", C.CLIP_MEAN, C.CLIP_STD, filt="bilinear")
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("patch", [32, 16, 14])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_patch_major_layout(patch, dtype):
    from ivr_amd.preprocess import preprocess_frames
    frames = synth_frames(99, 5, 224, 224)
    out = preprocess_frames(frames, "identity", patch=patch, out_dtype=dtype)
    ref = P.patch_major(P.preprocess(frames, "identity", C.CLIP_MEAN, C.CLIP_STD), patch)
    K = 3 * patch * patch
    assert out.shape == (5 * (224 // patch) ** 2, -(-K // 64) * 64)
    if dtype == torch.float32:
        o = out.cpu().numpy()
        assert np.array_equal(o[:, :K], ref) and (o[:, K:] == 0).all()
    else:
        o = _bits(out)
        assert np.array_equal(o[:, :K], P.to_bf16_bits(ref)) and (o[:, K:] == 0).all()


def test_custom_mean_std_uses_exact_table():
    """mean/std for which a single fma is not bit-exact must fall back to the tabulated values."""
    from ivr_amd.preprocess import preprocess_frames
    frames = synth_frames(3, 2, 224, 224)
    for mean, std in [((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)), ((0.1234567, 0.7654321, 0.3333333), (0.777, 0.0123, 1.5))]:
        out = preprocess_frames(frames, "identity", mean, std, out_dtype=torch.float32).cpu().numpy()
        ref = P.preprocess(frames, "identity", mean, std)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        b = preprocess_frames(frames, "identity", mean, std)
        assert np.array_equal(_bits(b), P.to_bf16_bits(ref))


def test_bad_arguments():
    from ivr_amd.preprocess import preprocess_frames
    with pytest.raises(ValueError):
        preprocess_frames(synth_frames(1, 1, 100, 100), "identity")
    with pytest.raises(ValueError):
        preprocess_frames(synth_frames(1, 1, 224, 224), "identity", std=(1, 0, 1))
    with pytest.raises(ValueError):
        preprocess_frames(np.zeros((1, 224, 224, 4), np.uint8))
    assert preprocess_frames(np.zeros((0, 224, 224, 3), np.uint8)).shape == (0, 3, 224, 224)


def test_two_threads_two_streams_different_sizes():
    """the resample scratch is per stream and outgrown blocks stay alive, so two host threads preprocessing
    different (growing) frame sizes on their own streams - a CLIP extractor and a DINO filter side by side, as the
    reference's thread pool would run them (unified_index.py:773) - must each get bit-exact results."""
    import threading
    from ivr_amd.preprocess import preprocess_frames
    sizes = {0: [(120, 90), (360, 640), (300, 400), (720, 1280)], 1: [(640, 360), (200, 200), (1080, 1920), (90, 160)]}
    modes = {0: "shortest_edge_crop", 1: "stretch"}
    errors = []

    def work(tid):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for rep in range(3):
                    for (h, w) in sizes[tid]:
                        n = 2 + (rep + tid) % 3
                        frames = synth_frames(100 * tid + h + rep, n, h, w)
                        out = preprocess_frames(frames, modes[tid], out_dtype=torch.float32)
                        stream.synchronize()
                        ref = P.preprocess(frames, modes[tid], C.CLIP_MEAN, C.CLIP_STD)
                        if not np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32)):
                            errors.append((tid, rep, h, w))
        except Exception as e:                       # pragma: no cover - reported below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
