"""GPU: Laplacian-variance blur score and Canny edge density (filter.py:63-92) vs oracle/quality_ref.py.
PARITY UNPINNED: OpenCV is absent here and the reference holds no fixture, so both sides restate the published operators;
what is pinned is that the HIP kernels equal the restatement exactly: the edge count is an integer and must match bit for
bit, the variance comes from exact integer sums and must match numpy's float64 .var() to rounding."""
import numpy as np
import pytest

from conftest import smooth_frames, synth_frames
from oracle import quality_ref as Q

pytestmark = pytest.mark.gpu


def _check(frames, bgr=False, low=20, high=80):
    from ivr_amd.quality import frame_quality_scores
    got = frame_quality_scores(frames, bgr=bgr, canny_low=low, canny_high=high)
    for f, g in zip(frames, got):
        ref = Q.quality_scores(f, bgr=bgr, low=low, high=high)
        h, w = f.shape[:2]
        assert round(g["edge_density"] * h * w / 100) == round(ref["edge_density"] * h * w / 100), (g, ref)
        assert abs(g["blur_score"] - ref["blur_score"]) <= 1e-9 * max(1.0, ref["blur_score"]), (g, ref)
    return got


@pytest.mark.parametrize("h,w", [(224, 224), (120, 160), (37, 53), (360, 640), (3, 5), (1, 1)])
def test_quality_scores_match_oracle(h, w):
    smooth = smooth_frames(60 + h, 3, h, w)
    noise = synth_frames(61 + w, 2, h, w)
    _check(np.concatenate([smooth, noise]))


def test_bgr_thresholds_and_long_chains():
    rng = np.random.default_rng(3)
    frames = smooth_frames(5, 2, 200, 300)
    _check(frames, bgr=True)
    _check(frames, low=5, high=200)
    _check(frames, low=0, high=0)
    # a one-pixel-wide spiral of weak gradient tied to one strong spot: the hysteresis has to walk the whole chain
    img = np.full((160, 160), 100, np.uint8)
    y, x, step, d = 10, 10, 140, 0
    while step > 8:
        for _ in range(step):
            img[y, x] = 112
            y, x = y + (0, 1, 0, -1)[d], x + (1, 0, -1, 0)[d]
        d = (d + 1) % 4
        step -= 6 if d % 2 == 0 else 0
    img[10:14, 10:14] = 255
    f = np.repeat(img[..., None], 3, axis=2)[None]
    got = _check(f, low=20, high=300)
    assert got[0]["edge_density"] > 2.0
    blurry = np.full((1, 64, 64, 3), 77, np.uint8)
    sc = _check(blurry)
    assert sc[0] == {"blur_score": 0.0, "edge_density": 0.0}
    del rng


def test_gating_pipeline_matches_reference_logic(tmp_path):
    """filter.py:92-140 end to end on files: scores by path, percentile thresholds, adaptive / fixed acceptance."""
    from PIL import Image
    from ivr_amd import quality as GQ
    frames = np.concatenate([smooth_frames(1, 6, 96, 128), synth_frames(2, 2, 96, 128),
                             np.full((2, 96, 128, 3), 128, np.uint8)])
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / f"{i:03d}.png"
        Image.fromarray(f).save(p)
        paths.append(str(p))
    scores = GQ.quality_scores_for_paths(paths + [str(tmp_path / "missing.png")])
    assert scores[-1] == {"blur_score": 0.0, "edge_density": 0.0}
    for f, s in zip(frames, scores):
        ref = Q.quality_scores(f)
        assert abs(s["blur_score"] - ref["blur_score"]) <= 1e-9 * max(1.0, ref["blur_score"])
        assert abs(s["edge_density"] - ref["edge_density"]) < 1e-9
    assert GQ.calculate_blur_score(paths[0]) == scores[0]["blur_score"] and GQ.calculate_edge_density(paths[7]) == scores[7]["edge_density"]
    cfg = {"blur_percentile": 10.0, "edge_percentile": 10.0, "enable_blur_detection": True, "enable_edge_detection": True,
           "blur_threshold": 10.0, "edge_threshold": 5.0}
    cfg30 = dict(cfg, blur_percentile=30.0, edge_percentile=30.0)
    bt, et = GQ.determine_adaptive_thresholds(scores[:-1], cfg30)
    ref_scores = [Q.quality_scores(f) for f in frames]
    assert np.isclose(bt, np.percentile([r["blur_score"] for r in ref_scores], 30.0)) and bt > 0
    verdicts = [GQ.is_frame_acceptable_adaptive(s, bt, et, cfg30) for s in scores[:-1]]
    assert verdicts[-1] == (False, "blur") and verdicts[-2] == (False, "blur") and verdicts[6] == (True, "acceptable")   # flat frames go, noise stays
    # the reference's default 10th percentile over these ten frames is 0 (two flat frames): `score < 0` rejects nothing
    bt10, et10 = GQ.determine_adaptive_thresholds(scores[:-1], cfg)
    assert bt10 == 0.0 and all(GQ.is_frame_acceptable_adaptive(s, bt10, et10, cfg)[0] for s in scores[:-1])
    assert GQ.is_frame_acceptable_fixed(scores[8], cfg) == (False, "blur")
    assert GQ.is_frame_acceptable_fixed(scores[6], cfg) == (True, "acceptable")
    assert GQ.determine_adaptive_thresholds([], cfg) == (None, None)


def _ridge_frame(h, w, rows, strong_at):
    """Weak one-pixel-high horizontal ridges over the whole width (a run of candidates hundreds of 64-pixel words long) with one
    strong spot each: the hysteresis has to carry the seed along the run across words, lanes and - past 4,096 pixels - across the
    per-lane word chunks, in the direction the spot's position asks for."""
    img = np.full((h, w), 100, np.uint8)
    for r, sx in zip(rows, strong_at):
        img[r, :] = 106
        img[r, sx:sx + 3] = 255
    return np.repeat(img[..., None], 3, axis=2)


@pytest.mark.parametrize("h,w", [(9, 700), (12, 4200), (10, 9000), (7, 16384)])
def test_long_runs_cross_words_and_chunks(h, w):
    """One wave holds a whole row of the candidate plane, 1, 2 or 4 words per lane: seeds at the far left, the far right and the
    middle of runs as wide as the frame, plus noise frames of the same shape."""
    frames = np.stack([_ridge_frame(h, w, [2, 6], [0, w - 3]), _ridge_frame(h, w, [3, 5], [w // 2, 64 * (w // 128) - 1]),
                       synth_frames(h + w, 1, h, w)[0]])
    got = _check(frames, low=20, high=300)
    assert got[0]["edge_density"] * h * w / 100 >= 2 * (w - 8)          # both ridges were followed to their ends


def test_frames_wider_than_the_row_limit_are_refused():
    import torch
    from ivr_amd.quality import frame_quality_scores
    with pytest.raises(Exception, match="wider"):
        frame_quality_scores(torch.zeros((1, 2, 16385, 3), dtype=torch.uint8, device="cuda"))


def test_vertical_serpentine_crosses_every_band_many_times():
    """A three-pixel weak line that runs the frame's height up and down twenty times, tied to a single strong spot: every leg crosses
    all the bands the sweep kernel cuts the frame into, and each reversal costs it another round."""
    h, w, lw = 400, 330, 3
    img = np.full((h, w), 100, np.uint8)
    cols = list(range(12, w - 12 - lw, 15))
    for i, x in enumerate(cols):
        img[12:h - 12, x:x + lw] = 112
        if i + 1 < len(cols):
            if i % 2 == 0:
                img[h - 12 - lw:h - 12, x:cols[i + 1] + lw] = 112
            else:
                img[12:12 + lw, x:cols[i + 1] + lw] = 112
    img[12:16, 10:15] = 255
    f = np.repeat(img[..., None], 3, axis=2)[None]
    got = _check(f, low=20, high=300)
    assert got[0]["edge_density"] * h * w / 100 > 1.5 * len(cols) * (h - 24)         # both flanks of every leg were reached


def test_random_shapes_thresholds_and_channel_orders():
    """Forty seeded draws of (h, w, low, high, bgr): widths around the 64-pixel word and tile edges, heights around the 32-row tile
    and the 16-row band edges, thresholds from 'everything is an edge' to 'nothing is'."""
    rng = np.random.default_rng(2024)
    widths = [1, 2, 3, 31, 62, 63, 64, 65, 66, 67, 126, 127, 128, 129, 130, 191, 193, 255, 257]
    heights = [1, 2, 15, 16, 17, 30, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 66, 95, 97]
    for i in range(40):
        h, w = int(rng.choice(heights)), int(rng.choice(widths))
        low = int(rng.choice([0, 5, 20, 60, 200, 2040]))
        high = low + int(rng.choice([0, 1, 40, 300]))
        smooth = smooth_frames(1000 + i, 1, h, w)
        noise = synth_frames(2000 + i, 1, h, w)
        mix = ((smooth.astype(np.int32) * 3 + noise.astype(np.int32)) // 4).astype(np.uint8)
        _check(np.concatenate([smooth, noise, mix]), bgr=bool(i & 1), low=low, high=high)


@pytest.mark.parametrize("h,w,n", [(1080, 1920, 3), (2160, 3840, 2), (1081, 1923, 2)])
def test_full_hd_and_4k_frames(h, w, n):
    """Real frame sizes (30 and 60 words per row, 34 and 68 tile rows, 16 sweep bands), one of them ragged in both directions: smooth
    content has contours that span the frame, the noise frame has an edge in every tile."""
    frames = np.concatenate([smooth_frames(7 + h, n - 1, h, w), synth_frames(8 + w, 1, h, w)])
    _check(frames)
