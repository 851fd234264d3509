"""CPU suite for the oracle files added in round 2 and the host logic above them (no GPU, no compute calls into the library):
  * oracle/quality_ref.py on patterns whose Laplacian variance and Canny edge maps can be worked out by hand
    (the file is "parity unpinned" against OpenCV - these checks pin it to the operators' definitions);
  * ivr_amd/quality.py gating logic (filter.py:102-140 restated);
  * oracle/quant_ref.py: the operand-rounding emulation reduces to the float32 oracle / the plain bf16 emulation where it must,
    and its e4m3 rounding agrees with the library's host quantiser (ivr_quantize_e4m3_host, the one the towers upload with)."""
import ctypes as C

import numpy as np
import torch

from conftest import synth_frames
from ivr_amd import _ffi
from ivr_amd import config as CFG
from ivr_amd.weights import make_weights
from oracle import preprocess_ref as P
from oracle import quality_ref as Q
from oracle import quant_ref as QR
from oracle import vit_ref as V


# ---- quality_ref -----------------------------------------------------------------------------------------------------------
def test_gray_is_the_14_bit_fixed_point_luma():
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    want = [(255 * 16384 + 8192) >> 14, 0, (255 * 4899 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 1868 + 8192) >> 14,
            (10 * 4899 + 20 * 9617 + 30 * 1868 + 8192) >> 14]
    assert Q.to_gray(px)[0].tolist() == want
    assert Q.to_gray(px[..., ::-1], bgr=True)[0].tolist() == want           # the same pixels in cv2 order


def test_laplacian_variance_of_known_patterns():
    flat = np.full((9, 11), 50, np.uint8)
    assert Q.laplacian_var(flat) == 0.0
    # one bright pixel of height a in the interior: responses -4a at it, +a at its four neighbours, 0 elsewhere
    a, h, w = 40, 7, 9
    spike = np.zeros((h, w), np.uint8)
    spike[3, 4] = a
    n = h * w
    mean = 0.0                                   # -4a + 4a
    assert abs(Q.laplacian_var(spike) - (16 * a * a + 4 * a * a) / n - mean) < 1e-12
    # a vertical step: only the two columns at the step respond (+-d), BORDER_REFLECT_101 keeps the rim rows like the interior
    d = 30
    step = np.zeros((6, 8), np.uint8)
    step[:, 4:] = d
    assert abs(Q.laplacian_var(step) - (2 * 6 * d * d) / 48) < 1e-12


def test_canny_on_a_step_edge_and_thresholds():
    d = 60                                                                    # Sobel response at the step: 4 d = 240 on both sides of it
    img = np.zeros((12, 16), np.uint8)
    img[:, 8:] = d
    e = Q.canny(img, 20, 80)
    # non-maximum suppression keeps ONE of the two equal columns: m > left neighbour and m >= right neighbour -> the left one
    assert e[:, 7].tolist() == [255] * 12 and int((e > 0).sum()) == 12
    assert int((Q.canny(img, 20, 240) > 0).sum()) == 0                       # strong needs m > high: 240 > 240 fails, no seed
    assert int((Q.canny(img, 20, 239) > 0).sum()) == 12
    assert int((Q.canny(img, 240, 400) > 0).sum()) == 0                      # candidates need m > low
    # hysteresis: a weak ridge (4 * 8 = 32) connected to a strong stretch (4 * 60) of the same edge survives, alone it does not
    weak = np.zeros((12, 16), np.uint8)
    weak[:, 8:] = 8
    assert int((Q.canny(weak, 20, 80) > 0).sum()) == 0
    both = weak.copy()
    both[:4, 8:] = d
    kept = Q.canny(both, 20, 80)
    assert kept[:, 7].tolist()[:3] == [255] * 3 and kept[6:, 7].tolist() == [255] * 6      # the weak part below is pulled in
    sc = Q.quality_scores(np.repeat(img[..., None], 3, axis=2))
    assert abs(sc["edge_density"] - 12 / (12 * 16) * 100) < 1e-12 and sc["blur_score"] > 0


def test_gating_logic_matches_filter_py():
    from ivr_amd import quality as G
    scores = [{"blur_score": float(b), "edge_density": float(e)} for b, e in [(0, 0), (5, 1), (50, 4), (500, 9), (5000, 30)]]
    cfg = {"blur_percentile": 40.0, "edge_percentile": 40.0, "enable_blur_detection": True, "enable_edge_detection": True,
           "blur_threshold": 10.0, "edge_threshold": 5.0}
    bt, et = G.determine_adaptive_thresholds(scores, cfg)
    assert bt == np.percentile([0, 5, 50, 500, 5000], 40.0) and et == np.percentile([0, 1, 4, 9, 30], 40.0)
    got = [G.is_frame_acceptable_adaptive(s, bt, et, cfg) for s in scores]
    assert got == [(False, "blur"), (False, "blur"), (True, "acceptable"), (True, "acceptable"), (True, "acceptable")]
    assert [G.is_frame_acceptable_fixed(s, cfg)[1] for s in scores] == ["blur", "blur", "low_edge", "acceptable", "acceptable"]
    off = dict(cfg, enable_blur_detection=False)
    assert G.is_frame_acceptable_fixed(scores[1], off) == (False, "low_edge")
    assert G.is_frame_acceptable_adaptive(scores[0], None, None, cfg) == (True, "acceptable")
    assert G.determine_adaptive_thresholds([], cfg) == (None, None)


# ---- quant_ref -------------------------------------------------------------------------------------------------------------
def test_e4m3_rounding_agrees_with_the_library_quantiser():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(50_000) * 2, rng.standard_normal(20_000) * 1e-2, rng.uniform(-500, 500, 20_000)]).astype(np.float32)
    lib = _ffi.load()
    codes = np.empty(x.shape, np.uint8)
    _ffi.check(lib.ivr_quantize_e4m3_host(x.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p), x.size), "quantize")
    lib_vals = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(QR.e4m3_round(torch.from_numpy(x)).numpy(), lib_vals)
    w = torch.from_numpy(rng.standard_normal((7, 64)).astype(np.float32))
    wq, ws = QR.quant_weight_e4m3(w)
    assert float(wq.abs().max()) == 448.0 and np.allclose((wq * ws).numpy(), w.numpy(), rtol=0.07, atol=1e-3)


def test_emulation_reduces_to_its_limits():
    cfg = CFG.TINY_VIT
    w = make_weights(cfg, 11)
    px = P.preprocess(synth_frames(3, 3, 224, 224), "identity", CFG.CLIP_MEAN, CFG.CLIP_STD)
    ref = V.vision_forward(cfg, w, px)
    f32 = QR.vision_forward(cfg, w, px, QR.QuantSpec((), base="f32"))
    assert np.abs(f32 - ref).max() < 1e-6                                    # no rounding anywhere = the float32 oracle
    b16 = QR.vision_forward(cfg, w, px, QR.QuantSpec(()))
    d16 = 1 - (b16 * ref).sum(1).min()
    assert 0 < d16 < 1e-4
    # e4m3 sites restricted to no block at all = the bf16 emulation, bit for bit; every added block moves it further from float32
    none = QR.vision_forward(cfg, w, px, QR.QuantSpec(QR.SITES, fp8_layers=()))
    assert np.array_equal(none, b16)
    last = QR.vision_forward(cfg, w, px, QR.QuantSpec(QR.SITES, fp8_layers=(cfg.layers - 1,)))
    every = QR.vision_forward(cfg, w, px, QR.QuantSpec(QR.SITES))
    d_last, d_all = 1 - (last * ref).sum(1).min(), 1 - (every * ref).sum(1).min()
    assert d16 < d_last < d_all
    # the bf16 side path for token 0 of the MLP sites only ever helps
    spec = QR.QuantSpec(("fc1", "fc2"), keep_rows=(0,))
    spec.keep_sites = {"fc1", "fc2"}
    side = QR.vision_forward(cfg, w, px, spec)
    plain = QR.vision_forward(cfg, w, px, QR.QuantSpec(("fc1", "fc2")))
    assert 1 - (side * ref).sum(1).min() < 1 - (plain * ref).sum(1).min()
    for mode in ("none", "row", "row_pow2", "block32"):
        aq, s = QR.quant_act_e4m3(torch.from_numpy(px.reshape(3, -1)[:, :256].copy()), mode)
        assert torch.isfinite(aq).all()


def test_outlier_stress_saturates_the_unit_scale_cast_and_row_scales_recover_it():
    """oracle/quant_ref.add_outliers drives hidden activations beyond 448: the unit-scale e4m3 cast of the emulation saturates there
    (what the HIP fc1 epilogue does, tests/test_fp8_gpu.py), per-row scales do not; the float32 function itself stays finite."""
    from ivr_amd import config as C
    from ivr_amd.weights import make_weights
    from oracle import quant_ref as QR
    from oracle import vit_ref as V
    cfg = C.TINY_VIT
    w0 = make_weights(cfg, 11)
    w = QR.add_outliers(cfg, w0, ln_gain=64.0, fc1_gain=50000.0, channels=2)
    assert not np.array_equal(w["l0.fc1_w"], w0["l0.fc1_w"]) and np.array_equal(w["l0.q_w"], w0["l0.q_w"])
    px = np.random.default_rng(0).standard_normal((3, 3, cfg.image, cfg.image)).astype(np.float32)
    ref = V.vision_forward(cfg, w, px)
    assert np.isfinite(ref).all()
    unit = QR.vision_forward(cfg, w, px, QR.QuantSpec(("fc1", "fc2")))
    row = QR.vision_forward(cfg, w, px, QR.QuantSpec(("fc1", "fc2"), act_scale="row"))
    e_unit, e_row = 1 - (unit * ref).sum(1).min(), 1 - (row * ref).sum(1).min()
    assert np.isfinite(unit).all() and e_row < e_unit
    x = torch.tensor([1000.0, -3000.0, 448.0, 500.0])
    assert QR.e4m3_round(x).tolist() == [448.0, -448.0, 448.0, 448.0]
