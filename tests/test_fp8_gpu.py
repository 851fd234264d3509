"""GPU parity of the fp8 mode (BASELINE.json configs[4]: "ViT-L/14 fp8 (CDNA4 fp8 MFMA)").

Two layers of checks, tolerances written here:
  * the fp8 GEMM kernel against a float32 product of the SAME e4m3 operands (torch's float8_e4m3fn -> float32 is exact),
    so only accumulation order and the output rounding differ: bf16 output 2^-8 relative, e4m3 output one e4m3 step
    (2^-3 relative, compared after dequantisation), float32 residual 1e-4 relative to the row scale;
  * the whole tower against the float32 HF golden vectors, per assignment of the four linear sites to e4m3
    (oracle/quant_ref.py + tools/fp8_error_budget.py give the CPU-emulated budget, profiles/r02_fp8_error_budget.json):
      compute="fp8"      fc1 + fc2 in e4m3 in the last third of the blocks, token-0 rows bf16: the preset that meets the north-star
                         bound (|score - f32 score| <= 1e-3 on every pair), 1 - cos <= 1e-4 asserted
      compute="fp8_mlp"  the same in every block: 1 - cos <= 1e-3 asserted (emulated 4-5e-4)
      compute="fp8_all"  all four sites: e4m3 carries 3 mantissa bits, 1 - cos ~ 3-4e-3; asserted cosine >= 0.99
    and against the operand-rounding EMULATION of the same assignment, which pins where the kernels quantise;
  * configs[4] as one workload: ViT-L/14 e4m3 rows in a 768-d index, mixed text + image queries, ids exact against the
    oracle search over the same rows, |score - f32-oracle score| reported against the 1e-3 north-star bound.
"""
import numpy as np
import pytest
import torch

from ivr_amd import config as C

pytestmark = pytest.mark.gpu


def _operands(M, N, K, seed):
    from ivr_amd.linear import quantize_rows_e4m3
    g = torch.Generator(device="cuda").manual_seed(seed)
    x8 = (torch.randn((M, K), generator=g, device="cuda") * 0.9).to(torch.float8_e4m3fn)
    w8, ws = quantize_rows_e4m3(torch.randn((N, K), generator=g, device="cuda") * K ** -0.5)
    b = torch.randn(N, generator=g, device="cuda") * 0.1
    return x8, w8, ws, b


def _ref(x8, w8, ws, b, act):
    y = (x8.float().cpu() @ w8.view(torch.float8_e4m3fn).float().cpu().T) * ws.cpu()[None, :] + (b.cpu() if b is not None else 0)
    if act == 0:
        y = y * torch.sigmoid(1.702 * y)
    elif act == 1:
        y = torch.nn.functional.gelu(y)
    return y


@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (50, 128, 256), (129, 320, 384), (400, 768, 768), (257, 2304, 768), (1000, 768, 3072),
                                   (515, 1024, 4096)])
def test_fp8_gemm_bf16_output(M, N, K):
    from ivr_amd.linear import linear_fp8
    x8, w8, ws, b = _operands(M, N, K, M * 31 + N)
    for act in (-1, 0, 1):
        y = linear_fp8(x8, w8, ws, b, act=act).float().cpu()
        ref = _ref(x8, w8, ws, b, act)
        assert (y - ref).abs().max() <= 1.2e-2 * max(1.0, ref.abs().max()), (act, (y - ref).abs().max())
    y = linear_fp8(x8, w8, None, None).float().cpu()            # no scale, no bias
    ref = x8.float().cpu() @ w8.view(torch.float8_e4m3fn).float().cpu().T
    assert (y - ref).abs().max() <= 1.2e-2 * max(1.0, ref.abs().max())


def test_fp8_gemm_exact_small_integers():
    """Integers up to 3 are exact in e4m3 and the sums are exact in float32: any fragment / K-permutation slip is != 0."""
    from ivr_amd.linear import EPI_RESID, linear_fp8
    rng = np.random.default_rng(0)
    M, N, K = 300, 192, 384
    x = torch.from_numpy(rng.integers(-3, 4, (M, K)).astype(np.float32))
    w = torch.from_numpy(rng.integers(-3, 4, (N, K)).astype(np.float32))
    r = torch.zeros((M, N), device="cuda")
    linear_fp8(x.cuda().to(torch.float8_e4m3fn), w.cuda().to(torch.float8_e4m3fn), None, None, epilogue=EPI_RESID, resid=r)
    assert torch.equal(r.cpu(), x @ w.T)


def test_fp8_gemm_residual_and_e4m3_output():
    from ivr_amd.linear import EPI_RESID, linear_fp8
    M, N, K = 333, 512, 1024
    x8, w8, ws, b = _operands(M, N, K, 7)
    r0 = torch.randn((M, N), device="cuda")
    r = r0.clone()
    linear_fp8(x8, w8, ws, b, epilogue=EPI_RESID, resid=r)
    ref = _ref(x8, w8, ws, b, -1)
    assert (r.cpu() - (r0.cpu() + ref)).abs().max() < 1e-4 * max(1.0, ref.abs().max())
    for act in (-1, 0):
        y8 = linear_fp8(x8, w8, ws, b, act=act, out_fp8=True)
        ref = _ref(x8, w8, ws, b, act)
        y = y8.float().cpu()
        # one e4m3 step: 2^-3 of the value for normals, 2^-9 absolute in the subnormal range
        assert ((y - ref).abs() <= 0.0626 * ref.abs() + 2.0 ** -9).all(), (act, (y - ref).abs().max())
    big = linear_fp8(x8, w8, ws * 1e4, b, out_fp8=True).float()      # saturation, not NaN / inf
    assert torch.isfinite(big).all() and big.abs().max() == 448.0


def test_fp8_gemm_rejects_bad_shapes():
    from ivr_amd.linear import linear_fp8
    x8 = torch.zeros((4, 192), dtype=torch.uint8, device="cuda")
    w8 = torch.zeros((64, 192), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        linear_fp8(x8, w8)                                        # K % 128 != 0
    x8 = torch.zeros((4, 128), dtype=torch.uint8, device="cuda")
    w8 = torch.zeros((72, 128), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        linear_fp8(x8, w8)                                        # N % 64 != 0


CASES = [(C.TINY_VIT, 4), (C.CLIP_VIT_B32, 8), (C.DINO_VIT_S16, 2), (C.CLIP_VIT_L14, 2)]


@pytest.mark.parametrize("cfg,n", CASES, ids=lambda v: getattr(v, "name", str(v)))
def test_vision_tower_fp8_all_sites(cfg, n, golden):
    from test_tower_gpu import _cos, _vision
    g = golden("towers")
    _, _, _, out = _vision(cfg, "fp8_all", n)
    ref = g[cfg.name + "_emb"][:n]
    cos = _cos(out, ref)
    print(f"{cfg.name} fp8_all (qkv, attn-out, fc1, fc2 in e4m3) 1 - min cos to fp32 HF = {1 - cos.min():.2e}")
    assert cos.min() > 0.99
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5


@pytest.mark.parametrize("cfg,n", CASES, ids=lambda v: getattr(v, "name", str(v)))
def test_vision_tower_fp8_mlp_assignment_within_1e3(cfg, n, golden):
    """compute="fp8_mlp": MLP sites in e4m3 in every block, token-0 rows of those sites in bf16 -> 1 - cos inside 1e-3."""
    from test_tower_gpu import _cos, _vision
    g = golden("towers")
    tw, _, _, out = _vision(cfg, "fp8_mlp", n)
    assert tw.fp8_sites == 12 and tw.fp8_cls_bf16 == 1 and tw.fp8_first_layer == 0
    ref = g[cfg.name + "_emb"][:n]
    cos = _cos(out, ref)
    print(f"{cfg.name} fp8_mlp (fc1+fc2 e4m3, token-0 rows bf16) 1 - min cos to fp32 HF = {1 - cos.min():.2e}")
    assert 1 - cos.min() <= 1e-3
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5


@pytest.mark.parametrize("cfg,n", CASES[1:], ids=lambda v: getattr(v, "name", str(v)))
def test_vision_tower_fp8_default_preset_is_the_one_inside_the_bound(cfg, n, golden):
    """compute="fp8" (old name "fp8_strict"): MLP sites in e4m3 only in the last third of the blocks (+ bf16 token-0 rows): within
    a few times the bf16 error, and equal to the CPU emulation's budget for that assignment."""
    from test_tower_gpu import _cos, _vision
    g = golden("towers")
    tw, _, _, out = _vision(cfg, "fp8", n)
    assert tw.fp8_first_layer == (2 * cfg.layers) // 3 and tw.fp8_sites == 12 and tw.fp8_cls_bf16 == 1
    old, _, _, out_old = _vision(cfg, "fp8_strict", n)
    assert (old.fp8_first_layer, old.fp8_sites, old.fp8_cls_bf16) == (tw.fp8_first_layer, tw.fp8_sites, tw.fp8_cls_bf16)
    assert np.array_equal(out, out_old)
    ref = g[cfg.name + "_emb"][:n]
    cos = _cos(out, ref)
    print(f"{cfg.name} fp8 (fc1+fc2 e4m3 in blocks >= {tw.fp8_first_layer}, token-0 rows bf16) 1 - min cos to fp32 HF = {1 - cos.min():.2e}")
    assert 1 - cos.min() <= 1e-4


def test_fp8_presets_under_outlier_channels():
    """Outlier stress: LayerNorm-2 gains x64 on four channels and four fc1 rows x2000 per block (fc2 columns divided
    back) put LN outputs at a few hundred and hidden activations at thousands - far beyond 448, where the unit-scale e4m3 cast
    saturates.  The HIP towers must saturate exactly where the emulation does (oracle/quant_ref.py: clamp, then RNE), and the
    measured cost is asserted: the preset inside the north-star bound is unaffected (its e4m3 blocks are the last third and the
    pooled token-0 rows stay bf16), fp8_mlp roughly doubles its error and stays inside 2e-3.  Emulated per-row activation scales
    halve fp8_mlp's error here (printed by tools/fp8_error_budget.py --outliers) - not needed for the shipped bound."""
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    from conftest import synth_frames
    from oracle import preprocess_ref as P
    from oracle import quant_ref as QR
    from oracle import vit_ref as V
    cfg = C.CLIP_VIT_B32
    w = QR.add_outliers(cfg, make_weights(cfg, 12))
    frames = synth_frames(99, 6, 224, 224)
    px = P.preprocess(frames, "identity", C.CLIP_MEAN, C.CLIP_STD)
    ref = V.vision_forward(cfg, w, px)
    L = cfg.layers
    specs = {"fp8": QR.QuantSpec(("fc1", "fc2"), fp8_layers=range(2 * L // 3, L), keep_rows=(0,)),
             "fp8_mlp": QR.QuantSpec(("fc1", "fc2"), keep_rows=(0,))}
    limit = {"fp8": 1e-4, "fp8_mlp": 2e-3}
    for compute, spec in specs.items():
        spec.keep_sites = {"fc1", "fc2"}
        emu = QR.vision_forward(cfg, w, px, spec)
        out = Tower(cfg, w, max_batch=6, compute=compute).encode_frames(frames).cpu().numpy()
        assert np.isfinite(out).all()
        d_ref, d_emu, e_ref = 1 - (out * ref).sum(1).min(), 1 - (out * emu).sum(1).min(), 1 - (emu * ref).sum(1).min()
        print(f"outliers {compute}: 1-cos gpu/f32 {d_ref:.2e}, emulation/f32 {e_ref:.2e}, gpu/emulation {d_emu:.2e}")
        assert d_ref <= limit[compute]
        assert 0.5 * e_ref - 2e-5 <= d_ref <= 1.6 * e_ref + 2e-5


SITE_SETS = [("qkv",), ("o",), ("fc1",), ("fc2",), ("fc1", "fc2"), ("qkv", "o", "fc1", "fc2")]


@pytest.mark.parametrize("sites", SITE_SETS, ids=lambda s: "+".join(s))
@pytest.mark.parametrize("cls_bf16", [False, True])
def test_fp8_site_assignments_match_emulation(sites, cls_bf16):
    """Every single-site assignment (and the two shipped ones) on ViT-B/32: the HIP tower must sit much closer to the CPU
    operand-rounding emulation of the SAME assignment than to the float32 oracle (the e4m3 error is deterministic: the same
    operands are rounded at the same points), and its distance to the float32 oracle must match the emulated budget."""
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    from conftest import synth_frames
    from oracle import preprocess_ref as P
    from oracle import quant_ref as QR
    from oracle import vit_ref as V
    if cls_bf16 and not ({"fc1", "fc2"} & set(sites)):
        pytest.skip("the bf16 side path only exists for the MLP sites")
    cfg = C.CLIP_VIT_B32
    w = make_weights(cfg, 12)
    frames = synth_frames(99, 6, 224, 224)
    px = P.preprocess(frames, "identity", C.CLIP_MEAN, C.CLIP_STD)
    ref = V.vision_forward(cfg, w, px)
    # the emulation's side path covers every e4m3 site; the kernels' only the MLP sites - emulate exactly what runs
    spec = QR.QuantSpec(sites, keep_rows=(0,) if cls_bf16 else None)
    spec.keep_sites = {"fc1", "fc2"}
    emu = QR.vision_forward(cfg, w, px, spec)
    out = Tower(cfg, w, max_batch=6, compute="fp8_all", fp8_sites=sites, fp8_cls_bf16=cls_bf16).encode_frames(frames).cpu().numpy()
    d_ref, d_emu, e_ref = 1 - (out * ref).sum(1).min(), 1 - (out * emu).sum(1).min(), 1 - (emu * ref).sum(1).min()
    print(f"sites={'+'.join(sites)} cls_bf16={cls_bf16}: 1-cos gpu/f32 {d_ref:.2e}, emulation/f32 {e_ref:.2e}, gpu/emulation {d_emu:.2e}")
    # Rounding to 3 mantissa bits amplifies last-bit differences (accumulation order, fast exp): an element whose pre-rounding
    # value moves across an e4m3 boundary changes by a whole step, so two exact-arithmetic-equivalent runs decorrelate layer by
    # layer and only agree to a fraction of the quantisation noise (measured 0.03 - 0.5 of it, least where attention averages
    # the perturbation).  What must hold: the HIP tower is closer to the emulation than the emulation is to float32, and its
    # distance to float32 IS the emulated budget (measured within 8 %).
    assert d_emu <= 0.75 * e_ref + 2e-5
    assert 0.7 * e_ref - 2e-5 <= d_ref <= 1.4 * e_ref + 2e-5


def test_config4_workload_fp8_rows_mixed_queries():
    """BASELINE configs[4] at test scale as ONE workload: 64 ViT-L/14 rows from the e4m3 tower in a 768-d index, a mixed batch of 16
    image queries (same e4m3 tower) and 16 text queries, exact top-k through the HIP search.
      ids: bit-exact against the oracle search over the SAME rows and queries;
      scores: all 64 x 32 pairs against the float32 oracle towers' scores for the same frames / token ids (text queries from the
      f32 text tower: a handful of queries costs nothing, and the bf16 text tower alone moves these scores by up to 1.1e-3).
    The north-star bound is 1e-3 on every pair: compute="fp8" IS the assignment that meets it (asserted on the max over the
    2,048 pairs and on every top-k pair; max and p99 printed per preset).  "fp8_mlp" (e4m3 MLP in every block) keeps the embedding
    bound 1 - cos <= 1e-3 and is asserted at 3e-3 on the scores; "fp8_all" at 1e-2.  Random-init weights (no trained checkpoint is
    reachable offline): profiles/r02_fp8_error_budget.json has the emulated table the presets come from."""
    from ivr_amd.index import FlatIPIndex
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    from conftest import synth_frames
    from oracle import preprocess_ref as P
    from oracle import search_ref as S
    from oracle import vit_ref as V
    vis, txt = C.CLIP_VIT_L14, C.CLIP_TEXT_L14
    wv, wt = make_weights(vis, 12), make_weights(txt, 13)
    n_rows, n_iq, n_tq, k = 64, 16, 16, 10
    frames = synth_frames(1234, n_rows + n_iq, 224, 224)
    rng = np.random.default_rng(77)
    ids = rng.integers(1, txt.vocab - 2, (n_tq, 16)).astype(np.int64)
    for r in range(n_tq):
        ids[r, rng.integers(4, 16):] = txt.eos_id
    results = {}
    tq = Tower(txt, wt, max_batch=n_tq, compute="f32").encode_ids(ids)     # a handful of queries: the f32 text tower costs nothing
    for compute in ("fp8", "fp8_mlp", "fp8_all", "bf16"):
        tw = Tower(vis, wv, max_batch=n_rows + n_iq, compute=compute)
        if compute == "fp8":
            assert tw.fp8_first_layer == (2 * vis.layers) // 3           # the default e4m3 preset is the one inside the bound
        emb = tw.encode_frames(frames)
        tw.close()
        idx = FlatIPIndex(768)
        idx.add(emb[:n_rows])
        queries = torch.cat([emb[n_rows:], tq])
        D, I = idx.search_device(queries, k)
        rows_h, q_h = emb[:n_rows].cpu().numpy(), queries.cpu().numpy()
        Dr, Ir = S.flat_ip_search(rows_h, q_h, k + 1, dtype=np.float64)
        # ids exact over the same rows - outside float32 near-ties: random-init towers put all 64 rows within ~1e-2 of each other, so
        # neighbours in the ranking can be closer than float32 scoring resolves (the exact float32 path orders them the same way)
        gap = np.minimum(np.abs(np.diff(Dr, axis=1, prepend=np.inf))[:, :k], np.abs(np.diff(Dr, axis=1))[:, :k])
        firm = gap > 1e-6
        assert np.array_equal(I.cpu().numpy()[firm], Ir[:, :k][firm]), compute
        assert firm.mean() > 0.9
        assert np.abs(D.cpu().numpy() - Dr[:, :k]).max() < 1e-5
        results[compute] = (rows_h, q_h, I.cpu().numpy())
    ref = np.concatenate([V.vision_forward(vis, wv, P.preprocess(frames[i:i + 16], "identity", C.CLIP_MEAN, C.CLIP_STD))
                          for i in range(0, len(frames), 16)])
    tref = V.text_forward(txt, wt, ids)
    Sref = np.concatenate([ref[n_rows:], tref]) @ ref[:n_rows].T
    worst = {}
    for compute, (rows_h, q_h, I) in results.items():
        d = np.abs(q_h @ rows_h.T - Sref)
        topk = np.take_along_axis(d, I, axis=1)                            # the pairs a caller actually sees
        worst[compute] = (d[:n_iq].max(), d[n_iq:].max(), 1 - (rows_h * ref[:n_rows]).sum(1).min(), topk.max())
        print(f"configs[4] {compute:8s} |score - f32 oracle score| over {d.size} pairs: image queries max {d[:n_iq].max():.2e} p99 "
              f"{np.quantile(d[:n_iq], 0.99):.2e}, text queries max {d[n_iq:].max():.2e} p99 {np.quantile(d[n_iq:], 0.99):.2e}, top-{k} pairs max "
              f"{topk.max():.2e}; 1 - min cos of the rows {worst[compute][2]:.2e}")
    assert worst["bf16"][1] <= 1e-3 and worst["bf16"][0] <= 1e-3
    assert max(worst["fp8"][0], worst["fp8"][1], worst["fp8"][3]) <= 1e-3 and worst["fp8"][2] <= 1e-4       # the north-star bound, every pair
    assert worst["fp8_mlp"][2] <= 1e-3 and worst["fp8_mlp"][1] <= 3e-3 and worst["fp8_mlp"][0] <= 3e-3
    assert worst["fp8_all"][1] <= 1e-2


def test_text_tower_fp8_mode(golden):
    from ivr_amd.tower import Tower
    from ivr_amd.weights import make_weights
    from oracle import vit_ref as V
    cfg = C.TINY_TEXT
    w = make_weights(cfg, 3)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, cfg.vocab - 1, (6, cfg.tokens))
    ids[:, -3] = cfg.eos_id
    ref = np.asarray(V.text_forward(cfg, w, ids))
    for compute in ("fp8_mlp", "fp8_all"):    # text towers pool at the EOS token: no token-0 side path, plain site masks
        out = Tower(cfg, w, max_batch=8, compute=compute).encode_ids(ids).cpu().numpy()
        cos = (out * ref).sum(1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
        print(f"tiny-text {compute} min cos={cos.min():.5f}")
        assert cos.min() > 0.99
