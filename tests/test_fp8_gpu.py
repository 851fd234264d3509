"""GPU parity of the fp8 mode (BASELINE.json configs[4]: "ViT-L/14 fp8 (CDNA4 fp8 MFMA)").

Two layers of checks, tolerances written here:
  * the fp8 GEMM kernel against a float32 product of the SAME e4m3 operands (torch's float8_e4m3fn -> float32 is exact),
    so only accumulation order and the output rounding differ: bf16 output 2^-8 relative, e4m3 output one e4m3 step
    (2^-3 relative, compared after dequantisation), float32 residual 1e-4 relative to the row scale;
  * the whole tower in fp8 mode against the float32 HF golden vectors.  e4m3 carries 3 mantissa bits, so this mode does
    NOT meet the 1e-3 score bound of the bf16 mode: the bound asserted is cosine >= 0.99 and the measured value is
    printed.  north_star states no tolerance for configs[4]; this one is ours.
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


@pytest.mark.parametrize("cfg,n", [(C.TINY_VIT, 4), (C.CLIP_VIT_B32, 8), (C.DINO_VIT_S16, 2), (C.CLIP_VIT_L14, 2)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_vision_tower_fp8_mode(cfg, n, golden):
    from test_tower_gpu import _cos, _vision
    g = golden("towers")
    _, _, _, out = _vision(cfg, "fp8", n)
    ref = g[cfg.name + "_emb"][:n]
    cos = _cos(out, ref)
    _, _, _, out_bf16 = _vision(cfg, "bf16", n)
    print(f"{cfg.name} fp8 min cos to fp32 HF={cos.min():.5f} (bf16 mode: {_cos(out_bf16, ref).min():.6f})")
    assert cos.min() > 0.99
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5


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
    out = Tower(cfg, w, max_batch=8, compute="fp8").encode_ids(ids).cpu().numpy()
    cos = (out * ref).sum(1) / (np.linalg.norm(out, axis=1) * np.linalg.norm(ref, axis=1))
    print(f"tiny-text fp8 min cos={cos.min():.5f}")
    assert cos.min() > 0.99
