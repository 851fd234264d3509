"""ORACLE (test infrastructure, never shipped, never imported by the product path).

Operand-rounding emulation of the HIP towers' reduced-precision modes on the CPU: the arithmetic of
oracle/vit_ref.py (which restates what the reference obtains from HuggingFace at core.py:1619-1620,
core.py:1541-1542, video_frame_filter.py:31-32) with the GEMM operands of the four linear sites of a block
rounded the way the kernels round them, everything else float32:

  * bf16 site: A and W rounded to bfloat16 (RNE), product accumulated in float32;
  * fp8 site:  W -> OCP e4m3 with one float32 scale per output channel (absmax / 448), A -> e4m3 with the
    activation scaling under test ("none": unit scale as in round 1; "row": one float32 scale per token row,
    absmax / 448; "row_pow2": the same rounded up to a power of two; "block32": one E8M0 scale per 32
    consecutive K elements of a row, the MX layout v_mfma_scale_f32_16x16x128_f8f6f4 takes).

Two uses: (1) tools/fp8_error_budget.py - which sites / layers can run in e4m3 inside the north-star
tolerance (|delta score| <= 1e-3), decided before any kernel is written; (2) tests/test_fp8_gpu.py - the
HIP fp8 tower must agree with THIS emulation far more tightly than with the float32 oracle, which pins the
kernels' quantisation points rather than only their overall accuracy.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import vit_ref as V

SITES = ("qkv", "o", "fc1", "fc2")
E4M3_MAX = 448.0


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def e4m3_round(x):
    """float32 -> nearest OCP e4m3 value (RNE, saturating at +-448), returned as float32."""
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def quant_weight_e4m3(w):
    """[N,K] -> (e4m3 values as f32, per-row scale): tower.hip upload_fp8."""
    amax = w.abs().amax(dim=1, keepdim=True)
    s = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    return e4m3_round(w / s), s


def quant_act_e4m3(a, mode):
    """[..., K] activations -> (e4m3 values as f32, scale broadcastable to a)."""
    if mode == "none":
        return e4m3_round(a), torch.ones((), dtype=a.dtype)
    if mode in ("row", "row_pow2"):
        amax = a.abs().amax(dim=-1, keepdim=True)
        s = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
        if mode == "row_pow2":
            s = torch.exp2(torch.ceil(torch.log2(s)))
        return e4m3_round(a / s), s
    if mode == "block32":
        K = a.shape[-1]
        b = a.reshape(*a.shape[:-1], K // 32, 32)
        amax = b.abs().amax(dim=-1, keepdim=True)
        s = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
        s = torch.exp2(torch.ceil(torch.log2(s)))                       # E8M0: powers of two only
        q = e4m3_round(b / s)
        return (q * s).reshape(a.shape), torch.ones((), dtype=a.dtype)   # scale already folded back
    raise ValueError(mode)


def add_outliers(cfg, w, ln_gain=64.0, fc1_gain=2000.0, channels=4, seed=1, compensate=True):
    """Outlier stress for the e4m3 sites (trained CLIP towers have outlier channels; the random-init weights the presets were
    budgeted on do not): in every block, `channels` LayerNorm-2 gains are multiplied by ln_gain (LN outputs of a few hundred: inside
    e4m3's range, where its RELATIVE precision is what it is everywhere) and `channels` rows of fc1 (+ bias) by fc1_gain, which
    drives those hidden activations far beyond 448 = the saturation point of the unit-scale e4m3 cast.  compensate=True divides
    the matching fc2 columns by fc1_gain, so the float32 function stays well-conditioned and what is measured is the cast."""
    w2 = {k: np.array(v, copy=True) for k, v in w.items()}
    rng = np.random.default_rng(seed)
    for i in range(cfg.layers):
        ch = rng.choice(cfg.width, channels, replace=False)
        w2[f"l{i}.ln2_g"][ch] *= ln_gain
        rows = rng.choice(cfg.mlp, channels, replace=False)
        w2[f"l{i}.fc1_w"][rows] *= fc1_gain
        w2[f"l{i}.fc1_b"][rows] *= fc1_gain
        if compensate:
            w2[f"l{i}.fc2_w"][:, rows] /= fc1_gain
    return w2


class QuantSpec:
    """Which (layer, site) pairs run in e4m3; every other site runs in bf16 (f32 when `base` == "f32")."""

    def __init__(self, fp8_sites=(), fp8_layers=None, act_scale="none", base="bf16", keep_rows=None):
        self.keep_rows = keep_rows        # token positions whose rows of e4m3 sites are recomputed in bf16 (e.g. (0,) = CLS)
        self.keep_sites = None            # ... restricted to these sites (None = every e4m3 site); the kernels: {"fc1", "fc2"}
        self.fp8_sites = frozenset(fp8_sites)
        self.fp8_layers = None if fp8_layers is None else frozenset(fp8_layers)
        self.act_scale = act_scale
        self.base = base

    def is_fp8(self, layer, site):
        return site in self.fp8_sites and (self.fp8_layers is None or layer in self.fp8_layers)


def _linear(spec, layer, site, a, w, b, wcache):
    key = (layer, site)
    if spec.is_fp8(layer, site):
        if key not in wcache:
            wcache[key] = quant_weight_e4m3(w)
        wq, ws = wcache[key]
        aq, asc = quant_act_e4m3(a, spec.act_scale)
        y = F.linear(aq, wq) * ws.view(-1) * asc
        if spec.keep_rows is not None and (spec.keep_sites is None or site in spec.keep_sites):
            idx = list(spec.keep_rows)
            if ("b", key) not in wcache:
                wcache[("b", key)] = bf16_round(w)
            y[:, idx, :] = F.linear(bf16_round(a[:, idx, :]), wcache[("b", key)])
    elif spec.base == "bf16":
        if key not in wcache:
            wcache[key] = bf16_round(w)
        y = F.linear(bf16_round(a), wcache[key])
    else:
        y = F.linear(a, w)
    return y + b


def _block(cfg, w, i, x, mask, spec, wcache):
    p = f"l{i}."
    t = V._t
    B, T, D = x.shape
    H, dh = cfg.heads, cfg.width // cfg.heads
    h = F.layer_norm(x, (D,), t(w[p + "ln1_g"]), t(w[p + "ln1_b"]), cfg.ln_eps)
    scale = dh ** -0.5                                                     # folded into the Q rows at upload (tower.hip:359-369)
    wqkv = torch.cat([t(w[p + "q_w"]) * scale, t(w[p + "k_w"]), t(w[p + "v_w"])], 0)
    bqkv = torch.cat([t(w[p + "q_b"]) * scale, t(w[p + "k_b"]), t(w[p + "v_b"])], 0)
    qkv = _linear(spec, i, "qkv", h, wqkv, bqkv, wcache)
    if spec.base == "bf16":
        qkv = bf16_round(qkv)                                               # the QKV buffer is bf16 in both modes
    q, k, v = (z.view(B, T, H, dh).transpose(1, 2) for z in qkv.split(D, dim=-1))
    s = torch.matmul(q, k.transpose(-1, -2))
    if mask is not None:
        s = s + mask
    a = torch.softmax(s, dim=-1)
    if spec.base == "bf16":
        a = bf16_round(a * 1.0)      # P is packed to bf16 before the PV product (unnormalised in the kernel: same relative rounding)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, D)
    x = x + _linear(spec, i, "o", o, t(w[p + "o_w"]), t(w[p + "o_b"]), wcache)
    h = F.layer_norm(x, (D,), t(w[p + "ln2_g"]), t(w[p + "ln2_b"]), cfg.ln_eps)
    h = V._act(_linear(spec, i, "fc1", h, t(w[p + "fc1_w"]), t(w[p + "fc1_b"]), wcache), cfg.act)
    if spec.base == "bf16" and spec.is_fp8(i, "fc2") and not spec.is_fp8(i, "fc1"):
        h = bf16_round(h)            # a bf16 fc1 kernel feeding an e4m3 fc2: the hidden is rounded twice (tower.hip run_layers)
    x = x + _linear(spec, i, "fc2", h, t(w[p + "fc2_w"]), t(w[p + "fc2_b"]), wcache)
    return x


@torch.no_grad()
def vision_forward(cfg, w, pixels, spec, normalize=True):
    t = V._t
    x = t(pixels)
    B, D = x.shape[0], cfg.width
    pb = t(w["patch_b"]) if cfg.patch_bias else None
    if spec.base == "bf16":
        x = F.conv2d(bf16_round(x), bf16_round(t(w["patch_w"])), pb, stride=cfg.patch)
    else:
        x = F.conv2d(x, t(w["patch_w"]), pb, stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([t(w["cls"]).view(1, 1, D).expand(B, 1, D), x], 1) + t(w["pos"]).unsqueeze(0)
    if cfg.pre_ln:
        x = F.layer_norm(x, (D,), t(w["pre_ln_g"]), t(w["pre_ln_b"]), cfg.ln_eps)
    wcache = {}
    for i in range(cfg.layers):
        x = _block(cfg, w, i, x, None, spec, wcache)
    if cfg.pool == 0:
        c = F.layer_norm(x[:, 0, :], (D,), t(w["post_ln_g"]), t(w["post_ln_b"]), cfg.ln_eps)
        if spec.base == "bf16":
            out = F.linear(bf16_round(c), bf16_round(t(w["proj_w"])))
        else:
            out = F.linear(c, t(w["proj_w"]))
    else:
        out = F.layer_norm(x, (D,), t(w["post_ln_g"]), t(w["post_ln_b"]), cfg.ln_eps)[:, 0, :]
    if normalize:
        out = F.normalize(out, p=2, dim=1)
    return out.numpy().copy()


@torch.no_grad()
def text_forward(cfg, w, ids, spec, normalize=True):
    t = V._t
    ids_t = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))
    Q, T = ids_t.shape
    D = cfg.width
    x = t(w["tok"])[ids_t] + t(w["pos"])[:T].unsqueeze(0)
    mask = torch.full((T, T), float("-inf")).triu(1)
    wcache = {}
    for i in range(cfg.layers):
        x = _block(cfg, w, i, x, mask, spec, wcache)
    x = F.layer_norm(x, (D,), t(w["post_ln_g"]), t(w["post_ln_b"]), cfg.ln_eps)
    pos = (ids_t == cfg.eos_id).int().argmax(dim=-1)
    c = x[torch.arange(Q), pos]
    out = F.linear(bf16_round(c), bf16_round(t(w["proj_w"]))) if spec.base == "bf16" else F.linear(c, t(w["proj_w"]))
    if normalize:
        out = F.normalize(out, p=2, dim=1)
    return out.numpy().copy()
