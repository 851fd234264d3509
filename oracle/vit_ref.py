"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU fp32 restatement of the encoder arithmetic the reference delegates to
HuggingFace at `core.py:1619-1620` (CLIPModel.get_image_features + F.normalize),
`core.py:1541-1542` (get_text_features + F.normalize) and
`video_frame_filter.py:28-33` (ViTModel(...).last_hidden_state[:, 0, :]).

The algorithm lives in the un-vendored, un-pinned dependency `transformers`
(installed here: 5.15.0); the restated arithmetic follows
transformers/models/clip/modeling_clip.py:138-218 (embeddings), :259-277
(attention, scale = head_dim**-0.5, softmax in fp32), :338-350 (MLP, quick_gelu
= x*sigmoid(1.702x)), :353-383 (pre-LN block), :594-656 (pre_layrnorm, CLS,
post_layernorm), :719-753 (visual_projection), :541-581 (text tower, first-EOS
pooling) and transformers/models/vit/modeling_vit.py:261-281,348,385.

Pinned: tests/golden/make_golden.py checks this file against the HF modules
themselves (same seeded weights) in the build container and commits the
expected embeddings under tests/golden/.  The reference's own tests hold no
vector for this path, so the HF library is the only anchor.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


def _act(x, act):
    if act == 0:
        return x * torch.sigmoid(1.702 * x)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _block(cfg, w, i, x, mask):
    """One pre-LN transformer block on x[B,T,D]."""
    p = f"l{i}."
    B, T, D = x.shape
    H, dh = cfg.heads, cfg.width // cfg.heads
    h = F.layer_norm(x, (D,), _t(w[p + "ln1_g"]), _t(w[p + "ln1_b"]), cfg.ln_eps)
    q = F.linear(h, _t(w[p + "q_w"]), _t(w[p + "q_b"])).view(B, T, H, dh).transpose(1, 2)
    k = F.linear(h, _t(w[p + "k_w"]), _t(w[p + "k_b"])).view(B, T, H, dh).transpose(1, 2)
    v = F.linear(h, _t(w[p + "v_w"]), _t(w[p + "v_b"])).view(B, T, H, dh).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    if mask is not None:
        s = s + mask
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(o, _t(w[p + "o_w"]), _t(w[p + "o_b"]))
    h = F.layer_norm(x, (D,), _t(w[p + "ln2_g"]), _t(w[p + "ln2_b"]), cfg.ln_eps)
    h = _act(F.linear(h, _t(w[p + "fc1_w"]), _t(w[p + "fc1_b"])), cfg.act)
    x = x + F.linear(h, _t(w[p + "fc2_w"]), _t(w[p + "fc2_b"]))
    return x


@torch.no_grad()
def vision_forward(cfg, w, pixels, return_hidden=False, normalize=True):
    """pixels fp32 [B,3,S,S] (already preprocessed) -> [B, embed_dim] fp32."""
    x = _t(pixels)
    B = x.shape[0]
    D = cfg.width
    pb = _t(w["patch_b"]) if cfg.patch_bias else None
    x = F.conv2d(x, _t(w["patch_w"]), pb, stride=cfg.patch)           # [B,D,g,g]
    x = x.flatten(2).transpose(1, 2)                                   # [B,g*g,D]
    x = torch.cat([_t(w["cls"]).view(1, 1, D).expand(B, 1, D), x], 1) + _t(w["pos"]).unsqueeze(0)
    if cfg.pre_ln:
        x = F.layer_norm(x, (D,), _t(w["pre_ln_g"]), _t(w["pre_ln_b"]), cfg.ln_eps)
    hidden = [x.numpy().copy()] if return_hidden else None
    for i in range(cfg.layers):
        x = _block(cfg, w, i, x, None)
        if return_hidden:
            hidden.append(x.numpy().copy())
    if cfg.pool == 0:      # CLIP: CLS -> post-LN -> projection
        c = F.layer_norm(x[:, 0, :], (D,), _t(w["post_ln_g"]), _t(w["post_ln_b"]), cfg.ln_eps)
        out = F.linear(c, _t(w["proj_w"]))
    else:                  # HF ViTModel: LN on all tokens, CLS row of last_hidden_state
        out = F.layer_norm(x, (D,), _t(w["post_ln_g"]), _t(w["post_ln_b"]), cfg.ln_eps)[:, 0, :]
    if normalize:          # core.py:1620  F.normalize(p=2, dim=1) == x / max(||x||, 1e-12)
        out = F.normalize(out, p=2, dim=1)
    out = out.numpy().copy()
    return (out, hidden) if return_hidden else out


@torch.no_grad()
def text_forward(cfg, w, ids, normalize=True):
    """ids int64 [Q,T] -> [Q,out_dim]; pooled at the first EOS position (modeling_clip.py:566-576)."""
    ids_t = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))
    Q, T = ids_t.shape
    D = cfg.width
    x = _t(w["tok"])[ids_t] + _t(w["pos"])[:T].unsqueeze(0)
    mask = torch.full((T, T), float("-inf")).triu(1)
    for i in range(cfg.layers):
        x = _block(cfg, w, i, x, mask)
    x = F.layer_norm(x, (D,), _t(w["post_ln_g"]), _t(w["post_ln_b"]), cfg.ln_eps)
    pos = (ids_t == cfg.eos_id).int().argmax(dim=-1)
    out = F.linear(x[torch.arange(Q), pos], _t(w["proj_w"]))
    if normalize:
        out = F.normalize(out, p=2, dim=1)
    return out.numpy().copy()
