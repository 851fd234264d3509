"""Tower weights: seeded synthetic generator and HF state-dict name mapping.

There is no network for checkpoints, so benchmarks and tests use random-init
weights of the named architecture drawn from a seeded NumPy PCG64 stream.  The
same fp32 master dict feeds the CPU oracle, the HF model used to pin the
oracle (tests/golden/make_golden.py) and the HIP encoder (cast to bf16 there).

Canonical (flat) names, all float32, nn.Linear convention W[out, in]:
  vision: patch_w[D,3,P,P] (patch_b[D]) cls[D] pos[T,D] (pre_ln_g/b[D])
  text:   tok[V,D] pos[T,D]
  per layer i: l{i}.ln1_g/b, l{i}.q_w/q_b, k_w/k_b, v_w/v_b, o_w/o_b,
               l{i}.ln2_g/b, l{i}.fc1_w[M,D]/fc1_b, l{i}.fc2_w[D,M]/fc2_b
  post_ln_g/b[D], proj_w[out,D] (when cfg.out_dim)
"""
import numpy as np

from .config import TowerConfig


def tensor_specs(cfg: TowerConfig):
    """Ordered (name, shape, kind) list; kind picks the init distribution."""
    D, M = cfg.width, cfg.mlp
    specs = []
    if cfg.kind == "vision":
        specs.append(("patch_w", (D, 3, cfg.patch, cfg.patch), "lin"))
        if cfg.patch_bias:
            specs.append(("patch_b", (D,), "bias"))
        specs.append(("cls", (D,), "emb"))
        specs.append(("pos", (cfg.tokens, D), "emb"))
        if cfg.pre_ln:
            specs += [("pre_ln_g", (D,), "gamma"), ("pre_ln_b", (D,), "bias")]
    else:
        specs.append(("tok", (cfg.vocab, D), "emb"))
        specs.append(("pos", (cfg.tokens, D), "emb"))
    for i in range(cfg.layers):
        p = f"l{i}."
        specs += [(p + "ln1_g", (D,), "gamma"), (p + "ln1_b", (D,), "bias")]
        for n in ("q", "k", "v", "o"):
            specs += [(p + n + "_w", (D, D), "lin"), (p + n + "_b", (D,), "bias")]
        specs += [(p + "ln2_g", (D,), "gamma"), (p + "ln2_b", (D,), "bias"),
                  (p + "fc1_w", (M, D), "lin"), (p + "fc1_b", (M,), "bias"),
                  (p + "fc2_w", (D, M), "lin"), (p + "fc2_b", (D,), "bias")]
    specs += [("post_ln_g", (D,), "gamma"), ("post_ln_b", (D,), "bias")]
    if cfg.out_dim:
        specs.append(("proj_w", (cfg.out_dim, D), "lin"))
    return specs


def make_weights(cfg: TowerConfig, seed: int = 0):
    """Random-init weights.  Linear weights are N(0, 1/fan_in) so activations
    keep O(1) variance through the stack and the softmax is not degenerate."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape, kind in tensor_specs(cfg):
        z = rng.standard_normal(shape, dtype=np.float32)
        if kind == "lin":
            fan_in = int(np.prod(shape[1:]))
            z *= np.float32(fan_in ** -0.5)
        elif kind == "bias":
            z *= np.float32(0.05)
        elif kind == "gamma":
            z = np.float32(1.0) + np.float32(0.1) * z
        elif kind == "emb":
            z *= np.float32(0.5)
        out[name] = np.ascontiguousarray(z, dtype=np.float32)
    return out


def to_hf_state_dict(cfg: TowerConfig, w, vit_style="v5"):
    """Canonical names -> HuggingFace state-dict names (CLIPVisionModelWithProjection /
    ViTModel / CLIPTextModelWithProjection).  Used by make_golden.py to load the same
    weights into the HF model, and in reverse by `from_hf_state_dict`.  `vit_style`:
    "v5" = transformers>=5 ViTModel module names, "v4" = the names stored in published
    checkpoints such as facebook/dino-vits16."""
    sd = {}
    if cfg.kind == "vision" and cfg.pool != 1:  # CLIP vision
        pre = "vision_model."
        sd[pre + "embeddings.patch_embedding.weight"] = w["patch_w"]
        sd[pre + "embeddings.class_embedding"] = w["cls"]
        sd[pre + "embeddings.position_embedding.weight"] = w["pos"]
        sd[pre + "pre_layrnorm.weight"] = w["pre_ln_g"]
        sd[pre + "pre_layrnorm.bias"] = w["pre_ln_b"]
        sd[pre + "post_layernorm.weight"] = w["post_ln_g"]
        sd[pre + "post_layernorm.bias"] = w["post_ln_b"]
        sd["visual_projection.weight"] = w["proj_w"]
        lay = pre + "encoder.layers.{}."
        names = _CLIP_LAYER
    elif cfg.kind == "vision":  # HF ViTModel (DINO)
        sd["embeddings.patch_embeddings.projection.weight"] = w["patch_w"]
        sd["embeddings.patch_embeddings.projection.bias"] = w["patch_b"]
        sd["embeddings.cls_token"] = w["cls"].reshape(1, 1, -1)
        sd["embeddings.position_embeddings"] = w["pos"].reshape(1, cfg.tokens, -1)
        sd["layernorm.weight"] = w["post_ln_g"]
        sd["layernorm.bias"] = w["post_ln_b"]
        lay = "layers.{}." if vit_style == "v5" else "encoder.layer.{}."
        names = _VIT_LAYER_V5 if vit_style == "v5" else _VIT_LAYER
    else:  # CLIP text
        pre = "text_model."
        sd[pre + "embeddings.token_embedding.weight"] = w["tok"]
        sd[pre + "embeddings.position_embedding.weight"] = w["pos"]
        sd[pre + "final_layer_norm.weight"] = w["post_ln_g"]
        sd[pre + "final_layer_norm.bias"] = w["post_ln_b"]
        sd["text_projection.weight"] = w["proj_w"]
        lay = pre + "encoder.layers.{}."
        names = _CLIP_LAYER
    for i in range(cfg.layers):
        for ours, theirs in names.items():
            sd[lay.format(i) + theirs] = w[f"l{i}.{ours}"]
    return sd


def from_hf_state_dict(cfg: TowerConfig, sd):
    """Inverse of `to_hf_state_dict` for real checkpoints (safetensors / weights_only loads)."""
    class _Tag(str):  # survives the reshape() calls in to_hf_state_dict
        def reshape(self, *a):
            return self
    tags = {n: _Tag(n) for n, _, _ in tensor_specs(cfg)}
    probe = to_hf_state_dict(cfg, tags)
    if cfg.kind == "vision" and cfg.pool == 1 and not any(k.startswith("layers.") for k in sd):
        probe = to_hf_state_dict(cfg, tags, vit_style="v4")
    out = {}
    for hf_name, ours in probe.items():
        if hf_name not in sd:
            raise KeyError(f"checkpoint is missing {hf_name}")
        shape = dict((n, s) for n, s, _ in tensor_specs(cfg))[ours]
        out[ours] = np.ascontiguousarray(np.asarray(sd[hf_name], dtype=np.float32).reshape(shape))
    return out


_CLIP_LAYER = {
    "ln1_g": "layer_norm1.weight", "ln1_b": "layer_norm1.bias",
    "q_w": "self_attn.q_proj.weight", "q_b": "self_attn.q_proj.bias",
    "k_w": "self_attn.k_proj.weight", "k_b": "self_attn.k_proj.bias",
    "v_w": "self_attn.v_proj.weight", "v_b": "self_attn.v_proj.bias",
    "o_w": "self_attn.out_proj.weight", "o_b": "self_attn.out_proj.bias",
    "ln2_g": "layer_norm2.weight", "ln2_b": "layer_norm2.bias",
    "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias",
    "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias",
}
_VIT_LAYER = {
    "ln1_g": "layernorm_before.weight", "ln1_b": "layernorm_before.bias",
    "q_w": "attention.attention.query.weight", "q_b": "attention.attention.query.bias",
    "k_w": "attention.attention.key.weight", "k_b": "attention.attention.key.bias",
    "v_w": "attention.attention.value.weight", "v_b": "attention.attention.value.bias",
    "o_w": "attention.output.dense.weight", "o_b": "attention.output.dense.bias",
    "ln2_g": "layernorm_after.weight", "ln2_b": "layernorm_after.bias",
    "fc1_w": "intermediate.dense.weight", "fc1_b": "intermediate.dense.bias",
    "fc2_w": "output.dense.weight", "fc2_b": "output.dense.bias",
}
_VIT_LAYER_V5 = {
    "ln1_g": "layernorm_before.weight", "ln1_b": "layernorm_before.bias",
    "q_w": "attention.q_proj.weight", "q_b": "attention.q_proj.bias",
    "k_w": "attention.k_proj.weight", "k_b": "attention.k_proj.bias",
    "v_w": "attention.v_proj.weight", "v_b": "attention.v_proj.bias",
    "o_w": "attention.o_proj.weight", "o_b": "attention.o_proj.bias",
    "ln2_g": "layernorm_after.weight", "ln2_b": "layernorm_after.bias",
    "fc1_w": "mlp.fc1.weight", "fc1_b": "mlp.fc1.bias",
    "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias",
}
