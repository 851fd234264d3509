"""CPU: a local CLIP checkpoint directory is mapped to tower shapes from its own config.json / weight shapes (width 768
alone cannot tell clip-vit-base-patch16 from -patch32), unsupported shapes are refused by name."""
import numpy as np
import pytest

from ivr_amd import config as C
from ivr_amd.compat import _clip_configs_from_checkpoint


def _sd(width, patch, layers, mlp, proj, tokens):
    sd = {"vision_model.embeddings.patch_embedding.weight": np.zeros((width, 3, patch, patch), np.float32),
          "vision_model.embeddings.position_embedding.weight": np.zeros((tokens, width), np.float32),
          "visual_projection.weight": np.zeros((proj, width), np.float32)}
    for i in range(layers):
        sd[f"vision_model.encoder.layers.{i}.mlp.fc1.weight"] = np.zeros((mlp, width), np.float32)
    return sd


def test_published_shapes_map_to_the_builtin_configs():
    vis, txt = _clip_configs_from_checkpoint("x", {"vision_config": {"hidden_size": 768}}, _sd(768, 32, 12, 3072, 512, 50))
    assert vis is C.CLIP_VIT_B32 and txt is C.CLIP_TEXT_B32
    hf = {"projection_dim": 768, "vision_config": {"hidden_size": 1024, "patch_size": 14, "num_hidden_layers": 24, "num_attention_heads": 16,
                                                    "intermediate_size": 4096, "image_size": 224},
          "text_config": {"hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 3072,
                          "max_position_embeddings": 77, "vocab_size": 49408, "eos_token_id": 2}}
    vis, txt = _clip_configs_from_checkpoint("x", hf, {})
    assert vis is C.CLIP_VIT_L14 and txt is C.CLIP_TEXT_L14


def test_patch16_is_not_mistaken_for_patch32():
    vis, txt = _clip_configs_from_checkpoint("x", {}, _sd(768, 16, 12, 3072, 512, 197))
    assert (vis.width, vis.patch, vis.tokens, vis.layers, vis.heads, vis.out_dim, vis.image) == (768, 16, 197, 12, 12, 512, 224)
    assert vis.pre_ln and not vis.patch_bias and vis.pool == C.POOL_CLS_POSTLN_PROJ
    assert txt is C.CLIP_TEXT_B32
    hf = {"projection_dim": 512, "vision_config": {"hidden_size": 768, "patch_size": 16, "num_hidden_layers": 12, "num_attention_heads": 12,
                                                    "intermediate_size": 3072, "image_size": 224}}
    assert _clip_configs_from_checkpoint("x", hf, {})[0] == vis


def test_unsupported_shapes_are_refused_by_name():
    with pytest.raises(ValueError, match="head_dim"):
        _clip_configs_from_checkpoint("ckpt", {"projection_dim": 512, "vision_config": {"hidden_size": 768, "patch_size": 32, "num_hidden_layers": 12,
                                                                                      "num_attention_heads": 8, "intermediate_size": 3072}}, {})
    with pytest.raises(ValueError, match="cannot read the vision tower"):
        _clip_configs_from_checkpoint("ckpt", {}, {})
