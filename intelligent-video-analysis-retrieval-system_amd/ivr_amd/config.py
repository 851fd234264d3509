"""Encoder shape descriptors for the three towers the reference runs.

The reference never spells these shapes out: it loads them by hub name
(`core.py:1392` ViT-L/14, `system.py:1438` ViT-B/32 fallback,
`video_frame_filter.py:14` DINO ViT-S/16).  The numbers below are the
published architectures of those checkpoints.
"""
from dataclasses import dataclass, asdict

ACT_QUICK_GELU = 0  # x * sigmoid(1.702 x)   (CLIP)
ACT_GELU_ERF = 1    # 0.5 x (1 + erf(x/sqrt2)) (ViT / DINO)

POOL_CLS_POSTLN_PROJ = 0  # CLIP vision: CLS -> post-LN -> projection
POOL_LN_ALL_CLS = 1       # HF ViTModel: final LN on every token, take CLS, no projection
POOL_EOS_LN_PROJ = 2      # CLIP text: final LN, first-EOS token, projection


@dataclass(frozen=True)
class TowerConfig:
    name: str
    kind: str            # "vision" | "text"
    width: int           # D
    layers: int          # L
    heads: int           # H (head_dim = width // heads)
    mlp: int             # MLP hidden
    tokens: int          # T (vision: 1 + (image/patch)^2 ; text: context length)
    out_dim: int         # projection dim (0 = no projection, output is width)
    act: int = ACT_QUICK_GELU
    ln_eps: float = 1e-5
    pool: int = POOL_CLS_POSTLN_PROJ
    # vision only
    image: int = 224
    patch: int = 32
    pre_ln: bool = True      # CLIP has a LayerNorm right after the embeddings
    patch_bias: bool = False  # CLIP conv has no bias, HF ViT conv has one
    # text only
    vocab: int = 0
    eos_id: int = 49407
    causal: bool = False

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def embed_dim(self) -> int:
        return self.out_dim if self.out_dim else self.width

    def to_dict(self):
        return asdict(self)


CLIP_VIT_B32 = TowerConfig("clip-vit-b32", "vision", 768, 12, 12, 3072, 50, 512, image=224, patch=32)
CLIP_VIT_L14 = TowerConfig("clip-vit-l14", "vision", 1024, 24, 16, 4096, 257, 768, image=224, patch=14)
DINO_VIT_S16 = TowerConfig("dino-vit-s16", "vision", 384, 12, 6, 1536, 197, 0, act=ACT_GELU_ERF, ln_eps=1e-12,
                           pool=POOL_LN_ALL_CLS, image=224, patch=16, pre_ln=False, patch_bias=True)
CLIP_TEXT_B32 = TowerConfig("clip-text-b32", "text", 512, 12, 8, 2048, 77, 512, pool=POOL_EOS_LN_PROJ,
                            vocab=49408, causal=True)
CLIP_TEXT_L14 = TowerConfig("clip-text-l14", "text", 768, 12, 12, 3072, 77, 768, pool=POOL_EOS_LN_PROJ,
                            vocab=49408, causal=True)
# bring-up shape: small enough to dump every intermediate
TINY_VIT = TowerConfig("tiny-vit", "vision", 128, 2, 2, 256, 50, 64, image=224, patch=32)
TINY_TEXT = TowerConfig("tiny-text", "text", 128, 2, 2, 256, 16, 64, pool=POOL_EOS_LN_PROJ, vocab=512,
                        eos_id=511, causal=True)

BY_NAME = {c.name: c for c in (CLIP_VIT_B32, CLIP_VIT_L14, DINO_VIT_S16, CLIP_TEXT_B32, CLIP_TEXT_L14,
                               TINY_VIT, TINY_TEXT)}

# preprocessing constants the reference inherits from its HF processors
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
