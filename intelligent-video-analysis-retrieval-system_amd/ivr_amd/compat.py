"""Host-side mirror of the reference's hot-path interface.

The reference has no plugin API: `core.py`, `system.py` and `unified_builder.py` hold duck-typed
objects and call a handful of methods on them.  The classes below keep those names, argument
meanings, return shapes and error behaviour, and route the arithmetic to libivr_hip.so:

  reference object / function                              here
  ------------------------------------------------------  ---------------------------------------------
  core.CLIPFeatureExtractor            (core.py:1384)      CLIPFeatureExtractor
  core.FAISSRetriever                  (core.py:687)       FAISSRetriever
  core.KeyframeMetadata / SearchResult (core.py:84,161)    KeyframeMetadata / SearchResult
  unified_index.UnifiedIndex           (unified_index.py:63)   UnifiedIndex (+ README aliases build_index /
                                                               augmented_search, README.md:124-136)
  unified_builder.UnifiedBuilderIntegration (unified_builder.py:26)  UnifiedBuilderIntegration
  video_frame_filter.extract_embedding / extract_unique_frames (:28,:35)  same names, + FrameFilter alias
  faiss.IndexFlatIP / faiss.normalize_L2                   ivr_amd.index.IndexFlatIP / normalize_L2

Out of scope here, as in SURVEY.md section 8: HDF5/LZ4 container, thumbnails, resume, LLM, GUI, network.
The index container written by `create_unified_index` is a plain .npz (vectors + JSON metadata).
"""
import csv
import json
import os
import threading
import time
from dataclasses import asdict, dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import config as C
from .dedup import DedupState
from .index import FlatIPIndex, count_nonfinite_and_normalize
from .tower import Tower
from .weights import from_hf_state_dict, make_weights


# ----------------------------------------------------------------------------------------------
# data models (field layout of core.py:84-172; part of the boundary because callers read them)
# ----------------------------------------------------------------------------------------------
@dataclass
class KeyframeMetadata:
    folder_name: str
    image_name: str
    frame_id: int
    file_path: str
    sequence_position: int = 0
    total_frames: int = 0
    neighboring_frames: List[int] = None
    scene_boundaries: List[Tuple[int, int]] = None
    clip_features: Optional[np.ndarray] = None
    llm_description: Optional[str] = None
    detected_objects: List[str] = None
    scene_tags: List[str] = None
    confidence_score: float = 0.0
    similar_frames: List[str] = None
    transition_frames: List[str] = None

    def __post_init__(self):
        for f in ("neighboring_frames", "scene_boundaries", "detected_objects", "scene_tags", "similar_frames",
                  "transition_frames"):
            if getattr(self, f) is None:
                setattr(self, f, [])
        self._validate()

    def _validate(self):
        if not self.folder_name or not isinstance(self.folder_name, str):
            raise ValueError("folder_name must be a non-empty string")
        if not self.image_name or not isinstance(self.image_name, str):
            raise ValueError("image_name must be a non-empty string")
        if not isinstance(self.frame_id, int):
            raise ValueError("frame_id must be an integer")
        if not self.file_path or not isinstance(self.file_path, str):
            raise ValueError("file_path must be a non-empty string")

    def to_dict(self):
        data = asdict(self)
        if self.clip_features is not None:
            data["clip_features"] = np.asarray(self.clip_features).tolist()
        return data

    @classmethod
    def from_dict(cls, data):
        data = dict(data)
        if data.get("clip_features") is not None:
            data["clip_features"] = np.array(data["clip_features"])
        return cls(**data)

    def get_unique_key(self):
        return f"{self.folder_name}_{self.image_name}"


@dataclass
class SearchResult:
    metadata: KeyframeMetadata
    similarity_score: float
    rank: int
    query_relevance: float = 0.0
    temporal_context: List["SearchResult"] = None
    explanation: Optional[str] = None

    def __post_init__(self):
        if self.temporal_context is None:
            self.temporal_context = []


class _NullLogger:
    def __getattr__(self, _):
        return lambda *a, **k: None


def _frame_id_of(name):
    digits = "".join(ch for ch in os.path.splitext(name)[0] if ch.isdigit())
    return int(digits) if digits else 0


# ----------------------------------------------------------------------------------------------
# CLIPFeatureExtractor
# ----------------------------------------------------------------------------------------------
class ByteTokenizer:
    """Stand-in used only when no CLIP BPE vocabulary is available offline: [BOS, utf-8 bytes + 1 ..., EOS, EOS pad].
    Row format matches what the CLIP tokenizer emits (core.py:1532-1538: pad, truncate to 77, EOS-padded)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __call__(self, texts, max_length=77):
        T = min(max_length, self.cfg.tokens)
        ids = np.full((len(texts), T), self.cfg.eos_id, dtype=np.int64)
        for r, t in enumerate(texts):
            b = [1 + x for x in t.encode("utf-8")][:T - 2]
            ids[r, 0] = self.cfg.eos_id - 1
            ids[r, 1:1 + len(b)] = b
        return ids


def _load_local_checkpoint(path):
    """A local HuggingFace checkpoint directory (the reference's `paths.models`/clip_model override, system.py:1435):
    state dict from model.safetensors, or pytorch_model.bin through torch.load(weights_only=True) - loaders that execute
    nothing from the file.  Returns (state_dict of numpy arrays, parsed config.json or {})."""
    cfg = {}
    cj = os.path.join(path, "config.json")
    if os.path.isfile(cj):
        with open(cj, encoding="utf-8") as f:
            cfg = json.load(f)
    st = os.path.join(path, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.numpy import load_file
        return load_file(st), cfg
    pb = os.path.join(path, "pytorch_model.bin")
    if os.path.isfile(pb):
        sd = torch.load(pb, map_location="cpu", weights_only=True)
        return {k: v.float().numpy() for k, v in sd.items()}, cfg
    raise FileNotFoundError(f"{path}: neither model.safetensors nor pytorch_model.bin")


def _clip_configs_from_checkpoint(path, hf_cfg, sd):
    """(vision TowerConfig, text TowerConfig) of a local CLIP checkpoint, from its config.json (hidden_size, patch_size, image_size,
    num_hidden_layers, num_attention_heads, intermediate_size, projection_dim; text: the same + max_position_embeddings, vocab_size,
    eos_token_id) and, where config.json is missing or silent, from the weight shapes.  The shape is taken from the checkpoint, not
    guessed from the hidden width alone: clip-vit-base-patch16 and -patch32 share width 768.  Shapes the kernels do not run
    (head_dim != 64) are refused here, by name."""
    import dataclasses
    v, t = dict(hf_cfg.get("vision_config") or {}), dict(hf_cfg.get("text_config") or {})

    def shape(name):
        return tuple(np.asarray(sd[name]).shape) if name in sd else None

    pe = shape("vision_model.embeddings.patch_embedding.weight")            # [D, 3, P, P]
    pos = shape("vision_model.embeddings.position_embedding.weight")         # [T, D]
    fc1 = shape("vision_model.encoder.layers.0.mlp.fc1.weight")              # [mlp, D]
    vproj = shape("visual_projection.weight")                                 # [proj, D]
    width = int(v.get("hidden_size") or (pe[0] if pe else 0))
    patch = int(v.get("patch_size") or (pe[2] if pe else 0))
    layers = int(v.get("num_hidden_layers") or (1 + max((int(k.split(".")[3]) for k in sd if k.startswith("vision_model.encoder.layers.")),
                                                         default=-1)))
    mlp = int(v.get("intermediate_size") or (fc1[0] if fc1 else 0))
    proj = int(hf_cfg.get("projection_dim") or v.get("projection_dim") or (vproj[0] if vproj else 0))
    image = int(v.get("image_size") or (int(round((pos[0] - 1) ** 0.5)) * patch if pos and patch else 224))
    heads = int(v.get("num_attention_heads") or (width // 64 if width else 0))
    known = {(c[0].width, c[0].patch, c[0].layers, c[0].out_dim): c for c in CLIPFeatureExtractor.ARCH.values()}
    supported = "supported towers: any CLIP ViT with head_dim 64, width and mlp multiples of 64 (openai/clip-vit-base-patch32, " \
                "-base-patch16, -large-patch14, ...)"
    if not (width and patch and layers and mlp and proj and heads) or image % patch:
        raise ValueError(f"{path}: cannot read the vision tower's shape (hidden_size={width}, patch_size={patch}, layers={layers}, "
                         f"intermediate_size={mlp}, projection_dim={proj}, image_size={image}) from config.json / the weights; {supported}")
    if width != heads * 64 or width % 64 or mlp % 64 or proj % 16:
        raise ValueError(f"{path}: vision tower {width}-wide with {heads} heads (head_dim {width // max(heads, 1)}), mlp {mlp}, projection "
                         f"{proj} is not runnable by the HIP towers; {supported}")
    if (width, patch, layers, proj) in known:
        vis, txt = known[(width, patch, layers, proj)]
    else:
        base = C.CLIP_VIT_B32
        vis = dataclasses.replace(base, name=f"clip-vit-{width}-p{patch}", width=width, layers=layers, heads=heads, mlp=mlp,
                                  tokens=1 + (image // patch) ** 2, out_dim=proj, image=image, patch=patch)
        txt = None
    # the text tower (kept as published unless the checkpoint says otherwise)
    tw = int(t.get("hidden_size") or (shape("text_model.embeddings.token_embedding.weight") or (0, 0))[1])
    if tw:
        tfc1 = shape("text_model.encoder.layers.0.mlp.fc1.weight")
        tl = int(t.get("num_hidden_layers") or (1 + max((int(k.split(".")[3]) for k in sd if k.startswith("text_model.encoder.layers.")),
                                                         default=-1)))
        th = int(t.get("num_attention_heads") or tw // 64)
        tm = int(t.get("intermediate_size") or (tfc1[0] if tfc1 else 4 * tw))
        tt = int(t.get("max_position_embeddings") or (shape("text_model.embeddings.position_embedding.weight") or (77,))[0])
        vocab = int(t.get("vocab_size") or (shape("text_model.embeddings.token_embedding.weight") or (49408,))[0])
        eos = int(t.get("eos_token_id") if t.get("eos_token_id") is not None else vocab - 1)
        if eos >= vocab or eos == 2:          # published CLIP configs carry eos_token_id = 2 although the tokenizer emits vocab - 1
            eos = vocab - 1
        if tw != th * 64:
            raise ValueError(f"{path}: text tower {tw}-wide with {th} heads is not runnable by the HIP towers (head_dim must be 64)")
        ref = txt or C.CLIP_TEXT_B32
        cand = dataclasses.replace(ref, name=f"clip-text-{tw}", width=tw, layers=tl, heads=th, mlp=tm, tokens=tt, out_dim=proj, vocab=vocab,
                                   eos_id=eos)
        txt = ref if txt is not None and (ref.width, ref.layers, ref.mlp, ref.tokens, ref.out_dim, ref.vocab) == (
            cand.width, cand.layers, cand.mlp, cand.tokens, cand.out_dim, cand.vocab) else cand
    elif txt is None:
        txt = C.CLIP_TEXT_B32 if proj == 512 else C.CLIP_TEXT_L14
    return vis, txt


class CLIPFeatureExtractor:
    """core.py:1384.  Where the weights come from, in this order:
      * `weights` / `text_weights`: canonical float32 dicts (ivr_amd.weights) or a HF state dict;
      * `model_path` naming a local checkpoint directory (model.safetensors / pytorch_model.bin + config.json, the layout
        CLIPModel.from_pretrained reads at core.py:1442) - there is no network, hub names are never fetched;
      * `allow_random_init=True`: seeded random-init weights of the architecture `model_path` names.  For tests and
        benchmarks only: the embeddings are meaningless for retrieval, so this is never taken silently."""

    ARCH = {"openai/clip-vit-base-patch32": (C.CLIP_VIT_B32, C.CLIP_TEXT_B32),
            "openai/clip-vit-large-patch14": (C.CLIP_VIT_L14, C.CLIP_TEXT_L14)}

    def __init__(self, model_path="openai/clip-vit-large-patch14", config=None, logger=None, weights=None,
                 text_weights=None, tokenizer=None, max_batch=256, compute="bf16", seed=0, with_text=True,
                 allow_random_init=False, decode_workers=None, text_compute="f32"):
        self.config = config
        self.logger = logger or _NullLogger()
        self.model_path = model_path
        if not torch.cuda.is_available():
            raise RuntimeError("CLIPFeatureExtractor: no MI355X visible and there is no CPU fallback")
        self.device = "cuda"
        self.max_batch_size = 32          # core.py:1420 (the reference's processing granularity)
        self.max_text_length = 77
        local_sd = None
        if model_path in self.ARCH:
            vis_cfg, txt_cfg = self.ARCH[model_path]
        elif isinstance(model_path, str) and os.path.isdir(model_path):
            local_sd, hf_cfg = _load_local_checkpoint(model_path)
            vis_cfg, txt_cfg = _clip_configs_from_checkpoint(model_path, hf_cfg, local_sd)
        else:
            raise ValueError(f"unknown model_path {model_path!r}: expected one of {sorted(self.ARCH)} or a local checkpoint directory")
        self.vision_config, self.text_config = vis_cfg, txt_cfg
        self._lock = threading.RLock()    # encode_images is called from 4 worker threads (unified_index.py:773)
        self.decode_workers = decode_workers
        if weights is None and local_sd is not None:
            weights = local_sd
        if text_weights is None and local_sd is not None and with_text:
            text_weights = local_sd
        self.random_init = weights is None or (with_text and text_weights is None)
        if self.random_init:
            if not allow_random_init:
                raise RuntimeError(
                    f"CLIPFeatureExtractor({model_path!r}): no weights - pass weights= (HF state dict or canonical dict), a local "
                    "checkpoint directory as model_path, or allow_random_init=True (tests / benchmarks: embeddings of "
                    "random-init towers are meaningless for retrieval); hub names are not fetched, there is no network")
            self.logger.warning("CLIPFeatureExtractor: RANDOM-INIT weights (allow_random_init=True): not a trained model",
                                model_path=model_path, seed=seed)
        w = self._resolve(vis_cfg, weights, seed)
        self.vision_model = Tower(vis_cfg, w, max_batch=max_batch, compute=compute)
        self.text_model = None
        if with_text:
            tw = self._resolve(txt_cfg, text_weights, seed + 1)
            # Queries are a handful of rows per search (core.py:1504 encodes one string): the float32 tower costs 0.4 ms more per
            # query (1.0 vs 0.6 ms for ViT-B/32, tools/bench_query_latency.py) and keeps text-vs-image scores within the 1e-3 bound -
            # the bf16 text tower alone moves them by up to 1.1e-3.  text_compute="bf16" restores
            # the fast mode for bulk text encoding.
            self.text_model = Tower(txt_cfg, tw, max_batch=64, compute=text_compute)
        self.model = self.vision_model    # truthy: health check at system.py:263
        if tokenizer is None and isinstance(model_path, str) and os.path.isfile(os.path.join(model_path, "vocab.json")):
            try:                          # the checkpoint's own BPE tokenizer, local files only
                from transformers import CLIPTokenizerFast
                tk = CLIPTokenizerFast.from_pretrained(model_path, local_files_only=True)
                tokenizer = lambda texts, max_length=77: tk(list(texts), padding="max_length", truncation=True,   # noqa: E731
                                                            max_length=max_length, return_tensors="np")["input_ids"]
            except Exception as e:        # pragma: no cover - depends on the checkpoint directory
                self.logger.warning("CLIP tokenizer files found but not loadable", error=str(e))
        if tokenizer is None:
            if not self.random_init:
                self.logger.warning("CLIPFeatureExtractor: no CLIP BPE tokenizer given (tokenizer=): encode_text falls back to a "
                                    "byte-level stand-in whose ids do NOT match a trained text tower")
            tokenizer = ByteTokenizer(txt_cfg)
        self.processor = tokenizer
        self.nlp = None

    @staticmethod
    def _resolve(cfg, weights, seed):
        if weights is None:
            return make_weights(cfg, seed)
        if any(k.startswith(("vision_model.", "text_model.", "embeddings.")) for k in weights):
            return from_hf_state_dict(cfg, weights)
        return weights

    # -- text (core.py:1504-1554) --------------------------------------------------------------
    def _validate_and_clean_text(self, text):
        out = []
        for t in text:
            if isinstance(t, str) and t.strip():
                out.append(" ".join(t.split()))
        return out

    def encode_text(self, text: Union[str, List[str]], validate_input: bool = True) -> np.ndarray:
        if isinstance(text, str):
            text = [text]
        if validate_input:
            text = self._validate_and_clean_text(text)
        if len(text) == 0:
            raise ValueError("No valid text provided for encoding")
        if self.text_model is None:
            raise RuntimeError("Text encoding failed: extractor was built with with_text=False")
        try:
            with self._lock:
                ids = np.asarray(self.processor(list(text), max_length=self.max_text_length), dtype=np.int64)
                return self.text_model.encode_ids(ids, normalize=True).cpu().numpy()
        except Exception as e:
            raise RuntimeError(f"Text encoding failed: {e}")

    # -- images (core.py:1556-1641) ------------------------------------------------------------
    def _validate_image_paths(self, image_paths):
        return [p for p in image_paths if isinstance(p, str) and os.path.isfile(p)]

    def _load_and_validate_image(self, path):
        from PIL import Image
        img = Image.open(path).convert("RGB")             # core.py:1852
        if img.size[0] < 32 or img.size[1] < 32:           # core.py:1855
            return None
        return img

    def _decode_many(self, image_paths):
        """CPU decode on a worker pool (PIL releases the GIL while it decodes): [(path index, uint8 HWC array)] in path order,
        unreadable / too-small images dropped with a warning (core.py:1597-1609)."""
        from concurrent.futures import ThreadPoolExecutor

        def one(ip):
            i, p = ip
            try:
                im = self._load_and_validate_image(p)
                return (i, np.asarray(im)) if im is not None else None
            except Exception as e:
                self.logger.warning("Failed to load image", path=p, error=str(e))
                return None
        workers = self.decode_workers or min(16, len(os.sched_getaffinity(0)))
        if workers <= 1 or len(image_paths) < 4:
            done = [one(ip) for ip in enumerate(image_paths)]
        else:
            with ThreadPoolExecutor(max_workers=workers) as ex:
                done = list(ex.map(one, enumerate(image_paths), chunksize=8))
        return [d for d in done if d is not None]

    def _staging(self, nb, h, w, dev):
        """Two pinned host buffers + two device buffers per (batch, frame size), kept across calls (pinning hundreds of MB costs
        tens of milliseconds); the cache is dropped when it would pass 4 GiB."""
        cache = self.__dict__.setdefault("_stage_cache", {})
        key = (nb, h, w)
        if key not in cache:
            if sum(v[2] for v in cache.values()) + 4 * nb * h * w * 3 > (4 << 30):
                cache.clear()
            cache[key] = ([torch.empty((nb, h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)],
                          [torch.empty((nb, h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)], 4 * nb * h * w * 3)
        return cache[key][0], cache[key][1]

    def encode_images(self, image_paths: List[str], batch_size: int = 32, validate_files: bool = True,
                      show_progress: bool = True, return_kept: bool = False) -> np.ndarray:
        """core.py:1556.  Decode on CPU workers, then every frame size present in the call goes to the GPU in batches of up to
        `max_batch` frames: pinned double-buffered staging, H2D on a side stream under the previous batch's encode, resize + crop +
        normalise + encode in HBM, ONE D2H of all embeddings at the end.  A folder of mixed sizes costs one launch chain per
        (size, batch), not one per image.  return_kept=True also returns which inputs produced the rows (n may be < len(paths))."""
        if len(image_paths) == 0:
            raise ValueError("No image paths provided")
        if validate_files:
            image_paths = self._validate_image_paths(image_paths)
        if len(image_paths) == 0:
            raise ValueError("No valid image paths found")
        decoded = self._decode_many(image_paths)
        if not decoded:
            raise RuntimeError("No images were successfully encoded")
        n, D = len(decoded), self.vision_model.embed_dim
        by_shape: Dict[Tuple[int, int], List[int]] = {}
        for pos, (_, a) in enumerate(decoded):
            by_shape.setdefault(a.shape[:2], []).append(pos)
        dev = self.vision_model.device
        mb = self.vision_model.max_batch
        with self._lock, torch.cuda.device(dev):
            out_dev = torch.empty((n, D), dtype=torch.float32, device=dev)
            compute = torch.cuda.current_stream(dev)
            copy = self._copy_stream = getattr(self, "_copy_stream", None) or torch.cuda.Stream(device=dev)
            for (h, w), poss in by_shape.items():
                mode = "identity" if (h, w) == (self.vision_config.image,) * 2 else "shortest_edge_crop"
                nb = min(mb, len(poss))
                host, devb = self._staging(nb, h, w, dev)
                uploaded = [torch.cuda.Event() for _ in range(2)]      # H2D of buffer b finished (the pinned side may be refilled)
                consumed = [torch.cuda.Event() for _ in range(2)]      # the encode that read device buffer b finished
                used = [False, False]
                for bi, b0 in enumerate(range(0, len(poss), nb)):
                    sel = poss[b0:b0 + nb]
                    b = bi & 1
                    if used[b]:
                        uploaded[b].synchronize()
                    hv = host[b][:len(sel)].numpy()
                    for r, pos in enumerate(sel):
                        hv[r] = decoded[pos][1]
                    with torch.cuda.stream(copy):
                        if used[b]:
                            copy.wait_event(consumed[b])
                        devb[b][:len(sel)].copy_(host[b][:len(sel)], non_blocking=True)
                        uploaded[b].record(copy)
                    compute.wait_event(uploaded[b])
                    emb = self.vision_model.encode_frames(devb[b][:len(sel)], mode, C.CLIP_MEAN, C.CLIP_STD, normalize=True)
                    out_dev[torch.as_tensor(sel, device=dev)] = emb
                    consumed[b].record(compute)
                    used[b] = True
            out = out_dev.cpu().numpy()
        if return_kept:                      # positions (in image_paths after validation) of the rows returned, for the build driver
            return out, [i for i, _ in decoded]
        return out

    def encode_frames(self, frames_u8, mode="identity", bgr=False):
        """Device path without files: uint8 [n,h,w,3] -> float32 CUDA [n,D], rows L2-normalised."""
        with self._lock:
            return self.vision_model.encode_frames(frames_u8, mode, C.CLIP_MEAN, C.CLIP_STD, bgr=bgr, normalize=True)

    def extract_features_batch(self, keyframe_folder: str):
        """core.py:1643: walk <root>/<folder>/*.jpg|png -> (features [N,D], [KeyframeMetadata])."""
        paths, metas = [], []
        for folder in sorted(os.listdir(keyframe_folder)):
            fdir = os.path.join(keyframe_folder, folder)
            if not os.path.isdir(fdir):
                continue
            names = sorted((n for n in os.listdir(fdir) if n.lower().endswith((".jpg", ".jpeg", ".png"))),
                           key=lambda n: (_frame_id_of(n), n))
            for pos, n in enumerate(names):
                paths.append(os.path.join(fdir, n))
                metas.append(KeyframeMetadata(folder_name=folder, image_name=n, frame_id=_frame_id_of(n),
                                              file_path=os.path.join(fdir, n), sequence_position=pos,
                                              total_frames=len(names)))
        if not paths:
            raise ValueError(f"No keyframes found in {keyframe_folder}")
        feats = self.encode_images(paths, validate_files=False, show_progress=False)
        if len(feats) != len(metas):
            raise RuntimeError("some keyframes could not be decoded; features and metadata would be misaligned")
        for m, f in zip(metas, feats):
            m.clip_features = f                            # core.py:1737-1738
        return feats, metas


# ----------------------------------------------------------------------------------------------
# FAISSRetriever (legacy index, core.py:687)
# ----------------------------------------------------------------------------------------------
class FAISSRetriever:
    def __init__(self, config=None, logger=None, cache=None):
        self.config = config
        self.logger = logger or _NullLogger()
        self.index = None
        self.index_type = "IndexFlatIP"       # every configured type is coerced to exact IP (core.py:1205-1219)
        self.use_gpu = True
        self.dimension = None
        self.is_trained = False
        self.id_to_metadata = {}
        self.metadata_to_id = {}
        self.next_id = 0
        self._lock = threading.RLock()

    def _calculate_proper_similarity(self, query_vec, target_vec):
        """core.py:736-756, host side: k x d flops per search."""
        if target_vec is None:
            return 0.0
        dot = np.dot(query_vec, target_vec)
        qn, tn = np.linalg.norm(query_vec), np.linalg.norm(target_vec)
        if qn == 0 or tn == 0:
            return 0.0
        return max(0.0, min(1.0, dot / (qn * tn)))

    def _normalize_and_validate_features(self, features):
        """core.py:1176-1196 with the arithmetic in HBM; returns a float32 CUDA tensor."""
        if not isinstance(features, np.ndarray):
            raise ValueError("Features must be numpy array")
        if features.size == 0:
            raise ValueError("Features array is empty")
        if features.ndim == 1:
            features = features.reshape(1, -1)
        elif features.ndim != 2:
            raise ValueError(f"Features must be 1D or 2D, got {features.ndim}D")
        t = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).cuda()
        if count_nonfinite_and_normalize(t) != 0:
            raise ValueError("Features contain NaN or infinite values")
        return t

    def build_index(self, features, metadata_list, index_type=None, validate_consistency=True) -> None:
        if len(features) != len(metadata_list):
            raise ValueError(f"Features count ({len(features)}) != metadata count ({len(metadata_list)})")
        if len(features) == 0:
            raise ValueError("Cannot build index from empty feature set")
        with self._lock:
            self.id_to_metadata, self.metadata_to_id = {}, {}
            for i, m in enumerate(metadata_list):
                try:
                    m._validate()
                except Exception as e:
                    raise ValueError(f"Invalid metadata at index {i}: {e}")
                key = m.get_unique_key()
                if validate_consistency and key in self.metadata_to_id:
                    raise ValueError(f"Duplicate metadata key found: {key}")
                self.id_to_metadata[i] = m
                self.metadata_to_id[key] = i
            self.next_id = len(metadata_list)
            t = self._normalize_and_validate_features(features)
            self.dimension = t.shape[1]
            try:
                self.index = FlatIPIndex(self.dimension, capacity=t.shape[0])
                self.index.add(t)
                self.is_trained = True
            except Exception as e:
                raise RuntimeError(f"Failed to add vectors to index: {e}")
            if validate_consistency and self.index.ntotal != len(self.id_to_metadata):
                raise RuntimeError("Index validation failed: index size != metadata count")

    def search(self, query_features, k: int = 50, search_params=None, validate_results=True) -> List[SearchResult]:
        if not self.is_trained or not self.index:
            raise RuntimeError("Index not trained. Call build_index first.")
        if len(self.id_to_metadata) == 0:
            return []
        with self._lock:
            q = self._normalize_and_validate_features(query_features)
            if q.shape[1] != self.dimension:
                raise ValueError(f"Query dimension ({q.shape[1]}) != index dimension ({self.dimension})")
            if search_params:
                for param, value in search_params.items():
                    if hasattr(self.index, param):
                        setattr(self.index, param, value)
            try:
                D, I = self.index.search_device(q, k)
                similarities, indices = D.cpu().numpy(), I.cpu().numpy()
            except Exception as e:
                raise RuntimeError(f"Search operation failed: {e}")
            qn = q.cpu().numpy()
            results = []
            for i, (sims, idxs) in enumerate(zip(similarities, indices)):
                for rank, (_, idx) in enumerate(zip(sims, idxs)):
                    idx = int(idx)
                    if idx >= 0 and idx in self.id_to_metadata:
                        md = self.id_to_metadata[idx]
                        if validate_results:
                            try:
                                md._validate()
                            except Exception:
                                continue
                        s = self._calculate_proper_similarity(qn[i], md.clip_features)
                        results.append(SearchResult(metadata=md, similarity_score=s, rank=rank + 1, query_relevance=s))
            return results

    def save_index(self, index_path: str, validate_before_save: bool = True) -> None:
        """core.py:960: <index_path>/index.faiss (flat FAISS container, ivr_amd.faiss_io) + metadata.json."""
        if not self.is_trained:
            raise RuntimeError("Cannot save untrained index")
        if validate_before_save and self.index.ntotal != len(self.id_to_metadata):
            raise RuntimeError("Cannot save inconsistent index: index size != metadata count")
        from .faiss_io import write_flat_index
        os.makedirs(index_path, exist_ok=True)
        write_flat_index(os.path.join(index_path, "index.faiss"), self.index.reconstruct_n(0, self.index.ntotal), "ip")
        meta = {"version": "2.1", "created_at": time.time(),
                "id_to_metadata": {str(k): v.to_dict() for k, v in self.id_to_metadata.items()},
                "metadata_to_id": dict(self.metadata_to_id), "next_id": self.next_id, "dimension": self.dimension,
                "index_type": self.index_type, "is_trained": self.is_trained, "index_size": self.index.ntotal}
        tmp = os.path.join(index_path, "metadata.json.tmp")
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(meta, f, indent=2, ensure_ascii=False)
        os.replace(tmp, os.path.join(index_path, "metadata.json"))

    def load_index(self, index_path: str, validate_after_load: bool = True) -> None:
        """core.py:1041: the rows go straight into HBM (no FAISS object is rebuilt)."""
        from .faiss_io import read_flat_index
        ff = os.path.join(index_path, "index.faiss")
        if not os.path.exists(ff):
            raise FileNotFoundError(f"Index file not found: {ff}")
        mf = os.path.join(index_path, "metadata.json")
        if not os.path.exists(mf):
            raise FileNotFoundError(f"Metadata file not found: {mf}")
        vectors, _ = read_flat_index(ff)
        with open(mf, encoding="utf-8") as f:
            meta = json.load(f)
        with self._lock:
            self.id_to_metadata = {int(k): KeyframeMetadata.from_dict(v) for k, v in meta["id_to_metadata"].items()}
            self.metadata_to_id = dict(meta.get("metadata_to_id", {}))
            self.next_id = int(meta.get("next_id", len(self.id_to_metadata)))
            self.dimension = vectors.shape[1]
            self.index = FlatIPIndex(self.dimension, capacity=len(vectors))
            self.index.add(vectors)                          # stored rows are already normalised (core.py:809-827)
            self.is_trained = True
            if validate_after_load and self.index.ntotal != len(self.id_to_metadata):
                self.index, self.is_trained = None, False
                raise RuntimeError("Index and metadata are inconsistent. This suggests the system was not properly built or "
                                   "saved. Please rebuild the system from keyframes.")

    def search_by_id(self, metadata_key: str, k: int = 10):
        if metadata_key not in self.metadata_to_id:
            return []
        md = self.id_to_metadata.get(self.metadata_to_id[metadata_key])
        if md is None or md.clip_features is None:
            return []
        return self.search(md.clip_features, k)


# ----------------------------------------------------------------------------------------------
# UnifiedIndex (unified_index.py:63) - build / load / search_vectors, container out of scope
# ----------------------------------------------------------------------------------------------
class UnifiedIndex:
    def __init__(self, config=None, logger=None):
        self.config = config
        self.logger = logger
        self.lock = threading.RLock()
        self.is_loaded = False
        self.faiss_index = None
        self.metadata_list: List[Dict[str, Any]] = []
        self.metadata_cache: Dict[int, Dict[str, Any]] = {}
        self._vectors = None
        self.index_file = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if self.faiss_index is not None:
            self.faiss_index.close()
        self.faiss_index, self.is_loaded = None, False

    def _scan_files(self, keyframes_dir):
        found = []
        for root, _, files in os.walk(keyframes_dir):
            for n in files:
                if n.lower().endswith(".jpg"):
                    found.append(os.path.join(root, n))
        return sorted(found)                                # deterministic row order (hard part 7)

    def create_unified_index(self, keyframes_dir, clip_processor, output_file, csv_mappings=None, progress_callback=None,
                             resume_from_existing=False, chunk_size=1000):
        """unified_index.py:94 with the batch-1 thread pool (:759-812) replaced by whole-chunk GPU batches
        (SURVEY.md section 8f rank 1).  Returns the same stats keys (:127-138)."""
        t0 = time.time()
        stats = {"total_files": 0, "processed_files": 0, "skipped_files": 0, "failed_files": 0, "build_time": 0.0,
                 "file_size_mb": 0.0, "vectors_count": 0, "thumbnails_count": 0, "full_images_count": 0,
                 "resumed": False, "chunks_processed": 0}
        files = self._scan_files(keyframes_dir)
        stats["total_files"] = len(files)
        # rows go straight into one preallocated array (the reference's 2.2M-file build, logs/system_20250821.log:4, is 6.9 GB of
        # 768-d rows: no per-row Python objects, no final np.stack copy)
        V, nrows, metas = None, 0, []
        for c0 in range(0, len(files), chunk_size):
            chunk = files[c0:c0 + chunk_size]
            try:
                try:                                         # this build's extractor says which files it dropped
                    feats, pos = clip_processor.encode_images(chunk, validate_files=False, show_progress=False, return_kept=True)
                    kept = [chunk[i] for i in pos]
                    stats["failed_files"] += len(chunk) - len(kept)
                except TypeError:                            # any duck-typed clip_processor with the reference's signature
                    feats = clip_processor.encode_images(chunk, validate_files=False, show_progress=False)
                    if len(feats) != len(chunk):
                        raise RuntimeError("dropped images")
                    kept = chunk
            except Exception:                                # fall back to per-file calls to find the bad ones
                feats, kept = [], []
                for p in chunk:
                    try:
                        feats.append(clip_processor.encode_images([p], validate_files=False, show_progress=False)[0])
                        kept.append(p)
                    except Exception:
                        stats["failed_files"] += 1
                feats = np.stack(feats) if feats else np.zeros((0, 1), np.float32)
            if len(kept):
                feats = np.asarray(feats, dtype=np.float32)
                if V is None:
                    V = np.empty((len(files), feats.shape[1]), dtype=np.float32)
                V[nrows:nrows + len(kept)] = feats
                nrows += len(kept)
            for p in kept:
                rel = os.path.relpath(p, keyframes_dir)
                folder = os.path.dirname(rel) or os.path.basename(os.path.normpath(keyframes_dir))
                metas.append({"folder_name": folder, "image_name": os.path.basename(p), "frame_id": _frame_id_of(p),
                              "file_path": p, "vector_index": len(metas)})
            stats["processed_files"] += len(kept)
            stats["chunks_processed"] += 1
            if progress_callback:
                progress_callback(stats["processed_files"], len(files))
        V = V[:nrows] if V is not None else np.zeros((0, 1), np.float32)
        np.savez(output_file if output_file.endswith(".npz") else output_file + ".npz", vectors=V,
                 metadata=np.frombuffer(json.dumps(metas).encode(), dtype=np.uint8),
                 csv_mappings=np.frombuffer(json.dumps(csv_mappings or {}).encode(), dtype=np.uint8))
        out = output_file if output_file.endswith(".npz") else output_file + ".npz"
        stats["vectors_count"] = len(V)
        stats["file_size_mb"] = os.path.getsize(out) / 1e6
        stats["build_time"] = time.time() - t0
        self._install(V, metas, out)
        return stats

    build_index = create_unified_index                      # README.md:124-136 alias

    def _install(self, V, metas, path):
        with self.lock:
            # no second host copy of the rows: `.vectors` (a reference attribute) is rebuilt from the device index on first access
            # (host copy + fp32 tiles + bf16 copy were 2.5x the rows, 17 GB at the reference's 2.2M x 768)
            self._vectors = None
            self._nvectors = int(len(V))
            self.metadata_list = metas
            self.metadata_cache = {}
            if self.faiss_index is not None:
                self.faiss_index.close()
            self.faiss_index = FlatIPIndex(V.shape[1] if V.size else 1, capacity=len(V))
            if len(V):
                # unified_index.py:1770-1779: normalize_L2 then add, here fused in the append kernel
                self.faiss_index.add(V, normalize=True)
            self.index_file = path
            self.is_loaded = True

    @property
    def vectors(self):
        """The stored rows as a host array [n,d] (unified_index.py keeps them in its HDF5 dataset): read back from the device index
        on first access - L2-normalised there (unified_index.py:1776 normalises before adding) - and cached until the next build."""
        if self._vectors is None and self.faiss_index is not None and self.is_loaded:
            self._vectors = self.faiss_index.reconstruct_n(0, self.faiss_index.ntotal) if self.faiss_index.ntotal else np.zeros((0, 1), np.float32)
        return self._vectors

    def load_unified_index(self, index_file):
        path = index_file if index_file.endswith(".npz") else index_file + ".npz"
        z = np.load(path, allow_pickle=False)
        metas = json.loads(bytes(z["metadata"]).decode())
        self._install(z["vectors"], metas, path)
        return {"vectors_count": self._nvectors, "metadata_count": len(metas), "index_file": path}

    def _get_metadata_cached(self, idx):
        idx = int(idx)
        if 0 <= idx < len(self.metadata_list):
            return self.metadata_list[idx]
        return None

    def search_vectors(self, query_vector, k: int = 50, filter_func: Callable = None) -> List[Dict[str, Any]]:
        """unified_index.py:480-538: 0-based rank, similarity_score = 1 - inner product."""
        if not self.is_loaded:
            raise ValueError("Index not loaded. Call load_unified_index() first.")
        distances, indices = self.faiss_index.search(np.asarray(query_vector, dtype=np.float32).reshape(1, -1), k)
        results = []
        for i, (dist, idx) in enumerate(zip(distances[0], indices[0])):
            if idx == -1:
                break
            md = self._get_metadata_cached(idx)
            if md is None:
                continue
            if filter_func and not filter_func(md):
                continue
            results.append({"rank": i, "similarity_score": float(1.0 - dist), "metadata": md, "index": int(idx)})
        return results

    def augmented_search(self, query, top_k=10, clip_processor=None):
        """README.md:152-158 alias: text (needs clip_processor) or vector query -> search_vectors."""
        if isinstance(query, str):
            if clip_processor is None:
                raise ValueError("augmented_search(text) needs clip_processor=")
            query = clip_processor.encode_text(query)[0]
        return self.search_vectors(np.asarray(query), k=top_k)

    def get_temporal_context(self, frame_index, window_size=3):
        return []                                            # always empty in the reference (unified_index.py:1221-1224)


def create_optimized_index(keyframes_dir, clip_processor, output_file, **kw):      # unified_index.py:1889
    ui = UnifiedIndex()
    return ui, ui.create_unified_index(keyframes_dir, clip_processor, output_file, **kw)


def load_optimized_index(index_file):                                               # unified_index.py:1922
    ui = UnifiedIndex()
    ui.load_unified_index(index_file)
    return ui


class UnifiedBuilderIntegration:
    """unified_builder.py:26: forwards build / load / search for an object holding `.clip_processor`."""

    def __init__(self, system):
        self.system = system
        self.unified_index: Optional[UnifiedIndex] = None

    def create_unified_index_fast(self, keyframes_dir, output_file, progress_callback=None, **kw):
        self.unified_index, stats = create_optimized_index(keyframes_dir, self.system.clip_processor, output_file,
                                                           progress_callback=progress_callback, **kw)
        return stats

    def load_unified_index_fast(self, index_file):
        self.unified_index = load_optimized_index(index_file)
        return True

    def search_unified_fast(self, query_vector, k: int = 50, similarity_threshold: float = 0.0):
        if not self.unified_index:
            raise ValueError("Unified index not loaded. Call load_unified_index_fast() first.")
        out = []
        for r in self.unified_index.search_vectors(query_vector, k=k, filter_func=lambda meta: True):
            if r["similarity_score"] >= similarity_threshold:          # unified_builder.py:229
                md = r["metadata"]
                legacy = KeyframeMetadata(folder_name=md["folder_name"], image_name=md["image_name"],
                                          frame_id=int(md["frame_id"]), file_path=md["file_path"])
                out.append({"metadata": legacy, "similarity_score": r["similarity_score"], "rank": r["rank"],
                            "temporal_context": self.unified_index.get_temporal_context(r["index"], 3),
                            "index": r["index"]})
        return out


def add_unified_index_support(system):                                              # unified_builder.py:427
    system.unified_builder = UnifiedBuilderIntegration(system)
    return system.unified_builder


class RAGBuilder:                                                                    # README.md:124-136
    def __init__(self, clip_processor):
        self.clip_processor = clip_processor

    def build_index(self, keyframes_dir, output_file="index.npz"):
        return create_optimized_index(keyframes_dir, self.clip_processor, output_file)[0]


class RAGRetriever:                                                                  # README.md:152-158
    def __init__(self, index: UnifiedIndex, clip_processor):
        self.index, self.clip_processor = index, clip_processor

    def search(self, text, top_k=10):
        return self.index.augmented_search(text, top_k=top_k, clip_processor=self.clip_processor)


# ----------------------------------------------------------------------------------------------
# video_frame_filter.py
# ----------------------------------------------------------------------------------------------
SIM_THRESHOLD = 0.98       # video_frame_filter.py:16
FRAME_SIZE = (224, 224)


class FrameFilter:
    """DINO ViT-S/16 CLS embedding + keep-if-cos<0.98-vs-last-kept, entirely in HBM (README alias of
    video_frame_filter.extract_unique_frames)."""

    def __init__(self, weights=None, mean=C.IMAGENET_MEAN, std=C.IMAGENET_STD, max_batch=256, compute="bf16", seed=0,
                 threshold=SIM_THRESHOLD, allow_random_init=False, logger=None):
        cfg = C.DINO_VIT_S16
        if isinstance(weights, str):                         # local facebook/dino-vits16 checkpoint directory
            weights, _ = _load_local_checkpoint(weights)
        if weights is None:
            if not allow_random_init:
                raise RuntimeError("FrameFilter: no weights - pass weights= (HF ViTModel state dict, canonical dict or a local "
                                   "checkpoint directory) or allow_random_init=True (tests / benchmarks only); "
                                   "video_frame_filter.py:24-25 fetches facebook/dino-vits16 by name, there is no network here")
            (logger or _NullLogger()).warning("FrameFilter: RANDOM-INIT DINO weights (allow_random_init=True)", seed=seed)
        w = make_weights(cfg, seed) if weights is None else (
            from_hf_state_dict(cfg, weights) if any(k.startswith(("embeddings.", "encoder.", "layers.")) for k in weights) else weights)
        self.tower = Tower(cfg, w, max_batch=max_batch, compute=compute)
        self.mean, self.std, self.threshold = mean, std, threshold
        self.state = DedupState(cfg.width)

    def extract_embedding(self, image) -> np.ndarray:
        """video_frame_filter.py:28: PIL image (already 224x224 RGB) -> CLS embedding [384] (not normalised)."""
        a = np.asarray(image.convert("RGB"))[None]
        mode = "identity" if a.shape[1:3] == (224, 224) else "stretch"
        return self.tower.encode_frames(a, mode, self.mean, self.std, normalize=False)[0].cpu().numpy()

    def filter_frames(self, frames_bgr, reset=True):
        """uint8 BGR frames [n,h,w,3] in decode order -> bool keep mask (video_frame_filter.py:58-70 for a batch)."""
        if reset:
            self.state.reset()
        keep = []
        for i in range(0, len(frames_bgr), self.tower.max_batch):
            emb = self.tower.encode_frames(frames_bgr[i:i + self.tower.max_batch], "stretch", self.mean, self.std, bgr=True,
                                           normalize=False)
            keep.append(self.state.keep_mask(emb, self.threshold))
        return torch.cat(keep).cpu().numpy().astype(bool) if keep else np.zeros(0, bool)

    def extract_frames(self, video_path, keyframe_root="keyframes", map_root="map", batch=64):
        return extract_unique_frames(video_path, keyframe_root, map_root, frame_filter=self, batch=batch)

    def apply_filters(self, frames, config=None, bgr=True, return_details=False):
        """README alias (README.md:196, SURVEY.md Fact 1 maps it to filter.py:317): the keyframe filter of
        filter_transition_frames_for_video over decoded frames - quality gating (blur / edge density) of all frames in batches,
        DINO embeddings of the accepted frames only, scene split, in-scene similarity filter - as batched GPU passes
        (ivr_amd.filters.filter_keyframes).  frames: list / array of uint8 [h,w,3] (bgr=True: cv2 order).  Returns the kept frames
        (or the details dict); [] where the reference returns None (too few acceptable frames, no valid scene)."""
        from .filters import filter_keyframes
        rgb = [None if f is None else np.ascontiguousarray(np.asarray(f)[..., ::-1] if bgr else np.asarray(f)) for f in frames]

        def embed(batch):
            # frames of one call may differ in size: the stretch resize is per frame size
            embs = [None] * len(batch)
            groups = {}
            for i, b in enumerate(batch):
                groups.setdefault(b.shape, []).append(i)
            for shape, idxs in groups.items():
                mode = "identity" if shape[:2] == (224, 224) else "stretch"
                e = self.tower.encode_frames(np.stack([batch[i] for i in idxs]), mode, self.mean, self.std, normalize=False)
                for j, i in enumerate(idxs):
                    embs[i] = e[j]
            return torch.stack(embs)
        res = filter_keyframes(rgb, embed, config, quality_batch=min(64, self.tower.max_batch))
        if return_details:
            return res
        return [] if res is None else [frames[i] for i in res["kept"]]


_default_filter = None


def set_default_frame_filter(frame_filter):
    """Install the FrameFilter behind the module-level extract_embedding / extract_unique_frames (the reference builds its
    model at import time from the hub, video_frame_filter.py:24-25; here the weights must be handed over explicitly)."""
    global _default_filter
    _default_filter = frame_filter


def _filter():
    if _default_filter is None:
        raise RuntimeError("no default FrameFilter: call set_default_frame_filter(FrameFilter(weights=...)) first")
    return _default_filter


def extract_embedding(image):                                                         # video_frame_filter.py:28
    return _filter().extract_embedding(image)


def extract_unique_frames(video_path, keyframe_root, map_root, frame_filter=None, batch=64):
    """video_frame_filter.py:35: decode with cv2 (CPU, as in the reference), embed + dedup on the GPU in batches,
    write kept frames as <count>.jpg and the CSV map (n, pts_time, fps, frame_idx as ints).  Returns #saved."""
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("extract_unique_frames needs OpenCV for video decode (cv2 is not installed); "
                           "use FrameFilter.filter_frames(frames) on decoded frames instead") from e
    ff = frame_filter or _filter()
    ff.state.reset()
    cap = cv2.VideoCapture(video_path)
    name = os.path.splitext(os.path.basename(video_path))[0]
    out_dir = os.path.join(keyframe_root, name)
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(map_root, exist_ok=True)
    fps = cap.get(cv2.CAP_PROP_FPS)
    count = saved = 0
    with open(os.path.join(map_root, f"{name}.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["n", "pts_time", "fps", "frame_idx"])
        buf, pts = [], []

        def flush():
            nonlocal saved, count
            if not buf:
                return
            keep = ff.filter_frames(np.stack(buf), reset=False)
            for fr, t, k in zip(buf, pts, keep):
                if k:
                    cv2.imwrite(os.path.join(out_dir, f"{count}.jpg"), fr)
                    wr.writerow([int(saved), int(t), int(fps), int(count)])
                    saved += 1
                count += 1
            buf.clear()
            pts.clear()

        while cap.isOpened():
            ret, frame = cap.read()
            if not ret:
                break
            buf.append(frame)
            pts.append(cap.get(cv2.CAP_PROP_POS_MSEC) / 1000.0)
            if len(buf) == batch:
                flush()
        flush()
    cap.release()
    return saved
