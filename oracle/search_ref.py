"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU restatement of the reference's normalise + exact inner-product top-k path:
  * `core.py:1176-1196`  FAISSRetriever._normalize_and_validate_features  -> normalize_rows_core
  * `unified_index.py:1776` faiss.normalize_L2 (in place)                 -> normalize_rows_faiss
  * `unified_index.py:503`, `core.py:891` faiss.IndexFlatIP.search        -> flat_ip_search
  * `unified_index.py:507-526` (score = 1 - ip, 0-based rank, stop at -1)   -> search_vectors_rows
  * `core.py:899-924` + `core.py:736-756` (recomputed clamped cosine,
    1-based rank)                                                          -> legacy_search_rows
  * `unified_builder.py:229-246` similarity_threshold filter                -> builder_filter
  * `video_frame_filter.py:63-70` keep-if-cos<0.98 against last kept       -> dedup_keep_mask

The arithmetic of IndexFlatIP lives in the un-vendored, un-pinned dependency
`faiss` (absent from this image).  Its published contract for a flat
inner-product index is restated: every query is scored against every stored
row, the k largest inner products are returned in descending order with int64
labels, and unused slots are labels -1 with distance -FLT_MAX.  Among equal
scores this restatement puts the LOWER id first (a rule of this build; the
reference never depends on FAISS tie order: it rewrites or recomputes scores).

Parity unpinned by the reference: it holds no golden vector for this path
(SURVEY.md §4/§8c).  tests/golden/search_*.npz pin this file against a float64
brute force on seeded, tie-free data.
"""
import numpy as np

NEG_FLT_MAX = np.float32(-3.4028234663852886e38)


def normalize_rows_core(features):
    """core.py:1176-1196: reshape 1-D, reject non-finite, x/||x|| with zero-norm rows divided by 1."""
    if not isinstance(features, np.ndarray):
        raise ValueError("Features must be numpy array")
    if features.size == 0:
        raise ValueError("Features array is empty")
    if features.ndim == 1:
        features = features.reshape(1, -1)
    elif features.ndim != 2:
        raise ValueError(f"Features must be 1D or 2D, got {features.ndim}D")
    if not np.isfinite(features).all():
        raise ValueError("Features contain NaN or infinite values")
    norms = np.linalg.norm(features, axis=1, keepdims=True)
    norms[norms == 0] = 1
    return features / norms


def normalize_rows_faiss(x):
    """faiss.normalize_L2 contract: in place on float32 [n,d]; all-zero rows stay zero."""
    assert x.dtype == np.float32 and x.ndim == 2
    n2 = np.einsum("ij,ij->i", x, x, dtype=np.float32)
    nz = n2 > 0
    x[nz] *= (np.float32(1.0) / np.sqrt(n2[nz]))[:, None]
    return x


def flat_ip_search(xb, xq, k, dtype=np.float32):
    """Exact inner-product top-k.  Returns (D float32 [nq,k], I int64 [nq,k]).

    Sorted by score descending, ties broken by ascending id; slots beyond ntotal
    are (-FLT_MAX, -1).  `dtype=np.float64` gives the brute-force ground truth
    used to pin ids on tie-free data."""
    xb = np.ascontiguousarray(xb, dtype=dtype)
    xq = np.ascontiguousarray(xq, dtype=dtype)
    if xq.ndim == 1:
        xq = xq.reshape(1, -1)
    nq, n = xq.shape[0], xb.shape[0]
    D = np.full((nq, k), NEG_FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0 or k == 0:
        return D, I
    kk = min(k, n)
    step = max(1, (1 << 26) // max(n, 1))       # bound the score block to ~256 MB
    for q0 in range(0, nq, step):
        s = xq[q0:q0 + step] @ xb.T                  # [q,n]
        if kk < n:
            part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
            # argpartition may cut through a tie group: pull in every id tied with the k-th score
            kth = np.take_along_axis(s, part, 1).min(axis=1)
            for r in range(s.shape[0]):
                row = s[r]
                cand = np.nonzero(row >= kth[r])[0]
                order = np.lexsort((cand, -row[cand]))[:kk]
                I[q0 + r, :kk] = cand[order]
                D[q0 + r, :kk] = row[cand[order]].astype(np.float32)
        else:
            for r in range(s.shape[0]):
                order = np.lexsort((np.arange(n), -s[r]))
                I[q0 + r, :kk] = order
                D[q0 + r, :kk] = s[r][order].astype(np.float32)
    return D, I


def search_vectors_rows(D_row, I_row, has_metadata=None, filter_func=None):
    """unified_index.py:507-526 on one query's (D,I): list of (rank, similarity_score, index)."""
    out = []
    for i, (dist, idx) in enumerate(zip(D_row, I_row)):
        if idx == -1:
            break
        if has_metadata is not None and not has_metadata(int(idx)):
            continue
        if filter_func is not None and not filter_func(int(idx)):
            continue
        out.append((i, float(1.0 - dist), int(idx)))
    return out


def proper_similarity(query_vec, target_vec):
    """core.py:736-756: cosine recomputed with np.dot / norms, clamped to [0,1]; 0.0 when features are missing."""
    if target_vec is None:
        return 0.0
    dot = np.dot(query_vec, target_vec)
    qn = np.linalg.norm(query_vec)
    tn = np.linalg.norm(target_vec)
    if qn == 0 or tn == 0:
        return 0.0
    return max(0.0, min(1.0, dot / (qn * tn)))


def legacy_search_rows(q_norm, D, I, stored_features):
    """core.py:899-924: flattened over queries, rank = position + 1, score = proper_similarity."""
    out = []
    for qi in range(D.shape[0]):
        for rank, idx in enumerate(I[qi]):
            if idx >= 0 and int(idx) in stored_features:
                out.append((rank + 1, proper_similarity(q_norm[qi], stored_features[int(idx)]), int(idx)))
    return out


def builder_filter(rows, similarity_threshold=0.0):
    """unified_builder.py:229: keep hits whose (1 - ip) score is >= threshold."""
    return [r for r in rows if r[1] >= similarity_threshold]


def cosine_1x1(a, b):
    """sklearn.metrics.pairwise.cosine_similarity([a],[b])[0][0]: rows L2-normalised (zero norm -> 1), then dot."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    na = np.sqrt(np.dot(a, a)) or 1.0
    nb = np.sqrt(np.dot(b, b)) or 1.0
    return float(np.dot(a / na, b / nb))


def dedup_keep_mask(embs, threshold=0.98):
    """video_frame_filter.py:63-70: frame t is kept iff cos(emb_t, emb_last_kept) < threshold
    (first frame always kept); the comparison state only advances on kept frames."""
    keep = np.zeros(len(embs), dtype=bool)
    prev = None
    for t, e in enumerate(embs):
        uniq = True
        if prev is not None and cosine_1x1(e, prev) >= threshold:
            uniq = False
        if uniq:
            prev = e
            keep[t] = True
    return keep


def merge_shards(D_parts, I_parts, k):
    """system.py:1744-1746 restated for (score,id) lists: concatenate per-shard candidates,
    sort by score descending (ties: lower id), keep k.  D_parts [G,nq,k], I_parts [G,nq,k] global ids."""
    G, nq, kk = D_parts.shape
    D = np.full((nq, k), NEG_FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        d = D_parts[:, q, :].reshape(-1)
        i = I_parts[:, q, :].reshape(-1)
        ok = i >= 0
        d, i = d[ok], i[ok]
        order = np.lexsort((i, -d))[:k]
        D[q, :len(order)] = d[order]
        I[q, :len(order)] = i[order]
    return D, I


# ---- adjacent consumers -------------------------------------------------------------------
def consecutive_similarities(embeddings):
    """filter.py:142-150."""
    return [cosine_1x1(embeddings[i - 1], embeddings[i]) for i in range(1, len(embeddings))]


def filter_similar_frames_in_scene(scene_embeddings, scene_indices, config):
    """filter.py:178-222, line by line."""
    if not config["enable_similarity_filtering"] or len(scene_embeddings) <= 1:
        return scene_indices
    thr, min_distance = config["similarity_threshold"], config["min_frame_distance"]
    kept, last = [0], 0
    for i in range(1, len(scene_embeddings)):
        if i - last < min_distance:
            continue
        if cosine_1x1(scene_embeddings[i], scene_embeddings[last]) < thr:
            kept.append(i)
            last = i
    if kept[-1] != len(scene_embeddings) - 1:
        kept.append(len(scene_embeddings) - 1)
    return [scene_indices[i] for i in kept]


def filter_similar_frames_advanced(scene_embeddings, scene_indices, config):
    """filter.py:224-258, line by line: keep frame i unless a KEPT frame j in [i - window, i) has cosine >= threshold
    (window = min(similarity_window_size, len(scene)), filter.py:233)."""
    if not config["enable_similarity_filtering"] or len(scene_embeddings) <= 1:
        return scene_indices
    thr = config["similarity_threshold"]
    window = min(config["similarity_window_size"], len(scene_embeddings))
    kept = [0]
    for i in range(1, len(scene_embeddings)):
        should_keep = True
        for j in range(max(0, i - window), i):
            if j in kept and cosine_1x1(scene_embeddings[i], scene_embeddings[j]) >= thr:
                should_keep = False
                break
        if should_keep:
            kept.append(i)
    return [scene_indices[i] for i in kept]


def group_into_scenes(transition_points, total_frames, min_length):
    """filter.py:160-176."""
    scenes, start = [], 0
    for t in transition_points:
        if t - start >= min_length:
            scenes.append((start, t - 1))
        start = t
    if total_frames - start >= min_length:
        scenes.append((start, total_frames - 1))
    return scenes


def keyframe_pipeline(quality_scores, embed_fn, config):
    """filter_transition_frames_for_video, filter.py:317-470, without its file I/O: per-frame quality scores (phase 1) ->
    thresholds (phase 2, np.percentile when adaptive) -> acceptance (phase 3; only accepted frames are embedded, embed_fn(i) may
    return None) -> consecutive cosines, scene cuts, scenes (phase 4) -> in-scene similarity filter, window variant when
    use_advanced_similarity_filtering (phase 5).  Returns the kept positions (indices into the input order) and the statistics
    the reference prints, or None where the reference returns None."""
    n = len(quality_scores)
    if n == 0:
        return None
    if config["enable_adaptive_filtering"]:
        bt = np.percentile([q["blur_score"] for q in quality_scores], config["blur_percentile"])
        et = np.percentile([q["edge_density"] for q in quality_scores], config["edge_percentile"])
    else:
        bt, et = config["blur_threshold"], config["edge_threshold"]
    stats = {"blur": 0, "low_edge": 0, "acceptable": 0, "embedding_error": 0}
    positions, embeddings = [], []
    for i, q in enumerate(quality_scores):
        reason = "acceptable"
        if config["enable_blur_detection"] and bt is not None and q["blur_score"] < bt:
            reason = "blur"
        elif config["enable_edge_detection"] and et is not None and q["edge_density"] < et:
            reason = "low_edge"
        if reason != "acceptable":
            stats[reason] += 1
            continue
        e = embed_fn(i)
        if e is None:
            stats["embedding_error"] += 1
            continue
        stats["acceptable"] += 1
        positions.append(i)
        embeddings.append(e)
    if len(embeddings) < config["min_scene_length"]:
        return None
    sims = consecutive_similarities(embeddings)
    transitions = [i + 1 for i, s_ in enumerate(sims) if s_ < config["transition_threshold"]]
    scenes = group_into_scenes(transitions, len(embeddings), config["min_scene_length"])
    if not scenes:
        return None
    kept = []
    for a, b in scenes:
        idx = list(range(a, b + 1))
        if not config["enable_similarity_filtering"]:
            kept.extend(idx)
        elif config.get("use_advanced_similarity_filtering", False):
            kept.extend(filter_similar_frames_advanced(embeddings[a:b + 1], idx, config))
        else:
            kept.extend(filter_similar_frames_in_scene(embeddings[a:b + 1], idx, config))
    return {"kept": [positions[i] for i in kept], "quality_stats": stats, "scenes": scenes, "transitions": transitions,
            "avg_similarity": float(np.mean(sims)) if sims else float("nan"), "blur_threshold": bt, "edge_threshold": et}


def similarity_graph(features, keys, top=10, threshold=0.7):
    """core.py:3513-3526 with sklearn's cosine_similarity restated (row-normalise, then X X^T)."""
    f = np.asarray(features, dtype=np.float64)
    n = np.linalg.norm(f, axis=1, keepdims=True)
    n[n == 0] = 1
    f = f / n
    sim = f @ f.T
    out = {}
    for i, key in enumerate(keys):
        order = np.argsort(sim[i])[::-1][1:top + 1]
        out[key] = [keys[j] for j in order if sim[i][j] > threshold]
    return out
