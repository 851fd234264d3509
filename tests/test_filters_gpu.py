"""GPU: the keyframe-filter / similarity-graph consumers vs the oracle."""
import numpy as np
import pytest

from oracle import search_ref as S

pytestmark = pytest.mark.gpu


def _walk(seed, n, d, step):
    rng = np.random.default_rng(seed)
    return (np.cumsum(rng.standard_normal((n, d)) * step, axis=0) + rng.standard_normal(d)).astype(np.float32)


def test_scene_pipeline_matches_oracle():
    from ivr_amd import filters as F
    E = np.concatenate([_walk(1, 40, 384, 0.05), _walk(2, 3, 384, 0.05) + 5, _walk(3, 60, 384, 0.05) - 5])
    sims = F.calculate_similarities(E)
    ref = S.consecutive_similarities(E)
    assert np.abs(np.array(sims) - np.array(ref)).max() < 1e-5
    cuts = F.detect_scene_transitions(sims, 0.75)
    assert cuts == [i + 1 for i, s in enumerate(ref) if s < 0.75] and 40 in cuts and 43 in cuts
    scenes = F.group_into_scenes(cuts, len(E), 5)
    assert scenes[0] == (0, 39) and scenes[-1][1] == len(E) - 1 and all(b - a + 1 >= 5 for a, b in scenes)
    cfg = {"enable_similarity_filtering": True, "similarity_threshold": 0.95, "min_frame_distance": 3}
    for a, b in scenes:
        idxs = list(range(a, b + 1))
        sub = E[a:b + 1]
        got = F.filter_similar_frames_in_scene(sub, idxs, cfg)
        want = S.filter_similar_frames_in_scene(sub, idxs, cfg)
        assert got == want and got[0] == a and got[-1] == b
    assert F.filter_similar_frames_in_scene(E[:1], [7], cfg) == [7]
    assert F.filter_similar_frames_in_scene(E[:5], [1, 2, 3, 4, 5], dict(cfg, enable_similarity_filtering=False)) == [1, 2, 3, 4, 5]


def test_similarity_graph_matches_oracle():
    from ivr_amd import filters as F
    rng = np.random.default_rng(5)
    centers = rng.standard_normal((6, 512)).astype(np.float32)
    feats = np.concatenate([c + 0.35 * rng.standard_normal((20, 512)).astype(np.float32) for c in centers])
    keys = [f"L01_{i:04d}.jpg" for i in range(len(feats))]
    got = F.similarity_graph(feats, keys)
    want = S.similarity_graph(feats, keys)
    assert got == want
    assert max(len(v) for v in got.values()) == 10 and all(k not in v for k, v in got.items())
    assert F.similarity_graph(feats[:1], keys[:1]) == {}


def test_scene_filter_is_one_launch_and_handles_edge_rules():
    """filter.py:178-222 corner cases against the line-by-line oracle: distance rule, nothing below the threshold (only first
    and last survive), every frame distinct, two-frame scenes."""
    from ivr_amd import filters as F
    rng = np.random.default_rng(11)
    base = rng.standard_normal((30, 384)).astype(np.float32)
    same = np.repeat(base[:1], 30, axis=0) + 1e-4 * rng.standard_normal((30, 384)).astype(np.float32)
    for E in (base, same, _walk(9, 50, 384, 0.08), base[:2], same[:2]):
        for thr, dist in ((0.95, 3), (0.95, 1), (0.5, 2), (1.01, 4), (-1.0, 2)):
            cfg = {"enable_similarity_filtering": True, "similarity_threshold": thr, "min_frame_distance": dist}
            idxs = list(range(100, 100 + len(E)))
            assert F.filter_similar_frames_in_scene(E, idxs, cfg) == S.filter_similar_frames_in_scene(E, idxs, cfg), (len(E), thr, dist)
