"""CPU suite for the round-3 oracle additions (oracle/search_ref.py): the window variant of the in-scene filter
(filter.py:224-258) and the keyframe pipeline (filter.py:317-470) on hand-computable patterns."""
import numpy as np

from oracle import search_ref as S

CFG = {"enable_similarity_filtering": True, "similarity_threshold": 0.95, "min_frame_distance": 1, "similarity_window_size": 2,
       "use_advanced_similarity_filtering": True, "transition_threshold": 0.5, "min_scene_length": 2, "enable_adaptive_filtering": False,
       "blur_threshold": 10.0, "edge_threshold": 5.0, "enable_blur_detection": True, "enable_edge_detection": True,
       "blur_percentile": 10.0, "edge_percentile": 10.0}


def unit(angle_deg):
    a = np.deg2rad(angle_deg)
    return np.array([np.cos(a), np.sin(a), 0.0], np.float32)


def test_window_filter_compares_with_kept_frames_only():
    # cos(18 deg) = 0.951 >= 0.95 > cos(19 deg): frames 1 and 2 sit 10 deg from their predecessor (dropped against a KEPT one only)
    e = [unit(0), unit(10), unit(20), unit(30), unit(31), unit(80)]
    # i=1: kept {0}: cos 10 deg -> drop.  i=2: window [0,2): only 0 is kept, cos 20 deg = 0.94 -> keep.  i=3: window [1,3): kept {2},
    # cos 10 deg -> drop.  i=4: window [2,4): kept {2}: cos 11 deg -> drop.  i=5: window [3,5): nothing kept -> keep.
    assert S.filter_similar_frames_advanced(e, list(range(10, 16)), CFG) == [10, 12, 15]
    # window 5 = min(5, len): frame 5 is compared with kept frame 2 (60 deg): still kept; frame 4 with kept 2 -> drop
    assert S.filter_similar_frames_advanced(e, list(range(6)), dict(CFG, similarity_window_size=5)) == [0, 2, 5]
    # unlike the chain variant, the scene's last frame is NOT forced in, and there is no minimum distance
    assert S.filter_similar_frames_advanced([unit(0), unit(1)], [7, 8], CFG) == [7]
    assert S.filter_similar_frames_in_scene([unit(0), unit(1)], [7, 8], CFG) == [7, 8]
    assert S.filter_similar_frames_advanced(e, [0, 1, 2, 3, 4, 5], dict(CFG, enable_similarity_filtering=False)) == [0, 1, 2, 3, 4, 5]


def test_keyframe_pipeline_phases():
    good, blurry, flat = {"blur_score": 50.0, "edge_density": 9.0}, {"blur_score": 2.0, "edge_density": 9.0}, {"blur_score": 50.0, "edge_density": 1.0}
    scores = [good, good, blurry, good, good, flat, good, good]
    emb = {0: unit(0), 1: unit(5), 3: unit(90), 4: unit(95), 6: unit(96), 7: None}
    calls = []

    def embed(i):
        calls.append(i)
        return emb[i]
    r = S.keyframe_pipeline(scores, embed, CFG)
    assert calls == [0, 1, 3, 4, 6, 7]                        # only accepted frames are embedded (filter.py:405-407)
    assert r["quality_stats"] == {"blur": 1, "low_edge": 1, "acceptable": 5, "embedding_error": 1}
    # accepted positions [0,1,3,4,6]; cosines 0.996, 0.087, 0.996, 0.9998 -> one cut before local index 2 -> scenes (0,1), (2,4)
    assert r["transitions"] == [2] and r["scenes"] == [(0, 1), (2, 4)]
    # window filter inside each scene: (0,1): 5 deg -> drop 1; (2,4): 3 kept, 4 (5 deg) dropped, 6 (6 deg from 3... window 2: kept {3}) dropped
    assert r["kept"] == [0, 3]
    chain = S.keyframe_pipeline(scores, lambda i: emb[i], dict(CFG, use_advanced_similarity_filtering=False))
    assert chain["kept"] == [0, 1, 3, 6]                      # chain variant keeps each scene's last frame
    assert S.keyframe_pipeline([blurry, flat, good], lambda i: emb[0], CFG) is None           # fewer acceptable frames than a scene needs
    assert S.keyframe_pipeline([], embed, CFG) is None
    adaptive = S.keyframe_pipeline(scores, lambda i: emb.get(i, unit(40)), dict(CFG, enable_adaptive_filtering=True, blur_percentile=20.0,
                                                                                 edge_percentile=20.0))
    assert adaptive["blur_threshold"] == np.percentile([s["blur_score"] for s in scores], 20.0)
    assert adaptive["quality_stats"]["blur"] == 1 and adaptive["quality_stats"]["low_edge"] == 1
