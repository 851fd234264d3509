import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "intelligent-video-analysis-retrieval-system_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must fail loudly, not skip, when the HIP library is missing on a GPU box; on a
    machine without a GPU they are deselected by `-m "not gpu"` and skipped otherwise."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def synth_frames(seed, n, h, w):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def smooth_frames(seed, n, h, w):
    """Same generator as tests/golden/make_golden.py (inputs are regenerated from seeds, not stored)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    for i in range(n):
        for c in range(3):
            f = rng.uniform(0.5, 6.0, 4)
            ph = rng.uniform(0, 6.28, 4)
            z = (np.sin(f[0] * xx / w * 6.28 + ph[0]) * np.cos(f[1] * yy / h * 6.28 + ph[1])
                 + 0.5 * np.sin(f[2] * (xx + yy) / (w + h) * 6.28 + ph[2]) + 0.2 * rng.standard_normal((h, w)))
            out[i, :, :, c] = np.clip(127.5 + 80 * z, 0, 255).astype(np.uint8)
    return out
