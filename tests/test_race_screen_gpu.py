"""The hand-synchronised kernels (counted vmcnt + raw s_barrier, LDS-DMA rings) must reproduce their results bit for bit
across launches, also under load from a concurrent copy stream: tools/race_screen.py, short form."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launches_reproduce_bit_for_bit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py"), "24"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RACE SCREEN clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
