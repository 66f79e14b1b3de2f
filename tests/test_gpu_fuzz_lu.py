"""A short run of tools/fuzz_lu.py: random sparse systems through every LU path (band / multifrontal,
forced cut depths, dominant / SPD-like / general matrices, mesh patterns) against scipy's SuperLU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_lu_paths(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_lu.py"), "11", "120"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "120 cases, 0 failures" in r.stdout
