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


def test_fuzz_complex_paths(gpu):
    """tools/fuzz_complex.py: the umfpack_zi_* path (real embedding, static pivoting inside the 2 x 2
    diagonal blocks) on random complex systems against scipy's complex SuperLU"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_complex.py"), "5", "60"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "60 cases, 0 failures" in r.stdout
