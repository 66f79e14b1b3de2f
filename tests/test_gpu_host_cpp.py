"""The compiled-language host side (sparse-linear_amd/host/DataMatrixSparse.hpp, the C++
mirror of Data.Matrix.Sparse / Numeric.LinearAlgebra.Umfpack) driving the C ABI from a plain
C++ process — the reference's hspec items and closed-form answers, no Python or torch involved."""
import os
import subprocess

import pytest


@pytest.mark.gpu
def test_cpp_selftest(gpu, pkg):
    import __graft_entry__ as g
    exe = os.path.join(g.PKG_DIR, "lib", "selftest")
    if not os.path.exists(exe):
        g.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


def test_cpp_selftest_builds_and_refuses_without_gpu(pkg):
    import __graft_entry__ as g
    exe = os.path.join(g.PKG_DIR, "lib", "selftest")
    if not os.path.exists(exe):
        g.build()
    assert os.path.exists(exe)
    if pkg._ffi.device_count() == 0:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "needs a GPU" in r.stdout
