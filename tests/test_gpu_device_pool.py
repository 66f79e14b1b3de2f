"""Large device blocks are kept on release and reused (csrc/device_pool.hip): same answers from a
reused block, the kept bytes are visible through spl_release_cached_memory, and switching the
pool off gives the memory back at once."""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import mat_to_tuple

pytestmark = pytest.mark.gpu


def _solve_once(pkg, O, m):
    n = m ** 3
    rp, ci, v = O.gen_poisson3d_csr(m)
    A = pkg.Matrix(n, n, rp, ci, v)
    xs = O.gen_vector(n)
    b = O.mulV(mat_to_tuple(A), xs)
    fact = pkg.umfpack.factor(A, pkg.umfpack.analyze(A))
    x = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackNormal, A, b)
    return x, xs


def test_band_factors_come_back_from_the_pool(gpu, pkg, O, monkeypatch):
    monkeypatch.setenv("SPL_LU_METHOD", "band")  # 40^3: band storage of 64 000 x 3 200 doubles = 1.6 GB
    pkg._ffi.release_cached_memory()
    x1, xs = _solve_once(pkg, O, 40)
    gc.collect()  # the Factors handle is gone: its band went to the pool, not to the driver
    kept = pkg._ffi.release_cached_memory()
    assert kept >= 1 << 30
    assert pkg._ffi.release_cached_memory() == 0
    x2, _ = _solve_once(pkg, O, 40)
    gc.collect()
    x3, _ = _solve_once(pkg, O, 40)  # this one runs in the block the second left behind
    assert np.array_equal(x1, x2) and np.array_equal(x2, x3)
    assert O.count_not_close(x1, xs, 1e-10) == 0
    gc.collect()
    assert pkg._ffi.release_cached_memory() >= 1 << 30


def test_pool_can_be_switched_off(gpu):
    code = (
        "import gc, numpy as np, torch\n"
        "from __graft_entry__ import load_package\n"
        "pkg = load_package(); torch.cuda.set_device(0)\n"
        "m = 40; n = m ** 3\n"
        "H = pkg.DeviceMatrix.synthetic('poisson3d', m); rp, ci, v = H.export_csr(); H.free()\n"
        "A = pkg.Matrix(n, n, rp, ci, v)\n"
        "x = pkg.umfpack.solve(A, np.ones(n))\n"
        "gc.collect()\n"
        "print('kept', pkg._ffi.release_cached_memory())\n"
    )
    env = dict(os.environ, SPL_CACHE_DEVICE_MEMORY="0", SPL_LU_METHOD="band")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "kept 0" in out.stdout
