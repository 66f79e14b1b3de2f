import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (test infrastructure)"""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    """the product package; building it is part of the CPU-side check"""
    import __graft_entry__ as g
    lib = os.path.join(g.PKG_DIR, "lib", "libsparse_linear_hip.so")
    if not os.path.exists(lib):
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def gpu(pkg):
    """GPU tests fail loudly (never skip, never fall back) when the HIP path is unusable"""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no GPU is visible"
    torch.cuda.set_device(0)
    pkg._ffi.require_gpu()
    return torch
