"""sparse-linear_amd — MI355X (gfx950) backend for the hot path of ttuegel/sparse-linear.

Layout:
  csrc/      hand-written HIP kernels + the extern "C" ABI (include/*.h)
  lib/       built shared library (git-ignored, travels with gpurun)
  _ffi.py    ctypes binding of the C ABI
  sparse.py  mirror of Data.Matrix.Sparse      (reference: sparse-linear/src/Data/Matrix/Sparse.hs)
  foreign.py mirror of Data.Matrix.Sparse.Foreign
  umfpack.py mirror of Numeric.LinearAlgebra.Umfpack
  dist.py    1-D row-block multi-GPU SpMV (torch.distributed, RCCL all-gather of y)
  feast.py   FEAST-style caller of the hot path (reference: feast/src/Numeric/LinearAlgebra/Feast.hs)

The directory name is not a Python identifier; ``__graft_entry__.load_package()``
registers it as the module ``sparse_linear_amd``.
"""
from . import _ffi  # noqa: F401
from ._ffi import BackendUnavailable, SparseLinearError  # noqa: F401
from .sparse import *  # noqa: F401,F403
from .sparse import DeviceMatrix, Matrix, SparseError  # noqa: F401
from .foreign import fromForeign, withConstMatrix  # noqa: F401
from . import dist  # noqa: F401
from . import umfpack  # noqa: F401
from . import feast  # noqa: F401
