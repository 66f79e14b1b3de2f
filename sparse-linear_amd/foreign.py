"""Mirror of ``Data.Matrix.Sparse.Foreign`` (sparse-linear/src/Data/Matrix/Sparse/Foreign.hs).

``withConstMatrix`` is the reference's designated FFI seam: it narrows the
64-bit ``Int`` arrays to ``CInt`` copies and hands the 5-tuple
``(nrows, ncols, Ap, Ai, Ax)`` to a callback (Foreign.hs:24-41).  ``fromForeign``
builds a ``Matrix`` back from such a tuple (Foreign.hs:43-88).
"""
import numpy as np

from .sparse import Matrix


def withConstMatrix(mat, action):
    """action(nrows, ncols, ptrs_i32, rows_i32, vals_f64) -> result"""
    nrows, ncols, ap, ai, ax = mat._tuple32()
    return action(nrows, ncols, ap, ai, ax)


def fromForeign(copy, nrows, ncols, ptrs, rows, vals):
    """Adopt (copy=False) or copy a foreign CSC tuple.  The reference then runs
    dedupInPlace on every column and ignores its deletion count (Foreign.hs:74-78):
    sorted duplicate-free columns — the contract of every output of this
    backend — pass through unchanged; each column is sorted by row index."""
    nptrs = int(ncols) + 1
    p = np.array(ptrs[:nptrs], dtype=np.int64, copy=True)
    nz = int(p[-1])
    r = np.array(rows[:nz], dtype=np.int64, copy=bool(copy) or True)
    v = np.array(vals[:nz], dtype=np.float64, copy=bool(copy) or True)
    for m in range(int(ncols)):
        s, e = int(p[m]), int(p[m + 1])
        if e - s > 1 and np.any(r[s + 1:e] <= r[s:e - 1]):
            order = np.argsort(r[s:e], kind="stable")
            r[s:e] = r[s:e][order]
            v[s:e] = v[s:e][order]
            # duplicate rows: summed into the first, the rest get the sentinel
            # `nrows` and sink to the end of the column, exactly as dedupInPlace
            w = s
            for k in range(s + 1, e):
                if r[k] == r[w]:
                    v[w] = v[w] + v[k]
                    r[k] = nrows
                else:
                    w = k
            order = np.argsort(r[s:e], kind="stable")
            r[s:e] = r[s:e][order]
            v[s:e] = v[s:e][order]
    return Matrix(int(ncols), int(nrows), p, r, v)
