"""shared generators for the property tests — the reference's QuickCheck
generators (sparse-linear/tests/Test/LinearAlgebra.hs:17-38) as hypothesis strategies"""
import numpy as np
from hypothesis import strategies as st

# arbdim = arbitrary `suchThat` (> 0) (Test/LinearAlgebra.hs:26-27); QuickCheck sizes
# stay small, so do these
arbdim = st.integers(min_value=1, max_value=12)
# element type of almost every reference property is Int: integer-valued doubles
# keep every sum and product exact, so `===` stays meaningful in fp64
arbval = st.integers(min_value=-50, max_value=50).map(float)


@st.composite
def arbitrary_triples(draw, nr, nc):
    """nr*nc/4 + 1 uniformly random (r, c, x) triples (Test/LinearAlgebra.hs:29-38)"""
    k = nr * nc // 4 + 1
    rows = draw(st.lists(st.integers(0, nr - 1), min_size=k, max_size=k))
    cols = draw(st.lists(st.integers(0, nc - 1), min_size=k, max_size=k))
    vals = draw(st.lists(arbval, min_size=k, max_size=k))
    return list(zip(rows, cols, vals))


@st.composite
def arbitrary_dims_triples(draw):
    nr, nc = draw(arbdim), draw(arbdim)
    return nr, nc, draw(arbitrary_triples(nr, nc))


def csc_tuple_to_scipy(m):
    import scipy.sparse as sp
    nrows, ncols, p, i, x = m
    return sp.csc_matrix((np.asarray(x), np.asarray(i), np.asarray(p)), shape=(nrows, ncols))


def mat_to_tuple(M):
    """product Matrix -> oracle tuple"""
    return (M.nrows, M.ncols, M.pointers, M.indices, M.values)


def tuple_to_mat(pkg, m):
    nrows, ncols, p, i, x = m
    return pkg.Matrix(ncols, nrows, p, i, x)


def tuples_equal(a, b):
    return (a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
            and np.array_equal(a[4], b[4]))
