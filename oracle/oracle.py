"""ctypes front-end of the CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  See the header of ``sparse_oracle.c`` for what the
oracle restates and how it is pinned.

A CSC matrix is passed around as the tuple ``(nrows, ncols, pointers, indices,
values)`` with int64 / float64 numpy arrays — the fields of the reference's
``data Matrix`` (sparse-linear/src/Data/Matrix/Sparse.hs:67-76).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsparse_oracle.so")

I64 = np.int64
F64 = np.float64
_pI = C.POINTER(C.c_int64)
_pi = C.POINTER(C.c_int32)
_pD = C.POINTER(C.c_double)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("sparse_oracle.c", "lu_oracle.c", "cpu_fair.c")]
    srcs.append(os.path.join(_HERE, "..", "include", "spl_synth.h"))
    if not force and os.path.exists(_LIB_PATH):
        t = os.path.getmtime(_LIB_PATH)
        if all(not os.path.exists(s) or os.path.getmtime(s) <= t for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libsparse_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_gen_random_csr.restype = C.c_int64
        _lib.orc_gen_banded_csr.restype = C.c_int64
        _lib.orc_gen_poisson2d_csr.restype = C.c_int64
        _lib.orc_gen_poisson3d_csr.restype = C.c_int64
        _lib.orc_dedup_in_place.restype = C.c_int64
        _lib.orc_count_not_close.restype = C.c_int64
        _lib.orc_lu_factor.restype = C.c_void_p
    return _lib


def _I(a):
    return a.ctypes.data_as(_pI)


def _i(a):
    return a.ctypes.data_as(_pi)


def _D(a):
    return a.ctypes.data_as(_pD)


def _c64(a):
    return np.ascontiguousarray(a, dtype=I64)


def _cf(a):
    return np.ascontiguousarray(a, dtype=F64)


class OracleError(Exception):
    """Mirrors the reference's errorWithStackTrace / error call sites."""


def _mat(m):
    nrows, ncols, p, i, x = m
    return int(nrows), int(ncols), _c64(p), _c64(i), _cf(x)


def _take(ptr, n, dtype):
    """copy n items out of a malloc'd C array, then free it"""
    n = int(n)
    if n:
        ct = C.c_int64 if dtype is I64 else C.c_double
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()
    else:
        arr = np.zeros(0, dtype=dtype)
    lib().orc_free(ptr)
    return arr


# -- construction --------------------------------------------------------------

def compress(nrows, ncols, rows, cols, vals):
    rows, cols, vals = _c64(rows), _c64(cols), _cf(vals)
    if not (len(rows) == len(cols) == len(vals)):
        raise OracleError("compress: array lengths differ")
    nnz = len(rows)
    ptrs = np.zeros(ncols + 1, dtype=I64)
    idx = np.zeros(max(nnz, 1), dtype=I64)
    val = np.zeros(max(nnz, 1), dtype=F64)
    bad = C.c_int64(-1)
    st = lib().orc_compress(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(nnz), _I(rows), _I(cols),
                            _D(vals), _I(ptrs), _I(idx), _D(val), C.byref(bad))
    if st != 0:
        raise OracleError("compress: index out of bounds at %d" % bad.value)
    nz = int(ptrs[ncols])
    return (nrows, ncols, ptrs, idx[:nz].copy(), val[:nz].copy())


def fromTriples(nrows, ncols, triples):
    triples = list(triples)
    r = [t[0] for t in triples]
    c = [t[1] for t in triples]
    v = [t[2] for t in triples]
    return compress(nrows, ncols, r, c, v)


def dedup_in_place(idim, ixs, xs):
    ixs, xs = _c64(ixs).copy(), _cf(xs).copy()
    d = lib().orc_dedup_in_place(C.c_int64(idim), C.c_int64(len(ixs)), _I(ixs), _D(xs))
    return int(d), ixs, xs


def transpose(m):
    nrows, ncols, p, i, x = _mat(m)
    nz = int(p[ncols])
    pt = np.zeros(nrows + 1, dtype=I64)
    it = np.zeros(max(nz, 1), dtype=I64)
    xt = np.zeros(max(nz, 1), dtype=F64)
    lib().orc_transpose(C.c_int64(nrows), C.c_int64(ncols), _I(p), _I(i), _D(x), _I(pt), _I(it), _D(xt))
    return (ncols, nrows, pt, it[:nz].copy(), xt[:nz].copy())


def check_matrix(m):
    nrows, ncols, p, i, x = _mat(m)
    return int(lib().orc_check_matrix(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(len(p)), _I(p),
                                      C.c_int64(len(i)), _I(i), C.c_int64(len(x))))


# -- SpMV ------------------------------------------------------------------------

def axpy_(m, x, y):
    """in place y <- A x + y (Sparse.hs:433-453)"""
    nrows, ncols, p, i, v = _mat(m)
    x = _cf(x)
    assert y.dtype == F64 and y.flags.c_contiguous
    st = lib().orc_axpy_(C.c_int64(nrows), C.c_int64(ncols), _I(p), _I(i), _D(v), C.c_int64(len(x)),
                         _D(x), C.c_int64(len(y)), _D(y))
    if st != 0:
        raise OracleError("axpy_: dimension mismatch")


def axpy_z(m, x, y):
    """in place y <- A x + y on Complex Double (Sparse.hs:433-453 under the SPECIALIZE of :456-457): the
    matrix tuple carries complex values; x, y complex128 arrays"""
    nrows, ncols, p, i, v = m
    p, i = _c64(p), _c64(i)
    v = np.ascontiguousarray(v, dtype=np.complex128)
    x = np.ascontiguousarray(x, dtype=np.complex128)
    assert y.dtype == np.complex128 and y.flags.c_contiguous
    st = lib().orc_axpy_z(C.c_int64(nrows), C.c_int64(ncols), _I(p), _I(i), _D(v.view(F64)), C.c_int64(len(x)),
                          _D(x.view(F64)), C.c_int64(len(y)), _D(y.view(F64)))
    if st != 0:
        raise OracleError("axpy_: dimension mismatch")


def mulV_z(m, x):
    y = np.zeros(int(m[0]), dtype=np.complex128)
    axpy_z(m, x, y)
    return y


def mulV(m, x):
    nrows, ncols, p, i, v = _mat(m)
    x = _cf(x)
    y = np.zeros(nrows, dtype=F64)
    st = lib().orc_mulv(C.c_int64(nrows), C.c_int64(ncols), _I(p), _I(i), _D(v), C.c_int64(len(x)),
                        _D(x), _D(y))
    if st != 0:
        raise OracleError("axpy_: dimension mismatch")
    return y


def axpy(m, x, y):
    nrows, ncols, p, i, v = _mat(m)
    x, y = _cf(x), _cf(y)
    out = np.zeros(nrows, dtype=F64)
    st = lib().orc_axpy(C.c_int64(nrows), C.c_int64(ncols), _I(p), _I(i), _D(v), C.c_int64(len(x)),
                        _D(x), C.c_int64(len(y)), _D(y), _D(out))
    if st != 0:
        raise OracleError("axpy_: dimension mismatch")
    return out


def mulM(m, B):
    nrows, ncols, p, i, v = _mat(m)
    B = np.ascontiguousarray(B, dtype=F64)
    Cm = np.zeros((nrows, B.shape[1]), dtype=F64)
    st = lib().orc_mulm(C.c_int64(nrows), C.c_int64(ncols), _I(p), _I(i), _D(v), C.c_int64(B.shape[0]),
                        C.c_int64(B.shape[1]), _D(B), _D(Cm))
    if st != 0:
        raise OracleError("mulM: inner dimension mismatch")
    return Cm


def csr_gaxpy32(rowptr, colidx, val, x, y):
    """CSR row-gather y <- A x + y in the reference's order, int32 indices."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    val, x = _cf(val), _cf(x)
    assert y.dtype == F64 and y.flags.c_contiguous
    lib().orc_csr_gaxpy32(C.c_int64(len(rowptr) - 1), _i(rowptr), _i(colidx), _D(val), _D(x), _D(y))
    return y


def csr_spmv_omp(rowptr, colidx, val, x, y):
    lib().orc_csr_spmv_omp(C.c_int64(len(rowptr) - 1), _i(rowptr), _i(colidx), _D(val), _D(x), _D(y))
    return y


def csr_spmv_omp_timed(rowptr, colidx, val, x, y, reps=5):
    """the fair CPU line of bench.py: arrays re-placed by parallel first touch, best of `reps` runs (seconds)"""
    f = lib().orc_csr_spmv_omp_timed
    f.restype = C.c_double
    return float(f(C.c_int64(len(rowptr) - 1), C.c_int64(len(x)), _i(rowptr), _i(colidx), _D(val), _D(x), _D(y),
                   C.c_int(reps)))


def omp_threads():
    return int(lib().orc_omp_threads())


# -- SpGEMM / add ------------------------------------------------------------------

def mm(a, b, literal=False):
    ar, ac, ap, ai, ax = _mat(a)
    br, bc, bp, bi, bx = _mat(b)
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().orc_mm(C.c_int64(ar), C.c_int64(ac), _I(ap), _I(ai), _D(ax), C.c_int64(br), C.c_int64(bc),
                      _I(bp), _I(bi), _D(bx), C.c_int(1 if literal else 0), C.byref(cp), C.byref(ci),
                      C.byref(cx))
    if st != 0:
        raise OracleError("mm: inner dimension mismatch")
    ptrs = _take(cp, bc + 1, I64)
    nz = int(ptrs[bc])
    return (ar, bc, ptrs, _take(ci, nz, I64), _take(cx, nz, F64))


def mm_z(a, b):
    """mm on Complex Double (orc_mm_z): matrix tuples with complex values (real ones are promoted)"""
    ar, ac, ap, ai, ax = _matc(a)
    br, bc, bp, bi, bx = _matc(b)
    ax = np.ascontiguousarray(ax, dtype=np.complex128)
    bx = np.ascontiguousarray(bx, dtype=np.complex128)
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().orc_mm_z(C.c_int64(ar), C.c_int64(ac), _I(ap), _I(ai), _D(ax.view(F64)), C.c_int64(br), C.c_int64(bc),
                        _I(bp), _I(bi), _D(bx.view(F64)), C.byref(cp), C.byref(ci), C.byref(cx))
    if st != 0:
        raise OracleError("mm: inner dimension mismatch")
    ptrs = _take(cp, bc + 1, I64)
    nz = int(ptrs[bc])
    return (ar, bc, ptrs, _take(ci, nz, I64), _take(cx, 2 * nz, F64).view(np.complex128))


def lin(alpha, a, beta, b):
    ar, ac, ap, ai, ax = _mat(a)
    br, bc, bp, bi, bx = _mat(b)
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().orc_lin(C.c_double(alpha), C.c_int64(ar), C.c_int64(ac), _I(ap), _I(ai), _D(ax),
                       C.c_double(beta), C.c_int64(br), C.c_int64(bc), _I(bp), _I(bi), _D(bx),
                       C.byref(cp), C.byref(ci), C.byref(cx))
    if st != 0:
        raise OracleError("glin: dimension mismatch")
    ptrs = _take(cp, ac + 1, I64)
    nz = int(ptrs[ac])
    return (ar, ac, ptrs, _take(ci, nz, I64), _take(cx, nz, F64))


def lin_z(alpha, a, beta, b):
    """lin on Complex Double (orc_lin_z): complex scalars, matrix tuples with complex values (real ones are
    promoted, as the reference's types would demand)"""
    ar, ac, ap, ai, ax = _matc(a)
    br, bc, bp, bi, bx = _matc(b)
    ax = np.ascontiguousarray(ax, dtype=np.complex128)
    bx = np.ascontiguousarray(bx, dtype=np.complex128)
    al = np.array([complex(alpha).real, complex(alpha).imag], dtype=F64)
    be = np.array([complex(beta).real, complex(beta).imag], dtype=F64)
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().orc_lin_z(_D(al), C.c_int64(ar), C.c_int64(ac), _I(ap), _I(ai), _D(ax.view(F64)),
                         _D(be), C.c_int64(br), C.c_int64(bc), _I(bp), _I(bi), _D(bx.view(F64)),
                         C.byref(cp), C.byref(ci), C.byref(cx))
    if st != 0:
        raise OracleError("glin: dimension mismatch")
    ptrs = _take(cp, ac + 1, I64)
    nz = int(ptrs[ac])
    return (ar, ac, ptrs, _take(ci, nz, I64), _take(cx, 2 * nz, F64).view(np.complex128))


def add(a, b):
    return lin(1.0, a, 1.0, b)


def sub(a, b):
    return lin(1.0, a, -1.0, b)


# -- FFI seam ------------------------------------------------------------------------

def with_const_matrix(m):
    nrows, ncols, p, i, x = _mat(m)
    Ap = np.zeros(ncols + 1, dtype=np.int32)
    Ai = np.zeros(max(len(i), 1), dtype=np.int32)
    lib().orc_with_const_matrix(C.c_int64(ncols), _I(p), _I(i), _i(Ap), _i(Ai))
    return nrows, ncols, Ap, Ai[:len(i)], x.copy()


def from_foreign(nrows, ncols, Ap, Ai, Ax):
    Ap = np.ascontiguousarray(Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Ax = _cf(Ax)
    nz = int(Ap[ncols])
    p = np.zeros(ncols + 1, dtype=I64)
    i = np.zeros(max(nz, 1), dtype=I64)
    x = np.zeros(max(nz, 1), dtype=F64)
    lib().orc_from_foreign(C.c_int32(nrows), C.c_int32(ncols), _i(Ap), _i(Ai), _D(Ax), _I(p), _I(i), _D(x))
    return (nrows, ncols, p, i[:nz].copy(), x[:nz].copy())


# -- synthetic workloads -------------------------------------------------------------

def _gen2(fn, nrows_local, *args):
    rowptr = np.zeros(nrows_local + 1, dtype=I64)
    nnz = fn(*args, _I(rowptr), None, None)
    col = np.zeros(max(nnz, 1), dtype=np.int32)
    val = np.zeros(max(nnz, 1), dtype=F64)
    fn(*args, _I(rowptr), _i(col), _D(val))
    return rowptr, col[:nnz], val[:nnz]


def gen_random_csr(n, K=20, seed=0x5EED, row0=0, row1=None):
    row1 = n if row1 is None else row1
    return _gen2(lib().orc_gen_random_csr, row1 - row0, C.c_uint64(seed), C.c_int64(n), C.c_int(K),
                 C.c_int64(row0), C.c_int64(row1))


def gen_banded_csr(n, seed=0x5EED, row0=0, row1=None):
    row1 = n if row1 is None else row1
    return _gen2(lib().orc_gen_banded_csr, row1 - row0, C.c_uint64(seed), C.c_int64(n), C.c_int64(row0),
                 C.c_int64(row1))


def gen_poisson2d_csr(m):
    return _gen2(lib().orc_gen_poisson2d_csr, m * m, C.c_int64(m))


def gen_poisson3d_csr(m):
    return _gen2(lib().orc_gen_poisson3d_csr, m * m * m, C.c_int64(m))


def rmat_thresholds(a, b, c):
    s = float(1 << 32)
    return (min(int(a * s), 0xFFFFFFFF), min(int((a + b) * s), 0xFFFFFFFF),
            min(int((a + b + c) * s), 0xFFFFFFFF))


def gen_rmat_coo(scale, nedges, abc=(0.25, 0.25, 0.25), seed=0x5EED, e0=0):
    ta, tb, tc = rmat_thresholds(*abc)
    rows = np.zeros(max(nedges, 1), dtype=I64)
    cols = np.zeros(max(nedges, 1), dtype=I64)
    vals = np.zeros(max(nedges, 1), dtype=F64)
    lib().orc_gen_rmat_coo(C.c_uint64(seed), C.c_int(scale), C.c_uint32(ta), C.c_uint32(tb), C.c_uint32(tc),
                           C.c_int64(e0), C.c_int64(nedges), _I(rows), _I(cols), _D(vals))
    return rows[:nedges], cols[:nedges], vals[:nedges]


def gen_vector(n, seed=0xBEEF, j0=0, j1=None):
    j1 = n if j1 is None else j1
    x = np.zeros(j1 - j0, dtype=F64)
    lib().orc_gen_vector(C.c_uint64(seed), C.c_int64(j0), C.c_int64(j1), _D(x))
    return x


def csr_to_csc_tuple(nrows, ncols, rowptr, colidx, val):
    """CSR arrays of A are the CSC arrays of A^T; transpose gives CSC(A)."""
    at = (ncols, nrows, _c64(rowptr), _c64(colidx), _cf(val))
    return transpose(at)


# -- comparison ------------------------------------------------------------------------

def count_not_close(a, b, tol=1e-10):
    """feast/tests/test-feast.hs:17-19: x == y || |x-y|/|x+y| < tol"""
    a, b = _cf(a), _cf(b)
    assert a.shape == b.shape
    return int(lib().orc_count_not_close(C.c_int64(a.size), _D(a), _D(b), C.c_double(tol)))


def kronecker(a, b):
    """kronecker (Sparse.hs:597-634), numpy restatement: pointers = scan of lenA[ja]*lenB[jb];
    column (ja, jb): rows ia*nrows(B)+ib for every ia (outer) and ib (inner), values b*a."""
    ar, ac, ap, ai, ax = _mat(a)
    br, bc, bp, bi, bx = _mat(b)
    la, lb = np.diff(ap[:ac + 1]), np.diff(bp[:bc + 1])
    ptrs = np.concatenate([[0], np.cumsum(np.outer(la, lb).ravel())]).astype(I64)
    idx = np.zeros(int(ptrs[-1]), dtype=I64)
    val = np.zeros(int(ptrs[-1]), dtype=np.float64)
    for ja in range(ac):
        ia_, xa = ai[ap[ja]:ap[ja + 1]], ax[ap[ja]:ap[ja + 1]]
        for jb in range(bc):
            ib_, xb = bi[bp[jb]:bp[jb + 1]], bx[bp[jb]:bp[jb + 1]]
            s = int(ptrs[ja * bc + jb])
            n = len(ia_) * len(ib_)
            idx[s:s + n] = (ia_[:, None] * br + ib_[None, :]).ravel()
            val[s:s + n] = (xb[None, :] * xa[:, None]).ravel()
    return (ar * br, ac * bc, ptrs, idx, val)


def take_diag(m):
    """takeDiag (Sparse.hs:636-648): first stored entry with row == column, else 0"""
    nrows, ncols, p, i, x = _mat(m)
    out = np.zeros(min(nrows, ncols), dtype=np.float64)
    for c in range(len(out)):
        hit = np.nonzero(i[p[c]:p[c + 1]] == c)[0]
        if len(hit):
            out[c] = x[p[c] + hit[0]]
    return out


def hcat(mats):
    """hcat (Sparse.hs:504-522): pointers = scan of the concatenated column lengths, indices / values
    concatenated; tuples (nrows, ncols, pointers, indices, values)"""
    mats = [_matc(m) for m in mats]
    if not mats:
        raise OracleError("hcat: empty list")
    if any(m[0] != mats[0][0] for m in mats):
        raise OracleError("hcat: nrows mismatch")
    lens = np.concatenate([np.diff(m[2][:m[1] + 1]) for m in mats])
    ptrs = np.concatenate([[0], np.cumsum(lens)]).astype(I64)
    return (mats[0][0], sum(m[1] for m in mats), ptrs, np.concatenate([m[3] for m in mats]).astype(I64),
            np.concatenate([m[4] for m in mats]))


def vcat(mats):
    """vcat (Sparse.hs:528-559): pointers summed; column c = the blocks' columns c one after the other in list
    order, row indices offset by the heights above (copyWithOffset)"""
    mats = [_matc(m) for m in mats]
    if not mats:
        raise OracleError("vcat: empty list")
    ncols = mats[0][1]
    if any(m[1] != ncols for m in mats):
        raise OracleError("vcat: ncols mismatch")
    ptrs = np.sum([m[2][:ncols + 1] for m in mats], axis=0).astype(I64)
    nz = int(ptrs[-1])
    cplx = any(np.iscomplexobj(m[4]) for m in mats)
    idx = np.zeros(nz, dtype=I64)
    val = np.zeros(nz, dtype=np.complex128 if cplx else F64)
    offs = np.concatenate([[0], np.cumsum([m[0] for m in mats])])
    for c in range(ncols):
        ix = int(ptrs[c])
        for m, off in zip(mats, offs):
            a, b = int(m[2][c]), int(m[2][c + 1])
            idx[ix:ix + b - a] = m[3][a:b] + off
            val[ix:ix + b - a] = m[4][a:b]
            ix += b - a
    return (int(offs[-1]), ncols, ptrs, idx, val)


def zeros(nrows, ncols):
    return (nrows, ncols, np.zeros(ncols + 1, dtype=I64), np.zeros(0, dtype=I64), np.zeros(0, dtype=F64))  # :671-677


def fromBlocks(blocks):
    """fromBlocks = vcat . map hcat . adjustDims (Sparse.hs:563-587): None = a zero block whose height / width
    come from the other blocks of its block row / column"""
    rows = [[None if m is None else _matc(m) for m in r] for r in blocks]
    cols = [[r[c] for r in rows if c < len(r)] for c in range(max(len(r) for r in rows))]

    def spec(groups, k, what):
        out = []
        for g in groups:
            ds = [m[k] for m in g if m is not None]
            if not ds:
                raise OracleError("fromBlocks: underspecified " + what)
            if any(d != ds[0] for d in ds):
                raise OracleError("fromBlocks: incompatible " + what)
            out.append(ds[0])
        return out

    heights, widths = spec(rows, 0, "heights"), spec(cols, 1, "widths")
    return vcat([hcat([m if m is not None else zeros(heights[r], widths[c]) for c, m in enumerate(row)])
                 for r, row in enumerate(rows)])


def fromBlocksDiag(blocks):
    """fromBlocksDiag (Sparse.hs:589-597): blocks listed by (super-)diagonals: transpose, pad, rotate row n by n"""
    blocks = [list(b) for b in blocks]
    n = len(blocks)
    trans = [[b[i] for b in blocks if i < len(b)] for i in range(max(len(b) for b in blocks))]
    out = []
    for k, as_ in enumerate(trans):
        as_ = as_ + [None] * (n - len(as_))
        cut = len(as_) - k
        out.append(as_[cut:] + as_[:cut])
    return fromBlocks(out)


def blockDiag(mats):
    mats = list(mats)
    return fromBlocksDiag([mats] + [[None] * len(mats) for _ in range(len(mats) - 1)])  # :659-665


def _matc(m):
    """like _mat, but complex values stay complex"""
    nrows, ncols, p, i, x = m
    x = np.asarray(x)
    return (int(nrows), int(ncols), _c64(p), _c64(i), np.ascontiguousarray(x, dtype=np.complex128 if np.iscomplexobj(x) else F64))


# -- solve --------------------------------------------------------------------------------

def linear_solve(m, b, sys=0):
    """mat <\\> b (Umfpack.hs:48-50); sys 1 = UmfpackTrans."""
    nrows, ncols, p, i, x = _mat(m)
    assert nrows == ncols
    b = _cf(b)
    out = np.zeros(ncols, dtype=F64)
    st = lib().orc_linear_solve(C.c_int64(ncols), _I(p), _I(i), _D(x), C.c_int(sys), _D(b), _D(out))
    return out, int(st)
