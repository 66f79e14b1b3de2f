"""Host-side mirror of ``Data.Matrix.Sparse`` (sparse-linear/src/Data/Matrix/Sparse.hs).

Same names, argument order, meaning and error behaviour as the reference's
export list (Sparse.hs:13-30), so code and tests written against the Haskell
API read the same here.  ``Matrix`` is the reference's CSC record
(Sparse.hs:67-76) with 64-bit ``pointers`` / ``indices`` and fp64 ``values``.

Everything on the hot path — ``mulV``, ``axpy``, ``axpy_``, ``mulM``, ``mm`` /
``*``, ``lin`` / ``+`` / ``-``, ``transpose``, ``compress`` / ``fromTriples`` —
crosses the C ABI (include/sparse_linear_hip.h) into the gfx950 kernels, with
the int64 -> int32 narrowing of ``withConstMatrix`` (Foreign.hs:39-41) at the
seam.  There is no CPU implementation of those operations in this package: if
the shared library or a GPU is missing they raise.

The purely structural combinators (``hcat``, ``vcat``, ``fromBlocks*``,
``kronecker``, ``diag``, ``ident``, ``zeros``, ``takeDiag``, ``pack`` …) only
rearrange index arrays; they are host-side numpy here (SURVEY.md §8f rank 4
lists their device versions as "next").
"""
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from ._ffi import (SPL_ERROR_dimension_mismatch, SPL_ERROR_index_out_of_bounds, as_f64, as_i32, check,
                   lib, p_f64, p_i32)

I64 = np.int64
F64 = np.float64
C128 = np.complex128


class SparseError(ValueError):
    """The reference's ``errorWithStackTrace`` / ``error`` call sites."""


def _oops(fn, msg):
    raise SparseError("%s: %s" % (fn, msg))


def _frozen(a):
    """a read-only view of `a` (the caller's own array object keeps its flags); an array that is read-only already —
    the field of another Matrix — is passed on as the object it is, so matrices built from the same pattern share it"""
    if not a.flags.writeable:
        return a
    v = a.view()
    v.setflags(write=False)
    return v


def _sample(a, k=2048):
    n = len(a)
    if n <= k:
        return a.copy()
    return a[np.linspace(0, n - 1, k).astype(np.int64)]


def _pattern_fingerprint(pointers, indices):
    return (len(pointers), len(indices), _sample(pointers), _sample(indices))


def _same_fingerprint(a, b):
    return a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


class Matrix(object):
    """Matrix in compressed sparse column (CSC) format (Sparse.hs:67-76)."""

    __slots__ = ("ncols", "nrows", "pointers", "indices", "values", "_handle", "_narrowed", "__weakref__")

    def __init__(self, ncols, nrows, pointers, indices, values):
        self.ncols = int(ncols)
        self.nrows = int(nrows)
        # read-only views: a Matrix is a value (the fields of the Haskell record cannot change), and the FFI seam keeps
        # narrowed copies of the pattern (_tuple32) — an in-place edit through the Matrix raises instead of going unnoticed
        self.pointers = _frozen(np.ascontiguousarray(pointers, dtype=I64))
        self.indices = _frozen(np.ascontiguousarray(indices, dtype=I64))
        values = np.asarray(values)
        # Double, or Complex Double (the reference's two SPECIALIZE instances, Sparse.hs:456-457)
        self.values = np.ascontiguousarray(values, dtype=C128 if np.iscomplexobj(values) else F64)
        self._handle = None

    @property
    def is_complex(self):
        return self.values.dtype == C128

    # deriving Eq (Sparse.hs:78): structural equality, explicit zeros included
    def __eq__(self, other):
        if not isinstance(other, Matrix):
            return NotImplemented
        return (self.ncols == other.ncols and self.nrows == other.nrows
                and np.array_equal(self.pointers, other.pointers)
                and np.array_equal(self.indices, other.indices)
                and np.array_equal(self.values, other.values))

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return "Matrix {ncols = %d, nrows = %d, pointers = %s, indices = %s, values = %s}" % (
            self.ncols, self.nrows, self.pointers.tolist(), self.indices.tolist(), self.values.tolist())

    # instance Num (Sparse.hs:100-113)
    def __add__(self, other):
        return lin(1.0, self, 1.0, other)  # glin 0 (+) a (+) b

    def __sub__(self, other):
        return lin(1.0, self, -1.0, other)  # glin 0 (+) a (-) b

    def __mul__(self, other):
        return mm(self, other)

    def __neg__(self):
        return cmap(np.negative, self)

    def __abs__(self):
        return cmap(np.abs, self)

    def signum(self):
        return cmap(np.sign, self)

    # -- FFI seam -----------------------------------------------------------------------
    def _tuple32(self):
        """withConstMatrix's marshalling (Foreign.hs:24-41): fresh int32 copies.  Complex values
        cross as packed (re, im) pairs, the form the reference passes to umfpack_zi_* with the
        imaginary pointer NULL (Umfpack/Internal.hs:124-132)."""
        # the 5-tuple carries no lengths: the C side reads pointers[ncols] indices and values, so a last pointer
        # beyond the arrays (or a pointer array shorter than ncols + 1) would be a heap over-read before any
        # validation on the device could refuse the matrix
        if len(self.pointers) != self.ncols + 1:
            _oops("withConstMatrix", "pointers has %d entries for %d columns" % (len(self.pointers), self.ncols))
        nz = int(self.pointers[-1]) if len(self.pointers) else 0
        if nz < 0 or nz > len(self.indices) or nz > len(self.values):
            _oops("withConstMatrix", "last pointer %d, but %d indices and %d values" % (nz, len(self.indices), len(self.values)))
        vals = self.values.view(F64) if self.is_complex else self.values
        # The reference narrows its 64-bit Int arrays on EVERY call (Foreign.hs:39-41: 2.4 GB of conversion per FFI call at
        # config C2).  A Matrix is a value — the fields of the Haskell record cannot change — so the
        # mirror narrows once and keeps the int32 copies while `pointers` and `indices` are the objects they were
        # (round 4: at config C5 the copy was 30 ms of every 170 ms `linearSolve_`).  The fields are read-only views, so
        # an edit through the Matrix raises; an edit through another alias of the caller's buffer is looked for on every
        # call by a fingerprint of the pattern (lengths, last pointer, ~4 000 evenly spaced entries: microseconds) — a
        # mismatch narrows again.  (identity alone let a stale pattern through.)
        kept = getattr(self, "_narrowed", None)
        if (kept is None or kept[0] is not self.pointers or kept[1] is not self.indices
                or not _same_fingerprint(kept[4], _pattern_fingerprint(self.pointers, self.indices))):
            kept = (self.pointers, self.indices, as_i32(self.pointers), as_i32(self.indices),
                    _pattern_fingerprint(self.pointers, self.indices))
            try:
                self._narrowed = kept
            except AttributeError:  # (a subclass with __slots__)
                pass
        return (self.nrows, self.ncols, kept[2], kept[3], as_f64(vals))

    def _parts(self):
        """real and imaginary parts as two real matrices with the same pattern"""
        return (Matrix(self.ncols, self.nrows, self.pointers, self.indices, self.values.real.copy()),
                Matrix(self.ncols, self.nrows, self.pointers, self.indices, self.values.imag.copy()))

    def device_handle(self):
        """Upload once, reuse for every later SpMV (handle API)."""
        if self._handle is None:
            if self.is_complex:
                self._handle = DeviceMatrix.from_csc_complex(self)
            else:
                self._handle = DeviceMatrix.from_csc(self)
                self._handle.optimize()  # one-time analysis, like umfpack_*_symbolic
        return self._handle


class DeviceMatrix(object):
    """Owner of a ``void *H`` from ``spl_matrix_create*`` (UMFPACK-style handle:
    callee-allocated, freed through ``void **``; Umfpack.hs:63-65)."""
    ORDER_REFERENCE = 0
    ORDER_FREE = 1

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self._finalizer = weakref.finalize(self, DeviceMatrix._free, self._h)

    @staticmethod
    def _free(h):
        try:
            lib().spl_matrix_free(C.byref(h))
        except Exception:  # interpreter shutdown
            pass

    def free(self):
        self._finalizer()

    @property
    def handle(self):
        if not self._h.value:
            raise _ffi.SparseLinearError("DeviceMatrix", _ffi.SPL_ERROR_invalid_handle)
        return self._h

    @classmethod
    def from_csc(cls, mat, part=0, nparts=1):
        _ffi.require_gpu()
        nr, nc, ap, ai, ax = mat._tuple32()
        h = C.c_void_p()
        check("spl_matrix_create_rowblock",
              lib().spl_matrix_create_rowblock(nr, nc, p_i32(ap), p_i32(ai), p_f64(ax), part, nparts,
                                               C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_csc_complex(cls, mat):
        """Complex Double handle (spl_matrix_create_z): packed (re, im) values, native SpMV (csrc/spmv_z.hip)"""
        _ffi.require_gpu()
        nr, nc, ap, ai, az = mat._tuple32()  # complex values cross as packed pairs
        h = C.c_void_p()
        check("spl_matrix_create_z", lib().spl_matrix_create_z(nr, nc, p_i32(ap), p_i32(ai), p_f64(az), C.byref(h)))
        return cls(h.value)

    @property
    def is_complex(self):
        return int(lib().spl_matrix_is_complex(self.handle)) == 1

    @classmethod
    def from_csr(cls, nrows_global, ncols, rowptr, colidx, val, row0=0):
        _ffi.require_gpu()
        rp, ci, v = as_i32(rowptr), as_i32(colidx), as_f64(val)
        if len(rp) < 1 or int(rp[-1]) < 0 or int(rp[-1]) > len(ci) or int(rp[-1]) > len(v):
            _oops("from_csr", "last row pointer %s, but %d column indices and %d values" % (rp[-1] if len(rp) else None, len(ci), len(v)))
        h = C.c_void_p()
        check("spl_matrix_create_csr",
              lib().spl_matrix_create_csr(nrows_global, ncols, row0, len(rp) - 1, p_i32(rp), p_i32(ci),
                                          p_f64(v), C.byref(h)))
        return cls(h.value)

    @classmethod
    def synthetic(cls, kind, n_or_m, K=20, seed=0x5EED, row0=0, row1=None):
        """kind: 'random' | 'banded' | 'poisson2d' | 'poisson3d' (include/spl_synth.h)."""
        _ffi.require_gpu()
        kinds = {"random": 0, "banded": 1, "poisson2d": 2, "poisson3d": 3}
        k = kinds[kind]
        n = n_or_m ** 2 if k == 2 else n_or_m ** 3 if k == 3 else n_or_m
        row1 = n if row1 is None else row1
        h = C.c_void_p()
        check("spl_matrix_create_synthetic",
              lib().spl_matrix_create_synthetic(k, n_or_m, K, seed, row0, row1, C.byref(h)))
        return cls(h.value)

    @classmethod
    def rmat(cls, scale, edge_factor=32, abc=(0.25, 0.25, 0.25), seed=0x5EED):
        _ffi.require_gpu()
        h = C.c_void_p()
        check("spl_matrix_create_rmat",
              lib().spl_matrix_create_rmat(scale, edge_factor, abc[0], abc[1], abc[2], seed, C.byref(h)))
        return cls(h.value)

    def spgemm(self, other):
        """device-resident C = self * other; returns (DeviceMatrix C, number of products)"""
        h = C.c_void_p()
        prod = C.c_int64(0)
        check("spl_matrix_spgemm", lib().spl_matrix_spgemm(self.handle, other.handle, C.byref(h), C.byref(prod)))
        return DeviceMatrix(h.value), int(prod.value)

    def lin(self, alpha, other, beta):
        """device-resident alpha * self + beta * other (Sparse.hs:401-431): handle in, handle out"""
        h = C.c_void_p()
        a = (C.c_double * 2)(complex(alpha).real, complex(alpha).imag)
        b = (C.c_double * 2)(complex(beta).real, complex(beta).imag)
        check("spl_matrix_lin", lib().spl_matrix_lin(self.handle, a, other.handle, b, C.byref(h)))
        return DeviceMatrix(h.value)

    def to_complex(self):
        """the Complex Double handle (x :+ 0) of a real one"""
        h = C.c_void_p()
        check("spl_matrix_to_complex", lib().spl_matrix_to_complex(self.handle, C.byref(h)))
        return DeviceMatrix(h.value)

    def transpose(self):
        """device-resident transpose (Sparse.hs:301-329)"""
        h = C.c_void_p()
        check("spl_matrix_transpose", lib().spl_matrix_transpose(self.handle, C.byref(h)))
        return DeviceMatrix(h.value)

    @classmethod
    def compress_dev(cls, nrows, ncols, ntriples, rows_ptr, cols_ptr, vals_ptr):
        """COO triples in device memory (int32, int32, float64 device pointers) -> handle (Sparse.hs:184-280)"""
        _ffi.require_gpu()
        h = C.c_void_p()
        bad = C.c_int64(-1)
        st = lib().spl_matrix_compress_dev(nrows, ncols, ntriples, C.c_void_p(rows_ptr), C.c_void_p(cols_ptr),
                                           C.c_void_p(vals_ptr), C.byref(h), C.byref(bad))
        check("spl_matrix_compress_dev", st)
        return cls(h.value)

    def info(self):
        buf = (C.c_int64 * 8)()
        check("spl_matrix_info", lib().spl_matrix_info(self.handle, buf))
        keys = ("nrows_global", "ncols", "row0", "nrows_local", "nnz", "device", "blocked_rows", "blocked_cols_log2")
        return dict(zip(keys, [int(v) for v in buf]))

    def export_csr(self):
        inf = self.info()
        rp = np.zeros(inf["nrows_local"] + 1, dtype=I64)
        ci = np.zeros(max(inf["nnz"], 1), dtype=np.int32)
        vw = 2 if self.is_complex else 1
        v = np.zeros(max(inf["nnz"], 1) * vw, dtype=F64)
        check("spl_matrix_export_csr", lib().spl_matrix_export_csr(self.handle, _ffi.p_i64(rp), p_i32(ci), p_f64(v)))
        if vw == 2:
            return rp, ci[:inf["nnz"]], v[:2 * inf["nnz"]].view(np.complex128)
        return rp, ci[:inf["nnz"]], v[:inf["nnz"]]

    def export_csr_rows(self, row0, row1):
        """rows [row0, row1) of the device CSR image: (rowptr relative to the window, colidx, val)"""
        rp = np.zeros(row1 - row0 + 1, dtype=I64)
        st = lib().spl_matrix_export_csr_rows(self.handle, row0, row1, _ffi.p_i64(rp), 0, None, None)
        cnt = int(rp[-1] - rp[0])
        if cnt == 0:
            check("spl_matrix_export_csr_rows", st)
            return rp - rp[0], np.zeros(0, dtype=np.int32), np.zeros(0, dtype=F64)
        ci = np.zeros(cnt, dtype=np.int32)
        v = np.zeros(cnt, dtype=F64)
        check("spl_matrix_export_csr_rows",
              lib().spl_matrix_export_csr_rows(self.handle, row0, row1, _ffi.p_i64(rp), cnt, p_i32(ci), p_f64(v)))
        return rp - rp[0], ci, v

    def export_csc(self):
        """the block as the reference's CSC fields (device transpose, Sparse.hs:301-329)"""
        inf = self.info()
        cp = np.zeros(inf["ncols"] + 1, dtype=I64)
        ri = np.zeros(max(inf["nnz"], 1), dtype=np.int32)
        v = np.zeros(max(inf["nnz"], 1), dtype=F64)
        check("spl_matrix_export_csc", lib().spl_matrix_export_csc(self.handle, _ffi.p_i64(cp), p_i32(ri), p_f64(v)))
        return cp, ri[:inf["nnz"]], v[:inf["nnz"]]

    def set_variant(self, variant):
        check("spl_matrix_set_variant", lib().spl_matrix_set_variant(self.handle, int(variant)))

    def spmm_dev(self, b_ptr, c_ptr, k, accumulate=False, stream=0):
        """C = A B for row-major device arrays B (ncols x k), C (nrows_local x k); no sync"""
        check("spl_matrix_spmm_dev",
              lib().spl_matrix_spmm_dev(self.handle, C.c_void_p(b_ptr), C.c_void_p(c_ptr), int(k),
                                        1 if accumulate else 0, C.c_void_p(stream)))

    def optimize(self):
        """one-time analysis; may build the column-blocked image (csrc/spmv_blocked.hip)"""
        check("spl_matrix_optimize", lib().spl_matrix_optimize(self.handle))

    def build_blocked(self, rows_per_panel=0, cols_log2=0, unroll=0):
        check("spl_matrix_build_blocked",
              lib().spl_matrix_build_blocked(self.handle, rows_per_panel, cols_log2, unroll))

    def build_panel(self, rows_per_panel=0, cols_log2=0, unroll=0, form=0):
        """the column-sorted panel image (csrc/spmv_panel.hip): order-free sums, 1e-10 contract;
        form 1 / 2: one chunk per load, 1 / 2 index blocks per phase; 4 / 5: paired storage (default)"""
        check("spl_matrix_build_panel",
              lib().spl_matrix_build_panel(self.handle, rows_per_panel, cols_log2, unroll, form))

    def panel_errors(self):
        """1 when a bounded wait of the ring form gave up during the last panel SpMV (diagnostics)"""
        r = lib().spl_matrix_panel_errors(self.handle)
        if r < 0:
            check("spl_matrix_panel_errors", r)
        return r

    def set_spmv_order(self, order):
        """ORDER_REFERENCE (0, default): sums in the reference's order, bit-identical; ORDER_FREE (1):
        any order, rounding-level differences (include/sparse_linear_hip.h)"""
        check("spl_matrix_set_spmv_order", lib().spl_matrix_set_spmv_order(self.handle, int(order)))

    def set_reserved_cus(self, reserved):
        """CUs left free for a kernel beside the SpMV (call before optimize(); include/sparse_linear_hip.h)"""
        check("spl_matrix_set_reserved_cus", lib().spl_matrix_set_reserved_cus(self.handle, int(reserved)))

    def spmv_kernel(self):
        """0 CSR-stream, 8 column-blocked lockstep, 15 sliced ELL, 16 column-sorted panels"""
        return int(lib().spl_matrix_spmv_kernel(self.handle))

    def mulv(self, x):
        if self.is_complex:  # packed complex vectors; the lengths count entries
            x = np.ascontiguousarray(x, dtype=C128)
            y = np.zeros(self.info()["nrows_local"], dtype=C128)
            st = lib().spl_matrix_mulv(self.handle, len(x), p_f64(x.view(F64)), p_f64(y.view(F64)))
            if st == SPL_ERROR_dimension_mismatch:
                _oops("axpy_", "column dimension does not match operand dimension %d" % len(x))
            check("spl_matrix_mulv", st)
            return y
        x = as_f64(x)
        y = np.zeros(self.info()["nrows_local"], dtype=F64)
        st = lib().spl_matrix_mulv(self.handle, len(x), p_f64(x), p_f64(y))
        if st == SPL_ERROR_dimension_mismatch:
            _oops("axpy_", "column dimension does not match operand dimension %d" % len(x))
        check("spl_matrix_mulv", st)
        return y

    def gaxpy(self, x, y):
        if self.is_complex:
            x = np.ascontiguousarray(x, dtype=C128)
            assert y.dtype == C128 and y.flags.c_contiguous
            st = lib().spl_matrix_gaxpy(self.handle, len(x), p_f64(x.view(F64)), len(y), p_f64(y.view(F64)))
            if st == SPL_ERROR_dimension_mismatch:
                _oops("axpy_", "dimension does not match operand dimension")
            check("spl_matrix_gaxpy", st)
            return y
        x = as_f64(x)
        assert y.dtype == F64 and y.flags.c_contiguous
        st = lib().spl_matrix_gaxpy(self.handle, len(x), p_f64(x), len(y), p_f64(y))
        if st == SPL_ERROR_dimension_mismatch:
            _oops("axpy_", "dimension does not match operand dimension")
        check("spl_matrix_gaxpy", st)
        return y

    def spmv_dev(self, x_ptr, y_ptr, accumulate=False, stream=0):
        """device pointers (ints); enqueues on `stream` (a hipStream_t as int), no sync"""
        check("spl_matrix_spmv_dev",
              lib().spl_matrix_spmv_dev(self.handle, C.c_void_p(x_ptr), C.c_void_p(y_ptr),
                                        1 if accumulate else 0, C.c_void_p(stream)))


# ---- accessors ------------------------------------------------------------------------------

def nonZero(mat):
    return int(mat.pointers[-1])  # Sparse.hs:115-117


def cmap(f, mat):
    return Matrix(mat.ncols, mat.nrows, mat.pointers, mat.indices, f(mat.values))  # :119-121


def scale(x, mat):
    return cmap(lambda v: v * x, mat)  # :123-125


def slice(mat, c):  # noqa: A001 - the reference's name
    """(indices, values) of column c as a sparse vector of length nrows (Sparse.hs:175-182)."""
    if c >= mat.ncols:
        _oops("slice", "column out of range")
    s, e = int(mat.pointers[c]), int(mat.pointers[c + 1])
    return mat.nrows, mat.indices[s:e], mat.values[s:e]


def toColumns(mat):
    return [slice(mat, c) for c in range(mat.ncols)]  # :381-383


# ---- construction on the device ---------------------------------------------------------------

def compress(nrows, ncols, rows, cols, vals):
    """COO -> CSC, duplicates summed, explicit zeros kept (Sparse.hs:184-255)."""
    rows = np.asarray(rows)
    cols = np.asarray(cols)
    vals = np.asarray(vals)
    if np.iscomplexobj(vals) and len(rows) == len(cols) == len(vals):
        # complex addition is componentwise: the duplicate sums of the two real runs ARE the complex sums
        re = compress(nrows, ncols, rows, cols, vals.real)
        im = compress(nrows, ncols, rows, cols, vals.imag)
        return Matrix(ncols, nrows, re.pointers, re.indices, re.values + 1j * im.values)
    vals = np.asarray(vals, dtype=F64)
    if len(rows) != len(cols):
        _oops("compress", "row and column array lengths differ")
    if len(rows) != len(vals):
        _oops("compress", "row and value array lengths differ")
    _ffi.require_gpu()
    r32, c32 = _checked_i32(rows, nrows, "row"), _checked_i32(cols, ncols, "column")
    ap = np.zeros(ncols + 1, dtype=np.int32)
    ai, ax, bad = C.c_void_p(), C.c_void_p(), C.c_int64(-1)
    st = lib().spl_compress(nrows, ncols, len(r32), p_i32(r32), p_i32(c32), p_f64(as_f64(vals)), p_i32(ap),
                            C.byref(ai), C.byref(ax), C.byref(bad))
    if st == SPL_ERROR_index_out_of_bounds:
        k = bad.value
        if not (0 <= rows[k] < nrows):
            _oops("compress", "row index out of bounds (0,%d) at %d" % (nrows, k))
        _oops("compress", "column index out of bounds (0,%d) at %d" % (ncols, k))
    check("spl_compress", st)
    nz = int(ap[ncols])
    return Matrix(ncols, nrows, ap.astype(I64), _ffi.take_malloced(ai, nz, C.c_int, I64),
                  _ffi.take_malloced(ax, nz, C.c_double, F64))


def _checked_i32(a, bound, what):
    """int64 -> int32 narrowing that keeps out-of-range values out of range"""
    a = np.asarray(a, dtype=I64)
    lim = np.iinfo(np.int32)
    return np.ascontiguousarray(np.where((a < lim.min) | (a > lim.max), -1, a), dtype=np.int32)


def fromTriples(nr, nc, triples):
    triples = list(triples)
    rows = [t[0] for t in triples]
    cols = [t[1] for t in triples]
    vals = [t[2] for t in triples]
    return compress(nr, nc, rows, cols, vals)  # Sparse.hs:357-363


def transpose(mat):
    """Counting-sort transpose (Sparse.hs:301-329); also the CSC -> CSR converter."""
    if mat.is_complex:
        re, im = (transpose(m) for m in mat._parts())
        return Matrix(re.ncols, re.nrows, re.pointers, re.indices, re.values + 1j * im.values)
    _ffi.require_gpu()
    nr, nc, ap, ai, ax = mat._tuple32()
    nz = int(ap[nc])
    tp = np.zeros(nr + 1, dtype=np.int32)
    ti = np.zeros(max(nz, 1), dtype=np.int32)
    tx = np.zeros(max(nz, 1), dtype=F64)
    check("spl_transpose", lib().spl_transpose(nr, nc, p_i32(ap), p_i32(ai), p_f64(ax), p_i32(tp), p_i32(ti),
                                              p_f64(tx)))
    return Matrix(nr, nc, tp.astype(I64), ti[:nz].astype(I64), tx[:nz])


def ctrans(mat):
    """omap conj . transpose (Sparse.hs:371-375); conj is the identity on Double"""
    t = transpose(mat)
    return cmap(np.conj, t) if t.is_complex else t


def hermitian(mat):
    return ctrans(mat) == mat  # :377-379


# ---- SpMV ---------------------------------------------------------------------------------------

def axpy_(mat, xs, ys):
    """in place ys <- A xs + ys (Sparse.hs:433-453); ys is a float64 numpy array."""
    if len(xs) != mat.ncols:
        _oops("axpy_", "column dimension %d does not match operand dimension %d" % (mat.ncols, len(xs)))
    if len(ys) != mat.nrows:
        _oops("axpy_", "row dimension %d does not match result dimension %d" % (mat.nrows, len(ys)))
    want = C128 if mat.is_complex else F64
    if not (isinstance(ys, np.ndarray) and ys.dtype == want and ys.flags.c_contiguous):
        raise TypeError("axpy_: ys must be a contiguous %s array (it is updated in place)" % np.dtype(want).name)
    mat.device_handle().gaxpy(xs, ys)  # Complex Double: the native kernel of csrc/spmv_z.hip


def axpy(mat, x, y):
    dt = C128 if mat.is_complex else F64
    y = np.array(y, dtype=dt)  # U.thaw _y (Sparse.hs:459)
    axpy_(mat, np.asarray(x, dtype=dt), y)
    return y


def mulV(mat, x):
    if mat.is_complex or np.iscomplexobj(x):
        # Complex Double (Sparse.hs:465-466): native packed-complex kernel (csrc/spmv_z.hip); a real matrix
        # applied to a complex vector is promoted as the reference's types would demand
        x = np.ascontiguousarray(x, dtype=C128)
        if len(x) != mat.ncols:
            _oops("axpy_", "column dimension %d does not match operand dimension %d" % (mat.ncols, len(x)))
        m = mat if mat.is_complex else cmap(lambda v: v.astype(C128), mat)
        return m.device_handle().mulv(x)
    x = np.asarray(x, dtype=F64)
    if len(x) != mat.ncols:
        _oops("axpy_", "column dimension %d does not match operand dimension %d" % (mat.ncols, len(x)))
    return mat.device_handle().mulv(x)  # Sparse.hs:464-471


def mulVT(mat, x):
    """A^T x — a gather on the CSC arrays themselves."""
    _ffi.require_gpu()
    x = as_f64(x)
    nr, nc, ap, ai, ax = mat._tuple32()
    y = np.zeros(nc, dtype=F64)
    st = lib().spl_gaxpy_t(nr, nc, p_i32(ap), p_i32(ai), p_f64(ax), len(x), p_f64(x), nc, p_f64(y))
    if st == SPL_ERROR_dimension_mismatch:
        _oops("axpy_", "row dimension %d does not match operand dimension %d" % (nr, len(x)))
    check("spl_gaxpy_t", st)
    return y


def mulM(matA, matB):
    """sparse x dense (Sparse.hs:473-498); matB is a 2-D array (rows x cols)."""
    B = np.ascontiguousarray(matB, dtype=F64)
    if matA.ncols != B.shape[0]:
        _oops("mulM", "inner dimension mismatch")
    _ffi.require_gpu()
    nr, nc, ap, ai, ax = matA._tuple32()
    out = np.zeros((nr, B.shape[1]), dtype=F64)
    check("spl_mulm", lib().spl_mulm(nr, nc, p_i32(ap), p_i32(ai), p_f64(ax), B.shape[0], B.shape[1],
                                    p_f64(B), p_f64(out)))
    return out


# ---- SpGEMM / sparse add ---------------------------------------------------------------------------

def _take_matrix(where, st, nr, nc, cp, ci, cx):
    check(where, st)
    ncols = nc.value
    ptrs = _ffi.take_malloced(cp, ncols + 1, C.c_int, I64)
    nz = int(ptrs[ncols])
    return Matrix(ncols, nr.value, ptrs, _ffi.take_malloced(ci, nz, C.c_int, I64),
                  _ffi.take_malloced(cx, nz, C.c_double, F64))


def mm(matA, matB):
    """C = A B (Sparse.hs:691-702): union pattern, cancellation keeps a stored 0."""
    if matA.ncols != matB.nrows:
        _oops("mm", "inner dimension mismatch")
    if matA.is_complex or matB.is_complex:
        # Complex Double: pattern from the real kernels, values accumulated in the reference's order with
        # Data.Complex's arithmetic (csrc/spgemm_z.hip, spl_spgemm_z)
        _ffi.require_gpu()
        a = (matA if matA.is_complex else cmap(lambda v: v.astype(C128), matA))._tuple32()
        b = (matB if matB.is_complex else cmap(lambda v: v.astype(C128), matB))._tuple32()
        nr, nc = C.c_int(), C.c_int()
        cp, ci, cz = C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = lib().spl_spgemm_z(a[0], a[1], p_i32(a[2]), p_i32(a[3]), p_f64(a[4]), b[0], b[1], p_i32(b[2]),
                                p_i32(b[3]), p_f64(b[4]), C.byref(nr), C.byref(nc), C.byref(cp), C.byref(ci),
                                C.byref(cz))
        check("spl_spgemm_z", st)
        ncols = nc.value
        ptrs = _ffi.take_malloced(cp, ncols + 1, C.c_int, I64)
        nz = int(ptrs[ncols])
        return Matrix(ncols, nr.value, ptrs, _ffi.take_malloced(ci, nz, C.c_int, I64),
                      _ffi.take_malloced(cz, 2 * nz, C.c_double, F64).view(C128))
    _ffi.require_gpu()
    a, b = matA._tuple32(), matB._tuple32()
    nr, nc = C.c_int(), C.c_int()
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().spl_spgemm(a[0], a[1], p_i32(a[2]), p_i32(a[3]), p_f64(a[4]), b[0], b[1], p_i32(b[2]),
                          p_i32(b[3]), p_f64(b[4]), C.byref(nr), C.byref(nc), C.byref(cp), C.byref(ci),
                          C.byref(cx))
    return _take_matrix("spl_spgemm", st, nr, nc, cp, ci, cx)


def lin(alpha, matA, beta, matB):
    """alpha A + beta B (Sparse.hs:426-431 over glin :401-424)."""
    if matA.nrows != matB.nrows:
        _oops("glin", "row number mismatch")
    if matA.ncols != matB.ncols:
        _oops("glin", "column number mismatch")
    if matA.is_complex or matB.is_complex or np.iscomplexobj(alpha) or np.iscomplexobj(beta):
        # Complex Double (what Feast.hs:216 calls: `lin (-1) matA _ze matB`): the native packed-complex kernel,
        # Data.Complex's evaluation order (spl_lin_z)
        _ffi.require_gpu()
        a = (matA if matA.is_complex else cmap(lambda v: v.astype(C128), matA))._tuple32()
        b = (matB if matB.is_complex else cmap(lambda v: v.astype(C128), matB))._tuple32()
        al = np.array([complex(alpha).real, complex(alpha).imag], dtype=F64)
        be = np.array([complex(beta).real, complex(beta).imag], dtype=F64)
        nr, nc = C.c_int(), C.c_int()
        cp, ci, cz = C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = lib().spl_lin_z(p_f64(al), a[0], a[1], p_i32(a[2]), p_i32(a[3]), p_f64(a[4].view(F64)),
                             p_f64(be), b[0], b[1], p_i32(b[2]), p_i32(b[3]), p_f64(b[4].view(F64)),
                             C.byref(nr), C.byref(nc), C.byref(cp), C.byref(ci), C.byref(cz))
        check("spl_lin_z", st)
        ncols = nc.value
        ptrs = _ffi.take_malloced(cp, ncols + 1, C.c_int, I64)
        nz = int(ptrs[ncols])
        return Matrix(ncols, nr.value, ptrs, _ffi.take_malloced(ci, nz, C.c_int, I64),
                      _ffi.take_malloced(cz, 2 * nz, C.c_double, F64).view(C128))
    _ffi.require_gpu()
    a, b = matA._tuple32(), matB._tuple32()
    nr, nc = C.c_int(), C.c_int()
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().spl_lin(float(alpha), a[0], a[1], p_i32(a[2]), p_i32(a[3]), p_f64(a[4]), float(beta), b[0],
                       b[1], p_i32(b[2]), p_i32(b[3]), p_f64(b[4]), C.byref(nr), C.byref(nc), C.byref(cp),
                       C.byref(ci), C.byref(cx))
    return _take_matrix("spl_lin", st, nr, nc, cp, ci, cx)


# ---- structural combinators: concatenations are host-side index plumbing (pure data movement
# between host arrays); kronecker and takeDiag run on the device ----------------------------------

def _lengths(mat):
    return np.diff(mat.pointers)


def diag(values):
    values = np.asarray(values)
    values = values.astype(C128) if np.iscomplexobj(values) else values.astype(F64)
    n = len(values)
    return Matrix(n, n, np.arange(n + 1, dtype=I64), np.arange(n, dtype=I64), values)  # :650-657


def ident(n):
    return diag(np.ones(n, dtype=F64))  # :667-669


def zeros(nrows, ncols):
    return Matrix(ncols, nrows, np.zeros(ncols + 1, dtype=I64), np.zeros(0, dtype=I64),
                  np.zeros(0, dtype=F64))  # :671-677


def takeDiag(mat):
    """diagonal as a dense vector, 0 where nothing is stored (Sparse.hs:636-648); on the device"""
    if mat.is_complex:
        re, im = mat._parts()
        return takeDiag(re) + 1j * takeDiag(im)
    _ffi.require_gpu()
    a = mat._tuple32()
    out = np.zeros(min(mat.nrows, mat.ncols), dtype=F64)
    check("spl_take_diag", lib().spl_take_diag(a[0], a[1], p_i32(a[2]), p_i32(a[3]), p_f64(a[4]), p_f64(out)))
    return out


def _assemble(blocks, row_off, col_off, nrows, ncols):
    """place CSC blocks at (row_off, col_off) of an nrows x ncols result on the device (spl_assemble_blocks,
    csrc/assemble.hip): the one kernel pair behind hcat / vcat / fromBlocks / fromBlocksDiag"""
    _ffi.require_gpu()
    k = len(blocks)
    cplx = any(m.is_complex for m in blocks)
    keep = []  # the int32 / float64 copies must outlive the call
    nr, nc = (C.c_int * k)(), (C.c_int * k)()
    ro, co = (C.c_int * k)(*[int(v) for v in row_off]), (C.c_int * k)(*[int(v) for v in col_off])
    ap, ai, ax = (C.c_void_p * k)(), (C.c_void_p * k)(), (C.c_void_p * k)()
    for b, m in enumerate(blocks):
        vals = m.values.astype(C128) if cplx else m.values
        t = (as_i32(m.pointers), as_i32(m.indices), as_f64(vals.view(F64) if cplx else vals))
        keep.append(t)
        nr[b], nc[b] = m.nrows, m.ncols
        ap[b], ai[b], ax[b] = t[0].ctypes.data, t[1].ctypes.data, t[2].ctypes.data
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    fn = lib().spl_assemble_blocks
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                   C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int,
                   C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    st = fn(k, nr, nc, ap, ai, ax, 2 if cplx else 1, ro, co, int(nrows), int(ncols), C.byref(cp), C.byref(ci), C.byref(cx))
    check("spl_assemble_blocks", st)
    ptrs = _ffi.take_malloced(cp, ncols + 1, C.c_int, np.int32)
    nz = int(ptrs[-1])
    idx = _ffi.take_malloced(ci, max(nz, 1), C.c_int, np.int32)[:nz]
    val = _ffi.take_malloced(cx, max(nz, 1) * (2 if cplx else 1), C.c_double, F64)[:nz * (2 if cplx else 1)]
    return Matrix(ncols, nrows, ptrs, idx, val.view(C128) if cplx else val)


def hcat(mats):
    """hcat (Sparse.hs:504-522), assembled on the device"""
    mats = list(mats)
    if not mats:
        _oops("hcat", "empty list")
    if any(m.nrows != mats[0].nrows for m in mats):
        _oops("hcat", "nrows mismatch")
    widths = [m.ncols for m in mats]
    return _assemble(mats, [0] * len(mats), np.concatenate([[0], np.cumsum(widths)])[:-1], mats[0].nrows, sum(widths))


def hjoin(a, b):
    return hcat([a, b])


def vcat(mats):
    """vcat (Sparse.hs:528-559), assembled on the device: blocks stacked in list order inside every column"""
    mats = list(mats)
    if not mats:
        _oops("vcat", "empty list")
    ncols = mats[0].ncols
    if any(m.ncols != ncols for m in mats):
        _oops("vcat", "ncols mismatch")
    heights = [m.nrows for m in mats]
    return _assemble(mats, np.concatenate([[0], np.cumsum(heights)])[:-1], [0] * len(mats), sum(heights), ncols)


def vjoin(a, b):
    return vcat([a, b])


def fromBlocks(blocks):
    """[[Maybe Matrix]] -> Matrix, None = zero block (Sparse.hs:563-587): vcat . map hcat . adjustDims in one
    device assembly — block (r, c) sits at (sum of the heights above, sum of the widths to the left), listed
    row-major so that blocks sharing columns come by ascending row offset."""
    rows = [list(r) for r in blocks]
    ncb = max(len(r) for r in rows)
    cols = [[r[c] for r in rows if c < len(r)] for c in range(ncb)]

    def spec(groups, attr, what):
        out = []
        for g in groups:
            ds = [getattr(m, attr) for m in g if m is not None]
            if not ds:
                _oops("fromBlocks", "underspecified " + what)
            if any(d != ds[0] for d in ds):
                _oops("fromBlocks", "incompatible " + what)
            out.append(ds[0])
        return out

    heights = spec(rows, "nrows", "heights")
    widths = spec(cols, "ncols", "widths")
    roff = np.concatenate([[0], np.cumsum(heights)])
    # hcat of a block row fails in the reference when the rows' widths differ in total (vcat: ncols mismatch)
    totals = [sum(widths[:len(r)]) for r in rows]
    if any(t != totals[0] for t in totals):
        _oops("vcat", "ncols mismatch")
    coff = np.concatenate([[0], np.cumsum(widths)])
    placed, ro, co = [], [], []
    for r, row in enumerate(rows):
        for c, m in enumerate(row):
            if m is not None:
                placed.append(m)
                ro.append(roff[r])
                co.append(coff[c])
    if not placed:  # every block a zero block cannot happen: heights would be underspecified
        return zeros(int(roff[-1]), totals[0])
    return _assemble(placed, ro, co, int(roff[-1]), totals[0])


def fromBlocksDiag(blocks):
    """blocks given by (super-)diagonals (Sparse.hs:589-597)."""
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
    n = len(mats)
    return fromBlocksDiag([list(mats)] + [[None] * n for _ in range(n - 1)])  # :659-665


def kronecker(matA, matB):
    """Kronecker product (Sparse.hs:597-634), assembled on the device: column ja*ncols(B)+jb holds
    the rows ia*nrows(B)+ib with values b*a."""
    if matA.is_complex or matB.is_complex:
        ar, ai = (matA if matA.is_complex else cmap(lambda v: v.astype(C128), matA))._parts()
        br, bi = (matB if matB.is_complex else cmap(lambda v: v.astype(C128), matB))._parts()
        rr, ii, ri, ir = kronecker(ar, br), kronecker(ai, bi), kronecker(ar, bi), kronecker(ai, br)
        # all four share the pattern of the complex product; (ar + i ai)(br + i bi) componentwise
        return Matrix(rr.ncols, rr.nrows, rr.pointers, rr.indices,
                      (rr.values - ii.values) + 1j * (ri.values + ir.values))
    _ffi.require_gpu()
    a, b = matA._tuple32(), matB._tuple32()
    nr, nc = C.c_int(), C.c_int()
    cp, ci, cx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    st = lib().spl_kronecker(a[0], a[1], p_i32(a[2]), p_i32(a[3]), p_f64(a[4]), b[0], b[1], p_i32(b[2]),
                             p_i32(b[3]), p_f64(b[4]), C.byref(nr), C.byref(nc), C.byref(cp), C.byref(ci),
                             C.byref(cx))
    return _take_matrix("spl_kronecker", st, nr, nc, cp, ci, cx)


def pack(mat):
    """dense copy (Sparse.hs:679-689)"""
    out = np.zeros((mat.nrows, mat.ncols), dtype=mat.values.dtype)
    cols = np.repeat(np.arange(mat.ncols), _lengths(mat))
    out[mat.indices, cols] = mat.values
    return out


def outer(sliceC, sliceR):
    """outer product of a sparse column and a sparse row vector (Sparse.hs:331-355);
    both given as (length, indices, values).  Mirrors the reference's naming,
    in which `sliceC` supplies the column dimension."""
    ncols, indicesC, valuesC = sliceC
    nrows, indicesR, valuesR = sliceR
    lenR = len(valuesR)
    lens = np.zeros(ncols + 1, dtype=I64)
    lens[np.asarray(indicesC, dtype=I64)] = lenR
    ptrs = np.concatenate([[0], np.cumsum(lens)]).astype(I64)
    idx = np.tile(np.asarray(indicesR, dtype=I64), len(valuesC))
    val = (np.asarray(valuesC, dtype=F64)[:, None] * np.asarray(valuesR, dtype=F64)[None, :]).ravel()
    return Matrix(ncols, nrows, ptrs[:ncols + 1], idx, val)
