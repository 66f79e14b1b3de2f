"""Mirror of ``Numeric.LinearAlgebra.Umfpack`` (suitesparse/src/Numeric/LinearAlgebra/Umfpack.hs).

Same interface as the reference: the simple ``linearSolve`` / ``solve`` (``<\\>``) and the
advanced ``analyze`` / ``factor`` / ``linearSolve_`` with ``UmfpackNormal`` / ``UmfpackTrans``,
over the same six ``umfpack_di_*`` C symbols (include/umfpack_hip.h) that the reference
imports (Umfpack/Internal.hs:137-148).  Handles are owned like the reference's ForeignPtrs:
the free function is the finalizer and receives a ``void**``.
"""
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from ._ffi import as_f64, lib, p_f64, p_i32

UmfpackNormal = 0  # sys = UMFPACK_A   (Umfpack.hs:95-97)
UmfpackTrans = 1   # sys = UMFPACK_At


class UmfpackError(RuntimeError):
    """negative status: the reference calls errorWithStackTrace (Umfpack.hs:67,81,101)"""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status  # the UMFPACK status behind it (None: raised by a check of this mirror)


def _declare():
    L = lib()
    if getattr(L, "_umf_declared", False):
        return L
    i, vp, ip, dp = C.c_int, C.c_void_p, _ffi.c_int_p, _ffi.c_dbl_p
    L.umfpack_di_symbolic.restype = i
    L.umfpack_di_symbolic.argtypes = [i, i, ip, ip, dp, C.POINTER(vp), dp, dp]
    L.umfpack_di_numeric.restype = i
    L.umfpack_di_numeric.argtypes = [ip, ip, dp, vp, C.POINTER(vp), dp, dp]
    L.umfpack_di_solve.restype = i
    L.umfpack_di_solve.argtypes = [i, ip, ip, dp, dp, dp, vp, dp, dp]
    L.umfpack_di_free_symbolic.restype = None
    L.umfpack_di_free_symbolic.argtypes = [C.POINTER(vp)]
    L.umfpack_di_free_numeric.restype = None
    L.umfpack_di_free_numeric.argtypes = [C.POINTER(vp)]
    L.umfpack_di_report_status.restype = None
    L.umfpack_di_report_status.argtypes = [dp, i]
    # instance Umfpack (Complex Double) (Umfpack/Internal.hs:117-135): packed mode, Az = Xz = Bz = NULL
    L.umfpack_zi_symbolic.restype = i
    L.umfpack_zi_symbolic.argtypes = [i, i, ip, ip, dp, dp, C.POINTER(vp), dp, dp]
    L.umfpack_zi_numeric.restype = i
    L.umfpack_zi_numeric.argtypes = [ip, ip, dp, dp, vp, C.POINTER(vp), dp, dp]
    L.umfpack_zi_solve.restype = i
    L.umfpack_zi_solve.argtypes = [i, ip, ip, dp, dp, dp, dp, dp, dp, vp, dp, dp]
    L.umfpack_zi_free_symbolic.restype = None
    L.umfpack_zi_free_symbolic.argtypes = [C.POINTER(vp)]
    L.umfpack_zi_free_numeric.restype = None
    L.umfpack_zi_free_numeric.argtypes = [C.POINTER(vp)]
    L.umfpack_zi_report_status.restype = None
    L.umfpack_zi_report_status.argtypes = [dp, i]
    L.spl_umfpack_di_solve_many.restype = i
    L.spl_umfpack_di_solve_many.argtypes = [i, ip, ip, dp, i, dp, dp, vp]
    L.spl_umfpack_zi_solve_many.restype = i
    L.spl_umfpack_zi_solve_many.argtypes = [i, ip, ip, dp, dp, i, dp, dp, dp, dp, vp]
    L.spl_umfpack_di_solve_many_dev.restype = i
    L.spl_umfpack_di_solve_many_dev.argtypes = [i, ip, ip, dp, i, vp, vp, vp]
    L.spl_umfpack_zi_solve_many_dev.restype = i
    L.spl_umfpack_zi_solve_many_dev.argtypes = [i, ip, ip, dp, i, vp, vp, vp]
    L.spl_umfpack_path.restype = i
    L.spl_umfpack_path.argtypes = [vp]
    L.spl_umfpack_stats.restype = C.c_int
    L.spl_umfpack_stats.argtypes = [vp, C.POINTER(C.c_double)]
    L.spl_umfpack_solve_report.restype = C.c_int
    L.spl_umfpack_solve_report.argtypes = [vp, C.POINTER(C.c_double)]
    L._umf_declared = True
    return L


def _report(where, status):
    L = _declare()
    L.umfpack_di_report_status(None, status)  # umfpack_report_status mat nullPtr _stat
    if status < 0:
        raise UmfpackError("%s failed (status %d)" % (where, status), status)
    return status


class _Handle(object):
    def __init__(self, value, free_name):
        self._h = C.c_void_p(value)
        self._finalizer = weakref.finalize(self, _Handle._free, self._h, free_name)

    @staticmethod
    def _free(h, free_name):
        try:
            getattr(lib(), free_name)(C.byref(h))
        except Exception:
            pass

    @property
    def value(self):
        return self._h


class Analysis(_Handle):
    """newtype Analysis a = Analysis { fsym :: ForeignPtr (Symbolic a) } (Umfpack.hs:56)"""
    complex = False


class Factors(_Handle):
    """newtype Factors a = Factors { fnum :: ForeignPtr (Numeric a) } (Umfpack.hs:58)"""
    status = 0
    complex = False

    @property
    def path(self):
        """factorisation held now: 0 band with partial pivoting, 1 band without interchanges
        (diagonally dominant), 2 the same as a speculation that every solve checks, 3 / 4
        multifrontal without interchanges (dominant / speculation), 5 multifrontal with static pivoting
        (maximum-product transversal + scalings, csrc/static_pivot.hpp) (include/umfpack_hip.h)"""
        return int(_declare().spl_umfpack_path(self.value))

    @property
    def stats(self):
        """figures of the factorisation held now (spl_umfpack_stats, include/umfpack_hip.h)"""
        buf = (C.c_double * 8)()
        if _declare().spl_umfpack_stats(self.value, buf) != 0:
            raise UmfpackError("spl_umfpack_stats: invalid Numeric object")
        keys = ("path", "n", "kl", "ku", "device_bytes", "flops", "fronts", "flags")
        st = dict(zip(keys, [float(v) if k in ("device_bytes", "flops") else int(v) for k, v in zip(keys, buf)]))
        flags = st.pop("flags")
        st["complex_fronts"] = flags & 1          # native complex fronts (zi objects)
        st["block_pivoting"] = (flags >> 1) & 1   # threshold pivoting inside the diagonal blocks of the fronts
        return st

    @property
    def solve_report(self):
        """the most recent solve on these factors and the bytes one walk over them reads (spl_umfpack_solve_report,
        include/umfpack_hip.h): what UMFPACK returns in Info[UMFPACK_IR_TAKEN / _IR_ATTEMPTED / _OMEGA1]"""
        buf = (C.c_double * 8)()
        if _declare().spl_umfpack_solve_report(self.value, buf) != 0:
            raise UmfpackError("spl_umfpack_solve_report: invalid Numeric object")
        return {"walks": int(buf[0]), "ir_taken": int(buf[1]), "ir_attempted": int(buf[2]),
                "backward_error": float(buf[3]), "walk_bytes": float(buf[4]),
                "chain_bytes": float(buf[5]), "chain_build_ms": float(buf[6]), "chain_span": int(buf[7])}


def analyze(mat):
    """symbolic analysis (Umfpack.hs:60-69)"""
    L = _declare()
    nr, nc, ap, ai, ax = mat._tuple32()
    sym = C.c_void_p()
    if mat.is_complex:
        st = L.umfpack_zi_symbolic(nr, nc, p_i32(ap), p_i32(ai), p_f64(ax), None, C.byref(sym), None, None)
        a = Analysis(sym.value, "umfpack_zi_free_symbolic")
        a.complex = True
        _report("analyze: umfpack_symbolic", st)
        return a
    st = L.umfpack_di_symbolic(nr, nc, p_i32(ap), p_i32(ai), p_f64(ax), C.byref(sym), None, None)
    a = Analysis(sym.value, "umfpack_di_free_symbolic")
    _report("analyze: umfpack_symbolic", st)
    return a


def factor(mat, analysis):
    """numeric factorisation (Umfpack.hs:71-83); needs a GPU"""
    L = _declare()
    _ffi.require_gpu()
    nr, nc, ap, ai, ax = mat._tuple32()
    num = C.c_void_p()
    if mat.is_complex:
        st = L.umfpack_zi_numeric(p_i32(ap), p_i32(ai), p_f64(ax), None, analysis.value, C.byref(num), None, None)
        f = Factors(num.value, "umfpack_zi_free_numeric")
        f.complex = True
        f.status = _report("factor: umfpack_numeric", st)
        return f
    st = L.umfpack_di_numeric(p_i32(ap), p_i32(ai), p_f64(ax), analysis.value, C.byref(num), None, None)
    f = Factors(num.value, "umfpack_di_free_numeric")
    f.status = _report("factor: umfpack_numeric", st)
    return f


def linearSolve_(fact, mode, mat, b):
    """solve with existing factors (Umfpack.hs:87-102); returns the solution vector"""
    L = _declare()
    nr, nc, ap, ai, ax = mat._tuple32()
    # the C side copies nrows (2 nrows) doubles out of b: a short vector would be a heap over-read, and real
    # factors with a complex matrix (or the reverse) would be read with the wrong dimension
    if np.shape(b) != (mat.nrows,):
        raise UmfpackError("linearSolve_: right-hand side has shape %s, the matrix has %d rows" % (np.shape(b), mat.nrows))
    if bool(fact.complex) != bool(mat.is_complex):
        raise UmfpackError("linearSolve_: %s factors used with a %s matrix"
                           % ("complex" if fact.complex else "real", "complex" if mat.is_complex else "real"))
    if mat.is_complex:
        b = np.ascontiguousarray(b, dtype=np.complex128)
        soln = np.zeros(mat.ncols, dtype=np.complex128)
        st = L.umfpack_zi_solve(int(mode), p_i32(ap), p_i32(ai), p_f64(ax), None, p_f64(soln.view(np.float64)), None,
                                p_f64(b.view(np.float64)), None, fact.value, None, None)
        _report("linearSolve_: umfpack_solve", st)
        return soln
    b = as_f64(b)
    soln = np.zeros(mat.ncols, dtype=np.float64)  # MV.replicate ncols 0 (:93)
    st = L.umfpack_di_solve(int(mode), p_i32(ap), p_i32(ai), p_f64(ax), p_f64(soln), p_f64(b), fact.value,
                            None, None)
    _report("linearSolve_: umfpack_solve", st)
    return soln


def linearSolveMany_(fact, mode, mat, bs):
    """`map (linearSolve_ fact mode mat) bs` (Umfpack.hs:103-108) in ONE pass of all right-hand
    sides through the factors (spl_umfpack_{di,zi}_solve_many); returns the list of solutions"""
    L = _declare()
    bs = list(bs)
    k = len(bs)
    if k == 0:
        return []
    nr, nc, ap, ai, ax = mat._tuple32()
    if bool(fact.complex) != bool(mat.is_complex):
        raise UmfpackError("linearSolveMany_: %s factors used with a %s matrix"
                           % ("complex" if fact.complex else "real", "complex" if mat.is_complex else "real"))
    dt = np.complex128 if mat.is_complex else np.float64
    B = np.empty((k, mat.nrows), dtype=dt)  # row c = right-hand side c: column-major n x k for the C side
    for c, b in enumerate(bs):
        b = np.asarray(b)
        if b.shape != (mat.nrows,):
            raise UmfpackError("linearSolveMany_: right-hand side %d has shape %s" % (c, b.shape))
        B[c] = b
    X = np.zeros((k, mat.ncols), dtype=dt)
    if mat.is_complex:
        st = L.spl_umfpack_zi_solve_many(int(mode), p_i32(ap), p_i32(ai), p_f64(ax), None, k,
                                         p_f64(X.view(np.float64)), None, p_f64(B.view(np.float64)), None,
                                         fact.value)
    else:
        st = L.spl_umfpack_di_solve_many(int(mode), p_i32(ap), p_i32(ai), p_f64(ax), k, p_f64(X), p_f64(B),
                                         fact.value)
    _report("linearSolveMany_: umfpack_solve", st)
    return [X[c] for c in range(k)]


def linearSolve(mat, bs):
    """factor once, solve for every right-hand side (Umfpack.hs:38-46)"""
    fact = factor(mat, analyze(mat))
    return linearSolveMany_(fact, UmfpackNormal, mat, bs)


def solve(mat, b):
    """the reference's (<\\>) (Umfpack.hs:48-50)"""
    return linearSolve(mat, [b])[0]


def linearSolveManyDevice_(fact, mode, mat, B):
    """`linearSolveMany_` with the right-hand sides and the solutions in device memory: B is a torch tensor on
    the GPU of shape (k, nrows) — row c = right-hand side c, float64 for real factors, complex128 for complex
    ones — and the result is a new tensor of the same shape and device (spl_umfpack_{di,zi}_solve_many_dev)."""
    import torch
    L = _declare()
    want = torch.complex128 if mat.is_complex else torch.float64
    if bool(fact.complex) != bool(mat.is_complex):
        raise UmfpackError("linearSolveManyDevice_: %s factors used with a %s matrix"
                           % ("complex" if fact.complex else "real", "complex" if mat.is_complex else "real"))
    if not (isinstance(B, torch.Tensor) and B.is_cuda and B.dtype == want and B.dim() == 2 and B.shape[1] == mat.nrows
            and B.is_contiguous()):
        raise UmfpackError("linearSolveManyDevice_: B must be a contiguous %s GPU tensor of shape (k, %d)" % (want, mat.nrows))
    k = int(B.shape[0])
    X = torch.zeros((k, mat.ncols), dtype=want, device=B.device)
    if k == 0:
        return X
    nr, nc, ap, ai, ax = mat._tuple32()
    torch.cuda.current_stream(B.device).synchronize()  # the library works on the null stream of the device
    fn = L.spl_umfpack_zi_solve_many_dev if mat.is_complex else L.spl_umfpack_di_solve_many_dev
    with torch.cuda.device(B.device):
        st = fn(int(mode), p_i32(ap), p_i32(ai), p_f64(ax), k, C.c_void_p(X.data_ptr()), C.c_void_p(B.data_ptr()), fact.value)
    _report("linearSolveManyDevice_: umfpack_solve", st)
    return X

