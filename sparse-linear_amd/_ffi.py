"""ctypes binding of the C ABI in include/sparse_linear_hip.h / include/umfpack_hip.h.

The shared library is the product: if it is missing, every operation raises
``BackendUnavailable`` — there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsparse_linear_hip.so")

c_int_p = C.POINTER(C.c_int)
c_i64_p = C.POINTER(C.c_int64)
c_dbl_p = C.POINTER(C.c_double)
c_void_pp = C.POINTER(C.c_void_p)

SPL_OK = 0
SPL_WARNING_singular_matrix = 1
SPL_ERROR_out_of_memory = -1
SPL_ERROR_invalid_handle = -3
SPL_ERROR_argument_missing = -5
SPL_ERROR_n_nonpositive = -6
SPL_ERROR_invalid_matrix = -8
SPL_ERROR_dimension_mismatch = -20
SPL_ERROR_index_out_of_bounds = -21
SPL_ERROR_index_overflow = -22
SPL_ERROR_device = -30
SPL_ERROR_internal = -911


class BackendUnavailable(RuntimeError):
    """The HIP shared library is not built / not loadable."""


class SparseLinearError(RuntimeError):
    """A negative status from the C ABI (the reference's errorWithStackTrace)."""

    def __init__(self, where, status, detail=""):
        self.status = status
        msg = "%s: %s (status %d)" % (where, status_string(status), status)
        if detail:
            msg += " [" + detail + "]"
        super().__init__(msg)


_lib = None


def lib():
    """Load the library once.  torch (if importable) is imported first so that
    both share ONE HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendUnavailable(
            "HIP backend not built: %s missing (run `python -c 'import __graft_entry__ as g; g.build()'`)"
            % LIB_PATH)
    try:
        import torch  # noqa: F401  (shares its libamdhip64 with us)
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    try:
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise BackendUnavailable("cannot load %s: %s" % (LIB_PATH, e))
    _declare(_lib)
    return _lib


def _declare(L):
    i, i64, u64, d = C.c_int, C.c_int64, C.c_uint64, C.c_double
    tup = [i, i, c_int_p, c_int_p, c_dbl_p]
    L.spl_status_string.restype = C.c_char_p
    L.spl_status_string.argtypes = [i]
    L.spl_last_error.restype = C.c_char_p
    L.spl_last_error.argtypes = []
    L.spl_device_count.restype = i
    L.spl_device_count.argtypes = []
    L.spl_release_cached_memory.restype = C.c_ulonglong
    L.spl_release_cached_memory.argtypes = []
    L.spl_device_alloc_seconds.restype = C.c_double
    L.spl_device_alloc_seconds.argtypes = []
    L.spl_free.restype = None
    L.spl_free.argtypes = [C.c_void_p]
    sigs = {
        "spl_gaxpy": tup + [i, c_dbl_p, i, c_dbl_p],
        "spl_mulv": tup + [i, c_dbl_p, c_dbl_p],
        "spl_gaxpy_t": tup + [i, c_dbl_p, i, c_dbl_p],
        "spl_mulm": tup + [i, i, c_dbl_p, c_dbl_p],
        "spl_transpose": tup + [c_int_p, c_int_p, c_dbl_p],
        "spl_spgemm": tup + tup + [c_int_p, c_int_p, c_void_pp, c_void_pp, c_void_pp],
        "spl_spgemm_z": tup + tup + [c_int_p, c_int_p, c_void_pp, c_void_pp, c_void_pp],
        "spl_lin": [d] + tup + [d] + tup + [c_int_p, c_int_p, c_void_pp, c_void_pp, c_void_pp],
        "spl_lin_z": [c_dbl_p] + tup + [c_dbl_p] + tup + [c_int_p, c_int_p, c_void_pp, c_void_pp, c_void_pp],
        "spl_kronecker": tup + tup + [c_int_p, c_int_p, c_void_pp, c_void_pp, c_void_pp],
        "spl_take_diag": tup + [c_dbl_p],
        "spl_compress": [i, i, i64, c_int_p, c_int_p, c_dbl_p, c_int_p, c_void_pp, c_void_pp, c_i64_p],
        "spl_matrix_create": tup + [c_void_pp],
        "spl_matrix_create_rowblock": tup + [i, i, c_void_pp],
        "spl_matrix_create_z": tup + [c_void_pp],
        "spl_matrix_is_complex": [C.c_void_p],
        "spl_matrix_create_csr": [i64, i64, i64, i64, c_int_p, c_int_p, c_dbl_p, c_void_pp],
        "spl_matrix_create_synthetic": [i, i64, i, u64, i64, i64, c_void_pp],
        "spl_matrix_create_rmat": [i, i, d, d, d, u64, c_void_pp],
        "spl_matrix_spgemm": [C.c_void_p, C.c_void_p, c_void_pp, c_i64_p],
        "spl_matrix_lin": [C.c_void_p, c_dbl_p, C.c_void_p, c_dbl_p, c_void_pp],
        "spl_matrix_to_complex": [C.c_void_p, c_void_pp],
        "spl_matrix_transpose": [C.c_void_p, c_void_pp],
        "spl_matrix_compress_dev": [i, i, i64, C.c_void_p, C.c_void_p, C.c_void_p, c_void_pp, c_i64_p],
        "spl_matrix_info": [C.c_void_p, c_i64_p],
        "spl_matrix_export_csr": [C.c_void_p, c_i64_p, c_int_p, c_dbl_p],
        "spl_matrix_export_csc": [C.c_void_p, c_i64_p, c_int_p, c_dbl_p],
        "spl_matrix_export_csr_rows": [C.c_void_p, i64, i64, c_i64_p, i64, c_int_p, c_dbl_p],
        "spl_matrix_mulv": [C.c_void_p, i, c_dbl_p, c_dbl_p],
        "spl_matrix_gaxpy": [C.c_void_p, i, c_dbl_p, i, c_dbl_p],
        "spl_matrix_spmv_dev": [C.c_void_p, C.c_void_p, C.c_void_p, i, C.c_void_p],
        "spl_matrix_spmm_dev": [C.c_void_p, C.c_void_p, C.c_void_p, i, i, C.c_void_p],
        "spl_matrix_set_variant": [C.c_void_p, i],
        "spl_matrix_optimize": [C.c_void_p],
        "spl_matrix_build_blocked": [C.c_void_p, i, i, i],
        "spl_matrix_build_panel": [C.c_void_p, i, i, i, i],
        "spl_matrix_panel_errors": [C.c_void_p],
        "spl_matrix_set_spmv_order": [C.c_void_p, i],
        "spl_matrix_set_reserved_cus": [C.c_void_p, i],
        "spl_debug_occupy": [i, i, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p],
        "spl_debug_sort_u64": [C.c_void_p, C.c_longlong, i, C.c_void_p],
        "spl_matrix_spmv_kernel": [C.c_void_p],
        "spl_vector_synthetic_dev": [u64, i64, i64, C.c_void_p, C.c_void_p],
        "spl_peer_exchange_create": [i, i, i, i64, c_i64_p, C.c_char_p, c_void_pp],
        "spl_peer_exchange_connect": [C.c_void_p, C.c_char_p],
        "spl_peer_exchange_push": [C.c_void_p, i, C.c_void_p, C.c_void_p],
        "spl_peer_exchange_finish": [C.c_void_p, C.c_void_p, c_void_pp],
        "spl_peer_exchange_failed": [C.c_void_p],
        "spl_peer_exchange_flags_finegrained": [C.c_void_p],
    }
    for name, args in sigs.items():
        fn = getattr(L, name, None)
        if fn is None:
            continue
        fn.restype = i
        fn.argtypes = args
    L.spl_matrix_free.restype = None
    L.spl_matrix_free.argtypes = [c_void_pp]
    if hasattr(L, "spl_peer_exchange_free"):
        L.spl_peer_exchange_free.restype = None
        L.spl_peer_exchange_free.argtypes = [c_void_pp]


def status_string(status):
    return lib().spl_status_string(int(status)).decode()


def check(where, status):
    """Negative status -> exception (Umfpack.hs:67,81,101); positive = warning, returned."""
    if status < 0:
        detail = lib().spl_last_error().decode() if status == SPL_ERROR_device else ""
        raise SparseLinearError(where, status, detail)
    return status


def device_count():
    return int(lib().spl_device_count())


def device_alloc_seconds():
    """seconds this process has spent inside hipMalloc for the library so far (spl_device_alloc_seconds)"""
    return float(lib().spl_device_alloc_seconds())


def release_cached_memory():
    """Gives the device blocks the library keeps for reuse back to the driver; returns the bytes released."""
    return int(lib().spl_release_cached_memory())


def require_gpu():
    if device_count() < 1:
        raise BackendUnavailable("no HIP device visible to libsparse_linear_hip.so")


# -- numpy <-> pointer helpers -----------------------------------------------------------

def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def p_i32(a):
    return a.ctypes.data_as(c_int_p)


def p_i64(a):
    return a.ctypes.data_as(c_i64_p)


def p_f64(a):
    return a.ctypes.data_as(c_dbl_p)


def take_malloced(ptr, n, ctype, dtype):
    """copy n items out of a malloc()'d array returned by the ABI and free it
    (what `fromForeign True` followed by free would do, Foreign.hs:47-55)"""
    n = int(n)
    if n > 0:
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)
    else:
        arr = np.zeros(0, dtype=dtype)
    lib().spl_free(ptr)
    return arr
