"""CPU-side check of the drop-in boundary: the shared library builds for gfx950,
loads, and exports every symbol that include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if os.path.basename(h) == "spl_synth.h":  # static inline data spec, exports nothing
            continue
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?(?:int|void|char|double)\s*\*?\s*((?:spl|umfpack)_\w+)\s*\(", text, re.M):
            names.append(m.group(1))
    return names


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._ffi.lib()
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_status_strings(pkg):
    lib = pkg._ffi.lib()
    assert lib.spl_status_string(0) == b"OK"
    assert b"dimension" in lib.spl_status_string(-20)
    assert lib.spl_device_count() >= 0


def test_no_gpu_fails_loudly(pkg):
    """without a GPU the product raises; it never computes on the CPU"""
    import numpy as np
    import pytest
    if pkg._ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.BackendUnavailable):
        pkg.mulV(pkg.ident(3), np.ones(3))
    with pytest.raises(pkg.BackendUnavailable):
        pkg.fromTriples(2, 2, [(0, 0, 1.0)])
    h = ctypes.c_void_p()
    ap = (ctypes.c_int * 2)(0, 0)
    st = pkg._ffi.lib().spl_matrix_create(1, 1, ap, None, None, ctypes.byref(h))
    assert st == pkg._ffi.SPL_ERROR_device and not h.value


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under sparse-linear_amd/ may reference it"""
    pkg_dir = os.path.join(ROOT, "sparse-linear_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libsparse_oracle" not in text, f


def test_product_is_written_from_scratch_no_vendor_primitives():
    """north_star: from scratch, not a vendor library wrapped — no sparse / BLAS / solver / primitive library of the
    platform is included or linked by the product (a library radix sort had crept into the LU analysis)"""
    import re
    pkg_dir = os.path.join(ROOT, "sparse-linear_amd")
    banned = re.compile(r"hipcub|rocprim|rocsparse|hipsparse|rocblas|hipblas|rocsolver|hipsolver|rocthrust|thrust/", re.I)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".hip", ".hpp", ".h", ".cpp", ".py")) or f == "Makefile":
                for no, line in enumerate(open(os.path.join(dirpath, f)), 1):
                    assert not banned.search(line), "%s:%d: %s" % (f, no, line.strip())


def test_marshalling_refuses_arrays_shorter_than_the_pointers_say():
    """the 5-tuple of withConstMatrix carries no lengths (Foreign.hs:24-41): the Python mirror checks them before the
    C side would read pointers[ncols] indices and values out of shorter arrays (no GPU needed: the check is host side)"""
    import numpy as np
    from __graft_entry__ import load_package
    pkg = load_package()
    for M in (pkg.Matrix(3, 3, [0, 2, 3, 9], [0, 2, 1, 2], [1.0, 2.0, 3.0, 4.0]),
              pkg.Matrix(3, 3, [0, 2, 3], [0, 2, 1], [1.0, 2.0, 3.0]),
              pkg.Matrix(3, 3, [0, 2, 3, 4], [0, 2, 1, 2], np.array([1.0, 2.0, 3.0]))):
        try:
            M._tuple32()
        except pkg.SparseError:
            continue
        raise AssertionError("short arrays were marshalled")
    ok = pkg.Matrix(3, 3, [0, 2, 3, 4], [0, 2, 1, 2], [1.0, 2.0, 3.0, 4.0])
    assert ok._tuple32()[0] == 3
