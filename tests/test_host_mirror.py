"""Host-side mirror (sparse-linear_amd/sparse.py) — logic that needs no GPU."""
import numpy as np
import pytest


def test_ffi_seam_keeps_narrowed_pattern_only_while_it_is_the_same_pattern(pkg):
    """withConstMatrix (Foreign.hs:24-41) narrows on every call; the mirror narrows once per Matrix.  ADVICE r4: the cache
    was keyed on the identity of the arrays alone — an in-place edit kept the identities and the stale int32 copy.  Now the
    fields are read-only views (an edit through the Matrix raises) and every call checks a fingerprint of the pattern, so
    an edit through another alias of the caller's buffer narrows again."""
    p = np.array([0, 1, 2, 3], dtype=np.int64)
    i = np.array([0, 1, 2], dtype=np.int64)
    A = pkg.Matrix(3, 3, p, i, np.ones(3))
    t1, t2 = A._tuple32(), A._tuple32()
    assert t1[2] is t2[2] and t1[3] is t2[3]  # narrowed once
    with pytest.raises(ValueError):
        A.indices[0] = 2
    with pytest.raises(ValueError):
        A.pointers[1] = 0
    i[0] = 1  # the caller's own alias
    t3 = A._tuple32()
    assert t3[3] is not t1[3] and t3[3][0] == 1
    p[3] = 2  # last pointer moved: validated on the current arrays, narrowed again
    t4 = A._tuple32()
    assert t4[2][-1] == 2 and t4[2] is not t3[2]
    # matrices built from the same pattern share its arrays (feast.py: ze*B - A per contour point) and the copies
    B = pkg.Matrix(3, 3, A.pointers, A.indices, 2.0 * np.ones(3))
    assert B.pointers is A.pointers and B.indices is A.indices


def test_ffi_seam_refuses_short_arrays(pkg):
    A = pkg.Matrix(2, 2, np.array([0, 1, 3]), np.array([0, 1]), np.ones(2))
    with pytest.raises(Exception):
        A._tuple32()
