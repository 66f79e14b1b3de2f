"""Complex Double (the reference's second SPECIALIZE instance): the complex fixtures of
sparse-linear/tests/Sparse.hs:61-73 and the reference's only UMFPACK test
(suitesparse/tests/test-umfpack.hs:16-19, `ident <\\> v == v` on Vector (Complex Double)).
Since round 2 mulV / axpy_, lin and mm run on native packed-complex kernels (bit-identical to the oracle's
complex restatements); the LU factors the real embedding (csrc/umfpack_zi.hip)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import mat_to_tuple, tuple_to_mat, tuples_equal

pytestmark = pytest.mark.gpu

cval = st.builds(complex, st.integers(-20, 20), st.integers(-20, 20))


def test_ctrans_fixtures(gpu, pkg):
    m = pkg.fromTriples(2, 2, [(0, 0, 2 + 0j), (0, 1, -1 + 0j), (1, 0, -1 + 0j), (1, 1, 2 + 0j)])
    assert m == pkg.ctrans(m)                                   # "preserves hermitian matrices"
    sx = pkg.fromTriples(2, 2, [(0, 1, 1 + 0j), (1, 0, 1 + 0j)])
    assert sx == pkg.ctrans(sx)                                 # "preserves sigma_x"
    sy = pkg.fromTriples(2, 2, [(0, 1, -1j), (1, 0, 1j)])
    assert sy == pkg.ctrans(sy) and pkg.hermitian(sy)           # "preserves sigma_y"
    assert not (pkg.transpose(sy) == sy)


@settings(max_examples=20, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.lists(st.builds(complex, st.floats(-1e6, 1e6, allow_nan=False), st.floats(-1e6, 1e6, allow_nan=False)),
                min_size=1, max_size=40))
def test_ident_solve_complex_exact(gpu, pkg, v):
    v = np.array(v, dtype=np.complex128)
    A = pkg.diag(np.ones(len(v), dtype=np.complex128))
    assert np.array_equal(pkg.umfpack.solve(A, v), v)           # prop_linSolveId


def test_complex_arithmetic_vs_numpy(gpu, pkg):
    rng = np.random.default_rng(8)
    nr, nc, k = 30, 25, 200
    tri = lambda n1, n2: [(int(rng.integers(0, n1)), int(rng.integers(0, n2)),
                           complex(rng.integers(-5, 6), rng.integers(-5, 6))) for _ in range(k)]
    A, B = pkg.fromTriples(nr, nc, tri(nr, nc)), pkg.fromTriples(nc, 17, tri(nc, 17))
    A2 = pkg.fromTriples(nr, nc, tri(nr, nc))
    x = rng.integers(-4, 5, nc) + 1j * rng.integers(-4, 5, nc)
    assert np.array_equal(pkg.mulV(A, x), pkg.pack(A) @ x)      # integer-valued: exact
    assert np.array_equal(pkg.pack(A * B), pkg.pack(A) @ pkg.pack(B))
    assert np.array_equal(pkg.pack(A + A2), pkg.pack(A) + pkg.pack(A2))
    assert A - A == pkg.cmap(lambda v: v * 0, A)
    assert np.array_equal(pkg.pack(pkg.ctrans(A)), pkg.pack(A).conj().T)
    assert (A * B).is_complex


def test_complex_solve_and_conjugate_transpose(gpu, pkg, O):
    rng = np.random.default_rng(9)
    n = 120
    rows = list(rng.integers(0, n, 700)) + list(range(n))
    cols = list(rng.integers(0, n, 700)) + list(range(n))
    vals = list(rng.normal(size=700) + 1j * rng.normal(size=700)) + [complex(25.0, 3.0)] * n
    A = pkg.fromTriples(n, n, list(zip(map(int, rows), map(int, cols), vals)))
    D = pkg.pack(A)
    xs = rng.normal(size=n) + 1j * rng.normal(size=n)
    fact = pkg.umfpack.factor(A, pkg.umfpack.analyze(A))
    x = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackNormal, A, D @ xs)
    assert np.max(np.abs(x - xs)) < 1e-10
    xh = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackTrans, A, D.conj().T @ xs)  # UMFPACK_At = A^H
    assert np.max(np.abs(xh - xs)) < 1e-10
    # the FEAST-style shifted matrix ze*I - A of feast/tests/test-feast.hs:25 at a complex contour point
    F = pkg.fromTriples(2, 2, [(0, 0, 2 + 0j), (0, 1, -1 + 0j), (1, 0, -1 + 0j), (1, 1, 2 + 0j)])
    ze = 2.0 + 1.5j
    S = pkg.lin(-1.0, F, 1.0, pkg.diag(np.full(2, ze)))
    b = np.array([1.0 + 0j, 0.0])
    y = pkg.umfpack.solve(S, b)
    assert np.max(np.abs(pkg.pack(S) @ y - b)) < 1e-13


def test_complex_batched_solve(gpu, pkg, O):
    """spl_umfpack_zi_solve_many: packed complex right-hand sides, both systems"""
    import scipy.sparse as sp
    m = 14
    n = m * m
    rp, ci, v = O.gen_poisson2d_csr(m)
    S = ((2.0 + 0.7j) * sp.identity(n) - sp.csc_matrix((v, ci, rp), shape=(n, n))).tocsc()
    S.sort_indices()
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    rng = np.random.default_rng(3)
    xs = [rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(-1.0, 1.0, n) for _ in range(5)]
    U = pkg.umfpack
    fact = U.factor(M, U.analyze(M))
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, S.conj().T.tocsc())):
        got = U.linearSolveMany_(fact, mode, M, [op @ x for x in xs])
        for x, g in zip(xs, got):
            assert np.max(np.abs(g - x)) / np.max(np.abs(x)) < 1e-10


def test_complex_diagonal_with_dominant_imaginary_parts(gpu, pkg, O):
    """rows whose diagonal entry has the larger imaginary part swap their two real equations in the
    embedding (static pivoting in the zi wrapper): mixed rows, both systems, one and several
    right-hand sides; then the shift z I - A with Re z on A's diagonal (real pivot exactly 0)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    n = 150
    rows = list(rng.integers(0, n, 600)) + list(range(n))
    cols = list(rng.integers(0, n, 600)) + list(range(n))
    diag = [complex(0.0, 30.0) if k % 3 == 0 else complex(28.0, 2.0) if k % 3 == 1 else complex(-1.0, -26.0)
            for k in range(n)]
    vals = list(rng.normal(size=600) + 1j * rng.normal(size=600)) + diag
    A = pkg.fromTriples(n, n, list(zip(map(int, rows), map(int, cols), vals)))
    D = pkg.pack(A)
    U = pkg.umfpack
    fact = U.factor(A, U.analyze(A))
    xs = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(9)]
    for mode, op in ((U.UmfpackNormal, D), (U.UmfpackTrans, D.conj().T)):
        x = U.linearSolve_(fact, mode, A, op @ xs[0])
        assert np.max(np.abs(x - xs[0])) < 1e-10
        for want, got in zip(xs, U.linearSolveMany_(fact, mode, A, [op @ x for x in xs])):
            assert np.max(np.abs(got - want)) < 1e-10
    m = 40
    rp, ci, v = O.gen_poisson2d_csr(m)
    S = ((4.0 + 0.01j) * sp.identity(m * m) - sp.csc_matrix((v, ci, rp), shape=(m * m, m * m))).tocsc()
    S.sort_indices()
    M = pkg.Matrix(m * m, m * m, S.indptr, S.indices, S.data)
    fact = U.factor(M, U.analyze(M))
    assert fact.path in (2, 4)  # not thrown back to partial pivoting by a zero real pivot
    want = rng.uniform(0.5, 1.5, m * m) + 1j * rng.uniform(0.5, 1.5, m * m)
    got = U.linearSolve_(fact, U.UmfpackNormal, M, S @ want)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-9
    assert fact.path in (2, 4)


def test_zi_split_arrays_through_the_c_abi(gpu, pkg):
    """umfpack_zi_* with separate real and imaginary arrays (Az, Xz, Bz non-NULL) — the form the
    reference does not use but UMFPACK's ABI defines —, on a matrix whose diagonal mixes dominant real
    and dominant imaginary parts; both systems, one and several right-hand sides"""
    import ctypes as C
    import scipy.sparse as sp
    L = pkg._ffi.lib()
    pkg.umfpack._declare()
    rng = np.random.default_rng(5)
    n = 90
    S = (sp.random(n, n, density=0.05, random_state=3) + 1j * sp.random(n, n, density=0.05, random_state=4)).tocsc()
    w = np.asarray(abs(S).sum(axis=0)).ravel() + 1.0
    S = sp.csc_matrix(S + sp.diags(np.where(np.arange(n) % 2 == 0, w, -1j * w)))
    S.sort_indices()
    ap, ai = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    ax, az = np.ascontiguousarray(S.data.real), np.ascontiguousarray(S.data.imag)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    P = lambda a, t: a.ctypes.data_as(t)
    sym, num = C.c_void_p(), C.c_void_p()
    assert L.umfpack_zi_symbolic(n, n, P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), C.byref(sym), None, None) == 0
    assert L.umfpack_zi_numeric(P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), sym, C.byref(num), None, None) == 0
    D = S.toarray()
    for sys_, op in ((0, D), (1, D.conj().T)):
        xs = rng.normal(size=n) + 1j * rng.normal(size=n)
        b = op @ xs
        bx, bz = np.ascontiguousarray(b.real), np.ascontiguousarray(b.imag)
        xx, xz = np.zeros(n), np.zeros(n)
        assert L.umfpack_zi_solve(sys_, P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), P(xx, dp), P(xz, dp), P(bx, dp),
                                  P(bz, dp), num, None, None) == 0
        assert np.max(np.abs(xx + 1j * xz - xs)) < 1e-10
        k = 3
        Xs = rng.normal(size=(k, n)) + 1j * rng.normal(size=(k, n))
        Bm = np.stack([op @ Xs[c] for c in range(k)])
        bx, bz = np.ascontiguousarray(Bm.real), np.ascontiguousarray(Bm.imag)  # n x k column-major = k rows of n
        xx, xz = np.zeros((k, n)), np.zeros((k, n))
        assert L.spl_umfpack_zi_solve_many(sys_, P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), k, P(xx, dp), P(xz, dp),
                                           P(bx, dp), P(bz, dp), num) == 0
        assert np.max(np.abs(xx + 1j * xz - Xs)) < 1e-10
    L.umfpack_zi_free_numeric(C.byref(num))
    L.umfpack_zi_free_symbolic(C.byref(sym))


# ---- complex symmetric matrices: symmetric real embedding, L D L^T on the tree (csrc/umfpack_zi.hip, head) ----------
def _bwd(S, x, b):
    r = np.abs(S @ x - b)
    den = abs(S) @ np.abs(x) + np.abs(b)
    return float(np.max(r / np.where(den > 0, den, 1.0)))


@pytest.mark.parametrize("limits", ["default", "small"])
@pytest.mark.parametrize("dim,m", [(2, 60), (3, 16)])
@pytest.mark.parametrize("z", [4.0 + 0.01j, 0.7 - 0.4j, -0.2 + 3.0j, 6.0 + 0.0j])
def test_complex_symmetric_shift_native_and_embedded(gpu, pkg, dim, m, z, limits, monkeypatch):
    """FEAST's z B - A for real symmetric A, B (complex symmetric, not Hermitian), three ways: (1) native complex fronts
    in their L D L^T mode (the default once the tree is chosen: a quarter of the flops of (3)), (2) SPL_ZI_NATIVE=0,
    SPL_ZI_SYMMETRIC=1: the symmetric real embedding of D A D (|u_r| = 1, real positive diagonal) on real fronts in
    L D L^T mode (half), (3) the general real embedding.  The same solutions of A x = b and A^H y = c to 1e-10, backward
    errors at rounding level; packed, batched and device right-hand sides.  z = 4 + 0.01i puts Re z on A's diagonal (the
    real pivot of the plain embedding is 0), z = 6 is a real shift.  "small": size classes lowered so that the
    one-workgroup, lockstep and multi-launch kernels of the factorisation and the many-workgroup solves all run."""
    import scipy.sparse as sp
    import torch
    if limits == "small":
        monkeypatch.setenv("SPL_MF_SMALL", "64")
        monkeypatch.setenv("SPL_MF_MIDMAX", "256")
        monkeypatch.setenv("SPL_MF_BIGSOLVE", "64")
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    rng = np.random.default_rng(m)
    T = sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    I = sp.identity(m)
    A = (sp.kron(I, T) + sp.kron(T, I)) if dim == 2 else (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I))
    n = A.shape[0]
    B = sp.diags(rng.uniform(0.8, 1.25, n))
    S = sp.csc_matrix(z * B - A)
    S.sort_indices()
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    U = pkg.umfpack
    monkeypatch.setenv("SPL_ZI_NATIVE", "1")  # by itself the wrapper goes native from 1e12 flops of the embedding's tree on
    fn = U.factor(M, U.analyze(M))
    monkeypatch.setenv("SPL_ZI_NATIVE", "0")
    monkeypatch.setenv("SPL_ZI_SYMMETRIC", "1")
    fs = U.factor(M, U.analyze(M))
    monkeypatch.setenv("SPL_ZI_SYMMETRIC", "0")
    fg = U.factor(M, U.analyze(M))
    monkeypatch.delenv("SPL_ZI_SYMMETRIC")
    monkeypatch.delenv("SPL_ZI_NATIVE")
    fd = U.factor(M, U.analyze(M))
    assert fd.stats["flops"] == fg.stats["flops"] < 1e12 and fd.stats["complex_fronts"] == 0  # small trees: general embedding
    assert fn.path in (3, 4) and fs.path in (3, 4) and fg.path in (3, 4)
    sn, ss, sg = fn.stats, fs.stats, fg.stats
    assert (sn["complex_fronts"], ss["complex_fronts"], sg["complex_fronts"]) == (1, 0, 0)
    assert ss["flops"] == 0.5 * sg["flops"] and ss["fronts"] == sg["fronts"] == sn["fronts"]
    assert abs(sn["flops"] - 0.25 * sg["flops"]) <= 1e-9 * sg["flops"] and sn["device_bytes"] < sg["device_bytes"]
    xs = [rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(-1, 1, n) for _ in range(5)]
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.conj().T))):
        bs = [np.asarray(op @ x).ravel() for x in xs]
        ref = U.linearSolve_(fg, mode, M, bs[0])
        for f in (fn, fs):
            one = U.linearSolve_(f, mode, M, bs[0])
            assert _bwd(op, one, bs[0]) <= 1e-13
            assert np.max(np.abs(one - ref)) <= 1e-10 * np.max(np.abs(ref))
            for got, b in zip(U.linearSolveMany_(f, mode, M, bs), bs):
                assert _bwd(op, got, b) <= 1e-13
            Xd = U.linearSolveManyDevice_(f, mode, M, torch.from_numpy(np.stack(bs)).cuda()).cpu().numpy()
            for got, b in zip(Xd, bs):
                assert _bwd(op, got, b) <= 1e-13
    assert fn.stats["complex_fronts"] == 1  # no fallback happened on the way


@pytest.mark.parametrize("limits", ["default", "small"])
@pytest.mark.parametrize("kind", ["dominant", "general", "hermitian"])
def test_native_complex_fronts_unsymmetric(gpu, pkg, kind, limits, monkeypatch):
    """complex matrices without symmetry on a mesh pattern through the native complex fronts (plain complex LU on the
    tree): a column-dominant one (path 3), one with a weak diagonal (a speculation, path 4 — or whatever the fallbacks
    end on: the answer is what is checked) and a Hermitian one; both systems, one and several right-hand sides, against
    the real fronts of the embedding (SPL_ZI_NATIVE=0)"""
    import scipy.sparse as sp
    if limits == "small":
        monkeypatch.setenv("SPL_MF_SMALL", "64")
        monkeypatch.setenv("SPL_MF_MIDMAX", "256")
        monkeypatch.setenv("SPL_MF_BIGSOLVE", "64")
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    rng = np.random.default_rng(17)
    m = 48
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    P = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)), dtype=np.complex128)
    n = m * m
    P.data = rng.uniform(-1, 1, P.nnz) + 1j * rng.uniform(-1, 1, P.nnz)
    if kind == "hermitian":
        P = sp.csc_matrix(sp.triu(P, 1) + sp.triu(P, 1).conj().T + sp.diags(rng.uniform(6.0, 7.0, n)))
    else:
        w = np.asarray(abs(P).sum(axis=0)).ravel()
        ph = np.exp(1j * rng.uniform(0, 2 * np.pi, n))
        # dominance is decided on the real embedding: |Re a_jj| against the 1-norms of the other entries
        P.setdiag(2.0 * w + 0j if kind == "dominant" else 0.6 * w * ph)
    S = sp.csc_matrix(P)
    S.sort_indices()
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    U = pkg.umfpack
    monkeypatch.setenv("SPL_ZI_NATIVE", "1")
    fn = U.factor(M, U.analyze(M))
    assert fn.stats["complex_fronts"] == 1 and fn.path in {"dominant": (3,), "general": (4,), "hermitian": (3, 4)}[kind]
    monkeypatch.setenv("SPL_ZI_NATIVE", "0")
    fe = U.factor(M, U.analyze(M))
    assert fe.stats["complex_fronts"] == 0 and abs(fn.stats["flops"] - 0.5 * fe.stats["flops"]) <= 1e-9 * fe.stats["flops"]
    xs = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(9)]
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.conj().T))):
        bs = [np.asarray(op @ x).ravel() for x in xs]
        one, ref = U.linearSolve_(fn, mode, M, bs[0]), U.linearSolve_(fe, mode, M, bs[0])
        assert _bwd(op, one, bs[0]) <= 1e-13
        assert np.max(np.abs(one - ref)) <= 1e-9 * np.max(np.abs(ref))
        for got, b in zip(U.linearSolveMany_(fn, mode, M, bs), bs):
            assert _bwd(op, got, b) <= 1e-13


@pytest.mark.parametrize("complex_values", [False, True])
def test_batched_solves_with_boundary_products_in_several_chunks(gpu, pkg, complex_values, monkeypatch):
    """a 3-D mesh large enough that fronts below the root have boundaries of more than 512 indices: the boundary product
    of the back substitution then runs in several chunks per block of rows, partial sums through scratch and the
    reduction kernel (multifrontal.hip: big_gemv_chunk_kernel, big_gemv_chunk_t_kernel, big_gemv_reduce_kernel); 9
    right-hand sides (a full group of eight and a group of one), both systems, unsymmetric values so that the transposed
    kernels run; real fronts and native complex fronts"""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_ZI_NATIVE", "1")
    rng = np.random.default_rng(23)
    m = 34
    n = m ** 3
    T = sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    I = sp.identity(m)
    K = sp.csc_matrix(sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I))
    K.sort_indices()
    vals = K.data * rng.uniform(0.8, 1.2, K.nnz)  # same pattern, no symmetry, rows stay dominant enough
    if complex_values:
        vals = vals * np.exp(0.3j * rng.uniform(-1, 1, K.nnz))
    S = sp.csc_matrix((vals, K.indices, K.indptr), shape=(n, n)) + 0.5 * sp.identity(n)
    S = sp.csc_matrix(S)
    S.sort_indices()
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    U = pkg.umfpack
    f = U.factor(M, U.analyze(M))
    st = f.stats
    assert st["path"] in (3, 4) and st["fronts"] > 100 and st["complex_fronts"] == (1 if complex_values else 0)
    xs = [rng.normal(size=n) + (1j * rng.normal(size=n) if complex_values else 0.0) for _ in range(9)]
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.conj().T))):
        bs = [np.asarray(op @ x).ravel() for x in xs]
        got = U.linearSolveMany_(f, mode, M, bs)
        for x, g, b in zip(xs, got, bs):
            assert _bwd(op, g, b) <= 1e-13
            assert np.max(np.abs(g - x)) <= 1e-9 * np.max(np.abs(x))
        one = U.linearSolve_(f, mode, M, bs[3])
        assert np.max(np.abs(one - got[3])) <= 1e-10 * np.max(np.abs(one))


@pytest.mark.parametrize("native", ["0", "1"])
def test_complex_symmetric_split_arrays_and_hermitian_stays_general(gpu, pkg, native, monkeypatch):
    """a complex symmetric matrix through umfpack_zi_* with split real / imaginary arrays (Az, Xz, Bz non-NULL), on the
    symmetric embedding (native = 0) and on native complex fronts; and a complex HERMITIAN matrix (A == A^H, not A^T)
    is factored as plain LU either way (all the flops of its tree)"""
    import ctypes as C
    import scipy.sparse as sp
    L = pkg._ffi.lib()
    pkg.umfpack._declare()
    rng = np.random.default_rng(9)
    m = 30
    T = sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    K = sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))
    n = m * m
    W = sp.triu(sp.random(n, n, density=3.0 / n, random_state=2), 1) * (0.2 + 0.3j)
    S = sp.csc_matrix((1.3 + 0.6j) * sp.identity(n) - K + W + W.T)  # complex symmetric
    S.sort_indices()
    assert abs(S - S.T).max() == 0 and abs(S - S.conj().T).max() > 0
    ap, ai = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    ax, az = np.ascontiguousarray(S.data.real), np.ascontiguousarray(S.data.imag)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    P = lambda a, t: a.ctypes.data_as(t)
    sym, num = C.c_void_p(), C.c_void_p()
    monkeypatch.setenv("SPL_ZI_SYMMETRIC", "1")
    monkeypatch.setenv("SPL_ZI_NATIVE", native)
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    assert L.umfpack_zi_symbolic(n, n, P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), C.byref(sym), None, None) == 0
    assert L.umfpack_zi_numeric(P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), sym, C.byref(num), None, None) == 0
    for sys_, op in ((0, S), (1, sp.csc_matrix(S.conj().T))):
        xs = rng.normal(size=n) + 1j * rng.normal(size=n)
        b = np.asarray(op @ xs).ravel()
        bx, bz = np.ascontiguousarray(b.real), np.ascontiguousarray(b.imag)
        xx, xz = np.zeros(n), np.zeros(n)
        assert L.umfpack_zi_solve(sys_, P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), P(xx, dp), P(xz, dp), P(bx, dp),
                                  P(bz, dp), num, None, None) == 0
        assert _bwd(op, xx + 1j * xz, b) <= 1e-13
        k = 3
        Xs = rng.normal(size=(k, n)) + 1j * rng.normal(size=(k, n))
        Bm = np.stack([np.asarray(op @ Xs[c]).ravel() for c in range(k)])
        bx, bz = np.ascontiguousarray(Bm.real), np.ascontiguousarray(Bm.imag)
        xx, xz = np.zeros((k, n)), np.zeros((k, n))
        assert L.spl_umfpack_zi_solve_many(sys_, P(ap, ip), P(ai, ip), P(ax, dp), P(az, dp), k, P(xx, dp), P(xz, dp),
                                           P(bx, dp), P(bz, dp), num) == 0
        for c in range(k):
            assert _bwd(op, xx[c] + 1j * xz[c], Bm[c]) <= 1e-13
    L.umfpack_zi_free_numeric(C.byref(num))
    L.umfpack_zi_free_symbolic(C.byref(sym))
    # Hermitian: general embedding
    H = sp.csc_matrix(4.5 * sp.identity(n) - K + W + W.conj().T)
    H.sort_indices()
    M = pkg.Matrix(n, n, H.indptr, H.indices, H.data)
    U = pkg.umfpack
    fh = U.factor(M, U.analyze(M))
    monkeypatch.setenv("SPL_ZI_SYMMETRIC", "0")
    fg = U.factor(M, U.analyze(M))
    assert fh.stats["flops"] == fg.stats["flops"] and fh.stats["complex_fronts"] == int(native)
    xs = rng.normal(size=n) + 1j * rng.normal(size=n)
    b = np.asarray(H @ xs).ravel()
    assert _bwd(H, U.linearSolve_(fh, U.UmfpackNormal, M, b), b) <= 1e-13


# ---- native Complex Double SpMV (csrc/spmv_z.hip) against the oracle's complex restatement of axpy_ -------------
def _rand_complex(O, rng, nr, nc, k, ints=False):
    A = O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), rng.normal(size=k))
    if ints:
        vals = rng.integers(-9, 10, len(A[4])) + 1j * rng.integers(-9, 10, len(A[4]))
    else:
        vals = A[4] + 1j * rng.normal(size=len(A[4]))
    return (nr, nc, A[2], A[3], vals.astype(np.complex128))


@pytest.mark.parametrize("nr,nc,k", [(1, 1, 1), (7, 5, 20), (300, 200, 5000), (5000, 7000, 90000), (64, 64, 4096), (20000, 20000, 400000)])
def test_native_complex_mulv_matches_oracle_bitwise(gpu, pkg, O, nr, nc, k):
    """mulV / axpy_ / axpy on Complex Double: y <- a * x + y per stored entry in ascending column order with
    Data.Complex's (a :+ b) * (c :+ d) = (a*c - b*d) :+ (a*d + b*c) — bit-identical to the oracle's restatement
    (oracle/sparse_oracle.c orc_axpy_z, Sparse.hs:433-453 under the SPECIALIZE of :456-457)"""
    rng = np.random.default_rng(nr + nc + k)
    A = _rand_complex(O, rng, nr, nc, k)
    M = pkg.Matrix(nc, nr, A[2], A[3], A[4])
    assert M.device_handle().is_complex
    x = rng.normal(size=nc) + 1j * rng.normal(size=nc)
    y = pkg.mulV(M, x)
    assert y.dtype == np.complex128 and np.array_equal(y, O.mulV_z(A, x))
    y0 = rng.normal(size=nr) + 1j * rng.normal(size=nr)
    ya = pkg.axpy(M, x, y0)
    yo = y0.copy()
    O.axpy_z(A, x, yo)
    assert np.array_equal(ya, yo)
    yin = y0.copy()
    pkg.axpy_(M, x, yin)
    assert np.array_equal(yin, yo)
    with pytest.raises(pkg.SparseError):
        pkg.mulV(M, np.ones(nc + 1, dtype=complex))


def test_native_complex_fixtures_and_long_rows(gpu, pkg, O):
    # sigma_y (Sparse.hs:70-72): [[0, -i], [i, 0]] (1, i) = (1, i)
    sy = pkg.Matrix(2, 2, [0, 1, 2], [1, 0], np.array([1j, -1j]))
    assert np.array_equal(pkg.mulV(sy, np.array([1.0, 1j])), np.array([1.0 + 0j, 1j]))
    # a real matrix applied to a complex vector == componentwise real products
    rng = np.random.default_rng(8)
    n = 3000
    R = O.compress(n, n, rng.integers(0, n, 40000), rng.integers(0, n, 40000), rng.integers(-5, 6, 40000).astype(float))
    Mr = pkg.Matrix(n, n, R[2], R[3], R[4])
    xr, xi = rng.integers(-5, 6, n).astype(float), rng.integers(-5, 6, n).astype(float)
    y = pkg.mulV(Mr, xr + 1j * xi)
    assert np.array_equal(y.real, O.mulV(R, xr)) and np.array_equal(y.imag, O.mulV(R, xi))
    # one row far longer than an LDS chunk: wavefront tree sum, the reference's closeness predicate
    k = 2500
    cols = rng.choice(n, k, replace=False)
    A = O.compress(3, n, np.concatenate([np.zeros(k, dtype=int), [1, 2]]), np.concatenate([cols, [5, 7]]), rng.normal(size=k + 2))
    Az = (3, n, A[2], A[3], (A[4] + 1j * rng.normal(size=len(A[4]))).astype(np.complex128))
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    y = pkg.mulV(pkg.Matrix(n, 3, Az[2], Az[3], Az[4]), x)
    yo = O.mulV_z(Az, x)
    assert np.max(np.abs(y - yo)) <= 1e-10 * np.max(np.abs(yo)) and np.array_equal(y[1:], yo[1:])


def test_native_complex_lin_matches_oracle_bitwise(gpu, pkg, O):
    """lin with complex scalars (what Feast.hs:216 calls: `lin (-1) matA _ze matB`) on the packed-complex kernel
    (spl_lin_z) against the oracle's restatement of glin at Complex Double: structure and values bit for bit —
    columns present in one operand only, overlapping entries, exact cancellation (stored 0), real operands
    promoted, the `+` / `-` of the Num instance"""
    rng = np.random.default_rng(23)
    for nr, nc, k in ((1, 1, 1), (7, 5, 12), (60, 40, 300), (300, 500, 4000)):
        A, B = _rand_complex(O, rng, nr, nc, k), _rand_complex(O, rng, nr, nc, k)
        # empty columns on either side
        for M in (A, B):
            c = int(rng.integers(0, nc))
            lo, hi = int(M[2][c]), int(M[2][c + 1])
            M[2][c + 1:] -= hi - lo
            keep = np.r_[0:lo, hi:len(M[3])]
            A_or_B = (M[0], M[1], M[2], M[3][keep], M[4][keep])
            if M is A:
                A = A_or_B
            else:
                B = A_or_B
        for alpha, beta in ((-1.0, 0.3 + 0.7j), (2.5 - 1j, -0.5j), (1.0, 1.0), (1.0, -1.0)):
            got = pkg.lin(alpha, tuple_to_mat(pkg, A), beta, tuple_to_mat(pkg, B))
            ref = O.lin_z(alpha, A, beta, B)
            assert got.is_complex and tuples_equal(mat_to_tuple(got), ref)
        # cancellation keeps the entry
        Z = pkg.lin(1.0 + 0j, tuple_to_mat(pkg, A), -1.0 + 0j, tuple_to_mat(pkg, A))
        assert np.array_equal(Z.indices, A[3]) and not np.any(Z.values)
        # Num instance
        Am, Bm = tuple_to_mat(pkg, A), tuple_to_mat(pkg, B)
        assert tuples_equal(mat_to_tuple(Am + Bm), O.lin_z(1.0, A, 1.0, B))
        assert tuples_equal(mat_to_tuple(Am - Bm), O.lin_z(1.0, A, -1.0, B))
    # a real matrix with a complex scalar is promoted; unsorted input columns are tolerated
    R = pkg.Matrix(2, 3, [0, 2, 3], [2, 0, 1], [1.0, 2.0, 3.0])
    Rs = (3, 2, np.array([0, 2, 3]), np.array([0, 2, 1]), np.array([2.0, 1.0, 3.0]))
    got = pkg.lin(1j, R, 2.0, R)
    assert tuples_equal(mat_to_tuple(got), O.lin_z(1j, Rs, 2.0, Rs))
    with pytest.raises(Exception):
        pkg.lin(1j, R, 1.0, pkg.transpose(R))


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (40, 30, 50, 300), (300, 300, 300, 3000), (2000, 1500, 1800, 30000),
                                   (200, 5000, 100, 8000)])
def test_native_complex_mm_matches_oracle_bitwise(gpu, pkg, O, shape):
    """mm on Complex Double (spl_spgemm_z): pattern of the real product, every value accumulated over ascending k
    with Data.Complex's arithmetic — structure and values bit-identical to the oracle's restatement (orc_mm_z,
    Sparse.hs:691-702)"""
    m, n, p, k = shape
    rng = np.random.default_rng(sum(shape))
    A, B = _rand_complex(O, rng, m, n, k), _rand_complex(O, rng, n, p, k)
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    assert C.is_complex and tuples_equal(mat_to_tuple(C), O.mm_z(A, B))
    assert O.check_matrix((C.nrows, C.ncols, C.pointers, C.indices, C.values.real)) == 0


def test_native_complex_mm_long_columns_and_cancellation(gpu, pkg, O):
    """columns of C longer than one 512-entry chunk (several chunks per column, bisection into A's columns),
    dense-ish operands, exact cancellation (the entry stays, as 0), a real operand promoted, dense numpy check"""
    rng = np.random.default_rng(41)
    n = 3000
    # B column 0 picks 400 columns of A with ~20 entries each: ~2500 distinct rows (5 chunks); column 1 is tiny
    A = _rand_complex(O, rng, n, n, 20 * n)
    brow = np.concatenate([rng.choice(n, 400, replace=False), [5], rng.choice(n, 1500, replace=False)])
    bcol = np.concatenate([np.zeros(400, dtype=int), [1], np.full(1500, 2)])
    Bre = O.compress(n, 3, brow, bcol, rng.normal(size=len(brow)))
    B = (n, 3, Bre[2], Bre[3], Bre[4] + 1j * rng.normal(size=len(Bre[4])))
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    ref = O.mm_z(A, B)
    assert tuples_equal(mat_to_tuple(C), ref)
    assert np.diff(ref[2]).max() > 4 * 512
    # cancellation: [[1, -1]] * [[z], [z]] = stored 0
    a = pkg.fromTriples(1, 2, [(0, 0, 1 + 0j), (0, 1, -1 + 0j)])
    b = pkg.fromTriples(2, 1, [(0, 0, 2 + 3j), (1, 0, 2 + 3j)])
    c = a * b
    assert c.pointers.tolist() == [0, 1] and c.indices.tolist() == [0] and c.values.tolist() == [0j]
    # small integers: exact against dense numpy; real x complex promoted
    Ai_, Bi_ = _rand_complex(O, rng, 60, 50, 400, ints=True), _rand_complex(O, rng, 50, 70, 400, ints=True)
    Cm = pkg.mm(tuple_to_mat(pkg, Ai_), tuple_to_mat(pkg, Bi_))
    assert np.array_equal(pkg.pack(Cm), pkg.pack(tuple_to_mat(pkg, Ai_)) @ pkg.pack(tuple_to_mat(pkg, Bi_)))
    R = pkg.Matrix(50, 60, Ai_[2], Ai_[3], np.real(Ai_[4]))
    Cr = pkg.mm(R, tuple_to_mat(pkg, Bi_))
    assert Cr.is_complex and np.array_equal(pkg.pack(Cr), pkg.pack(R) @ pkg.pack(tuple_to_mat(pkg, Bi_)))
    with pytest.raises(Exception):
        pkg.mm(tuple_to_mat(pkg, Ai_), tuple_to_mat(pkg, Ai_))


def test_native_complex_edge_shapes(gpu, pkg, O):
    """empty and degenerate operands of the packed-complex entry points: no entries at all, no columns, a single
    column, operands whose product is empty — structure as the oracle's, nothing read out of bounds"""
    z = np.zeros(0, dtype=np.complex128)
    E = pkg.Matrix(4, 3, [0, 0, 0, 0, 0], [], z)            # 3 x 4, no entries
    F = pkg.Matrix(2, 4, [0, 0, 0], [], z)                  # 4 x 2, no entries
    A = pkg.fromTriples(3, 4, [(0, 1, 1 + 2j), (2, 3, -1j)])
    B = pkg.fromTriples(4, 2, [(1, 0, 2 - 1j), (0, 1, 5 + 0j)])
    for X, Y in ((E, F), (A, F), (E, B), (A, B)):
        C = pkg.mm(X, Y)
        ref = O.mm_z(mat_to_tuple(X), mat_to_tuple(Y))
        assert C.is_complex and tuples_equal(mat_to_tuple(C), ref)
    for X, Y in ((E, E), (A, E), (E, A)):
        L = pkg.lin(2 - 1j, X, 1j, Y)
        assert tuples_equal(mat_to_tuple(L), O.lin_z(2 - 1j, mat_to_tuple(X), 1j, mat_to_tuple(Y)))
    N0 = pkg.Matrix(0, 3, [0], [], z)                       # 3 x 0
    assert pkg.lin(1j, N0, 1.0, N0).ncols == 0
    C0 = pkg.mm(A, pkg.Matrix(0, 4, [0], [], z))            # (3 x 4)(4 x 0) = 3 x 0
    assert (C0.nrows, C0.ncols, C0.pointers.tolist()) == (3, 0, [0])
    y = pkg.mulV(E, np.ones(4, dtype=np.complex128))
    assert y.shape == (3,) and not np.any(y)


def test_native_complex_fronts_block_pivoting(gpu, pkg, monkeypatch):
    """threshold pivoting inside the diagonal blocks on native complex fronts: 2 x 2 blocks [[1e-14, 3i], [3, 1e-14 i]]
    with weak complex coupling (no usable diagonal: without interchanges the factors are useless) — the speculation
    holds with block pivoting, both systems"""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_ZI_NATIVE", "1")
    rng = np.random.default_rng(12)
    m = 40
    n = m * m
    lo = np.zeros(n - 1, dtype=np.complex128)
    up = np.zeros(n - 1, dtype=np.complex128)
    lo[0::2] = 3.0
    up[0::2] = 3.0j
    d = np.where(np.arange(n) % 2 == 0, 1e-14, 1e-14j)
    far = rng.uniform(-0.1, 0.1, n - m) + 1j * rng.uniform(-0.1, 0.1, n - m)
    B = sp.diags([lo, d, up, far], (-1, 0, 1, m), format="csc")
    B.sort_indices()
    M = pkg.Matrix(n, n, B.indptr, B.indices, B.data)
    U = pkg.umfpack
    f = U.factor(M, U.analyze(M))
    st = f.stats
    assert st["complex_fronts"] == 1 and st["block_pivoting"] == 1 and st["path"] == 4
    xs = rng.normal(size=n) + 1j * rng.normal(size=n)
    for mode, op in ((U.UmfpackNormal, B), (U.UmfpackTrans, sp.csc_matrix(B.conj().T))):
        b = np.asarray(op @ xs).ravel()
        x = U.linearSolve_(f, mode, M, b)
        assert _bwd(op, x, b) <= 1e-13
    assert f.stats["complex_fronts"] == 1 and f.path == 4  # no fallback was needed


def test_native_complex_fronts_fall_back_to_static_pivoting(gpu, pkg, monkeypatch):
    """a complex mesh matrix with random values and a useless diagonal on native complex fronts: whatever the stages do
    — block pivoting holds, or static pivoting refactors the EMBEDDING on real fronts (the complex tree is given up:
    complex_fronts drops to 0 with path 5) — both systems come back backward stable; packed and device right-hand sides"""
    import scipy.sparse as sp
    import torch
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_ZI_NATIVE", "1")
    rng = np.random.default_rng(31)
    m = 44
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    S = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)), dtype=np.complex128)
    S.data = rng.uniform(-1, 1, S.nnz) + 1j * rng.uniform(-1, 1, S.nnz)
    S.setdiag(1e-12 * (rng.uniform(0.5, 1.0, S.shape[0]) + 0j))
    S = sp.csc_matrix(S)
    S.sort_indices()
    n = S.shape[0]
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    U = pkg.umfpack
    for pivot in ("1", "0"):
        monkeypatch.setenv("SPL_LU_BLOCK_PIVOT", pivot)
        fact = U.factor(M, U.analyze(M))
        st = fact.stats
        # (a zero pivot on the way — block pivoting out of candidates inside a pivot block, or total cancellation behind
        # the 1e-12 pivots of the natural order — sends numeric to static pivoting at once; else the speculation stands
        # until a solve checks it)
        assert (st["path"], st["complex_fronts"]) in ((4, 1), (5, 0)), st
        xs = rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(-1, 1, n)
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.conj().T))):
            b = np.asarray(op @ xs).ravel()
            x = U.linearSolve_(fact, mode, M, b)
            assert _bwd(op, x, b) <= 1e-12
            xd = U.linearSolveManyDevice_(fact, mode, M, torch.from_numpy(b[None, :].copy()).cuda()).cpu().numpy()[0]
            assert _bwd(op, xd, b) <= 1e-12
        st = fact.stats
        assert (st["path"], st["complex_fronts"]) in ((4, 1), (5, 0), (0, 0)), st
        if pivot == "0":
            assert st["path"] in (5, 0)  # no interchanges at all on a diagonal of 1e-12: the speculation cannot hold
