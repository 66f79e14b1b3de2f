"""The reference's own test-suite (sparse-linear/tests/Sparse.hs, tests/Test/LinearAlgebra.hs)
re-expressed against the CPU oracle: this is what PINS the oracle (no GHC exists here, so
the reference itself can never be run).  Element type: integer-valued doubles, the
analogue of the reference's `Matrix Vector Int` properties (exact arithmetic)."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import arbdim, arbitrary_dims_triples, arbitrary_triples, arbval, csc_tuple_to_scipy, tuples_equal

S = dict(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))


def ident(n):
    return (n, n, np.arange(n + 1), np.arange(n), np.ones(n))


def diag(v):
    n = len(v)
    return (n, n, np.arange(n + 1), np.arange(n), np.asarray(v, dtype=float))


def zeros(nr, nc):
    return (nr, nc, np.zeros(nc + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0))


# -- fromTriples: checkMatrix arbitrary (Sparse.hs:23) ---------------------------------------
@settings(**S)
@given(arbitrary_dims_triples())
def test_fromTriples_format(O, dt):
    nr, nc, triples = dt
    m = O.fromTriples(nr, nc, triples)
    assert O.check_matrix(m) == 0
    # independent third party: scipy sums duplicates too; explicit zeros stay stored in ours
    dense = np.zeros((nr, nc))
    for r, c, x in triples:
        dense[r, c] += x
    assert np.array_equal(csc_tuple_to_scipy(m).toarray(), dense)
    assert len(m[3]) == len({(r, c) for r, c, _ in triples})  # one entry per distinct position


def test_compress_errors(O):
    import pytest
    with pytest.raises(O.OracleError, match="out of bounds at 1"):
        O.fromTriples(2, 2, [(0, 0, 1.0), (2, 0, 1.0)])
    with pytest.raises(O.OracleError, match="out of bounds at 0"):
        O.fromTriples(2, 2, [(0, -1, 1.0)])
    with pytest.raises(O.OracleError, match="lengths differ"):
        O.compress(2, 2, [0], [0, 1], [1.0])


def test_explicit_zeros_kept_and_dedup(O):
    m = O.fromTriples(3, 2, [(1, 0, 2.0), (1, 0, -2.0), (0, 1, 0.0), (2, 1, 5.0), (0, 1, 0.0)])
    assert m[2].tolist() == [0, 1, 3] and m[3].tolist() == [1, 0, 2] and m[4].tolist() == [0.0, 0.0, 5.0]
    d, ix, xs = O.dedup_in_place(9, [3, 1, 3, 3, 0], [1.0, 2.0, 4.0, 8.0, 16.0])
    assert d == 2 and ix.tolist() == [0, 1, 3, 9, 9] and xs[:3].tolist() == [16.0, 2.0, 13.0]


# -- mulV: ident `mulV` v == v (Sparse.hs:41-47) -----------------------------------------------
@settings(**S)
@given(st.lists(arbval, min_size=1, max_size=40))
def test_ident_mulV(O, v):
    v = np.array(v)
    assert np.array_equal(O.mulV(ident(len(v)), v), v)


@settings(**S)
@given(arbitrary_dims_triples(), st.data())
def test_mulV_vs_scipy_and_csr_form(O, dt, data):
    nr, nc, triples = dt
    m = O.fromTriples(nr, nc, triples)
    x = np.array(data.draw(st.lists(arbval, min_size=nc, max_size=nc)))
    y = O.mulV(m, x)
    assert np.array_equal(y, csc_tuple_to_scipy(m) @ x)  # exact: integer-valued data
    # the CSR row-gather statement of the same fold is bit-identical
    t = O.transpose(m)
    y2 = np.zeros(nr)
    O.csr_gaxpy32(t[2], t[3], t[4], x, y2)
    assert np.array_equal(y, y2)


def test_axpy_dimension_guards(O):
    import pytest
    with pytest.raises(O.OracleError):
        O.mulV(ident(3), np.ones(4))
    with pytest.raises(O.OracleError):
        O.axpy_(ident(3), np.ones(3), np.ones(2))


def test_csr_gather_equals_csc_scatter_on_floats(O):
    """non-exact data: the two statements of the fold still agree bit for bit"""
    rng = np.random.default_rng(1)
    m = O.compress(200, 150, rng.integers(0, 200, 4000), rng.integers(0, 150, 4000), rng.normal(size=4000))
    x = rng.normal(size=150)
    y0 = rng.normal(size=200)
    t = O.transpose(m)
    y2 = y0.copy()
    O.csr_gaxpy32(t[2], t[3], t[4], x, y2)
    assert np.array_equal(O.axpy(m, x, y0), y2)
    B = rng.normal(size=(150, 5))
    C = O.mulM(m, B)
    for j in range(5):
        assert np.array_equal(C[:, j], O.mulV(m, B[:, j]))


# -- addition (Sparse.hs:49-54,147-178) ------------------------------------------------------------
@settings(**S)
@given(arbitrary_dims_triples())
def test_add_ident_and_inverse(O, dt):
    nr, nc, triples = dt
    a = O.fromTriples(nr, nc, triples)
    assert tuples_equal(O.add(a, zeros(nr, nc)), a)          # a + zeros == a
    z = O.sub(a, a)                                           # a - a == cmap (const 0) a
    assert tuples_equal(z, (a[0], a[1], a[2], a[3], np.zeros(len(a[4]))))
    assert O.check_matrix(z) == 0


@settings(**S)
@given(st.data())
def test_add_commute_assoc_format(O, data):
    nr, nc = data.draw(arbdim), data.draw(arbdim)
    a, b, c = (O.fromTriples(nr, nc, data.draw(arbitrary_triples(nr, nc))) for _ in range(3))
    assert tuples_equal(O.add(a, b), O.add(b, a))
    assert tuples_equal(O.add(a, O.add(b, c)), O.add(O.add(a, b), c))
    s = O.add(a, b)
    assert O.check_matrix(s) == 0
    assert np.array_equal(csc_tuple_to_scipy(s).toarray(), csc_tuple_to_scipy(a).toarray() + csc_tuple_to_scipy(b).toarray())


def test_lin_scalars(O):
    a = O.fromTriples(3, 3, [(0, 0, 1.0), (2, 1, 3.0)])
    b = O.fromTriples(3, 3, [(0, 0, 5.0), (1, 2, 7.0)])
    r = O.lin(2.0, a, -1.0, b)
    assert r[2].tolist() == [0, 1, 2, 3] and r[3].tolist() == [0, 2, 1] and r[4].tolist() == [-3.0, 6.0, -7.0]


# -- transpose (Sparse.hs:56-59) ------------------------------------------------------------------------
@settings(**S)
@given(st.lists(arbval, min_size=1, max_size=30))
def test_transpose_diag(O, v):
    d = diag(v)
    assert tuples_equal(O.transpose(d), d)


@settings(**S)
@given(arbitrary_dims_triples())
def test_transpose_involution_and_scipy(O, dt):
    nr, nc, triples = dt
    a = O.fromTriples(nr, nc, triples)
    t = O.transpose(a)
    assert O.check_matrix(t) == 0
    assert tuples_equal(O.transpose(t), a)
    assert np.array_equal(csc_tuple_to_scipy(t).toarray(), csc_tuple_to_scipy(a).toarray().T)


def test_ctrans_fixtures_real(O):
    """tests/Sparse.hs:61-73: the hermitian 2x2 fixtures whose entries are real
    (sigma_y is complex: SURVEY.md §8f rank 3, not built yet)"""
    m = O.fromTriples(2, 2, [(0, 0, 2.0), (0, 1, -1.0), (1, 0, -1.0), (1, 1, 2.0)])
    assert tuples_equal(O.transpose(m), m)
    sx = O.fromTriples(2, 2, [(0, 1, 1.0), (1, 0, 1.0)])
    assert tuples_equal(O.transpose(sx), sx)


# -- mul (Sparse.hs:75-102) --------------------------------------------------------------------------------
@settings(**S)
@given(arbitrary_dims_triples())
def test_mul_identities(O, dt):
    nr, nc, triples = dt
    a = O.fromTriples(nr, nc, triples)
    assert tuples_equal(O.mm(ident(nr), a), a)
    assert tuples_equal(O.mm(a, ident(nc)), a)
    assert tuples_equal(O.mm(ident(nr), a, literal=True), a)


@settings(**S)
@given(st.data())
def test_mul_assoc_format_literal(O, data):
    m, n, p, q = (data.draw(arbdim) for _ in range(4))
    a = O.fromTriples(m, n, data.draw(arbitrary_triples(m, n)))
    b = O.fromTriples(n, p, data.draw(arbitrary_triples(n, p)))
    c = O.fromTriples(p, q, data.draw(arbitrary_triples(p, q)))
    ab = O.mm(a, b)
    assert O.check_matrix(ab) == 0
    assert tuples_equal(O.mm(ab, c), O.mm(a, O.mm(b, c)))
    # the touched-list variant is the literal dense-SPA algorithm, result for result
    assert tuples_equal(ab, O.mm(a, b, literal=True))
    assert np.array_equal(csc_tuple_to_scipy(ab).toarray(), csc_tuple_to_scipy(a).toarray() @ csc_tuple_to_scipy(b).toarray())


def test_mm_cancellation_keeps_stored_zero(O):
    a = O.fromTriples(1, 2, [(0, 0, 1.0), (0, 1, -1.0)])
    b = O.fromTriples(2, 1, [(0, 0, 1.0), (1, 0, 1.0)])
    c = O.mm(a, b)
    assert c[2].tolist() == [0, 1] and c[3].tolist() == [0] and c[4].tolist() == [0.0]
    import pytest
    with pytest.raises(O.OracleError):
        O.mm(a, a)


# -- FFI seam (Sparse.hs:138-145) -------------------------------------------------------------------------------
@settings(**S)
@given(arbitrary_dims_triples())
def test_fromForeign_withConstMatrix_roundtrip(O, dt):
    nr, nc, triples = dt
    a = O.fromTriples(nr, nc, triples)
    assert tuples_equal(O.from_foreign(*O.with_const_matrix(a)), a)


# -- solve (suitesparse/tests/test-umfpack.hs:16-19, feast/tests/test-feast.hs) --------------------------------------
@settings(**S)
@given(st.lists(st.floats(-1e6, 1e6, allow_nan=False), min_size=1, max_size=30))
def test_ident_solve_exact(O, v):
    v = np.array(v)
    x, st_ = O.linear_solve(ident(len(v)), v)
    assert st_ == 0 and np.array_equal(x, v)
    x, st_ = O.linear_solve(ident(len(v)), v, sys=1)
    assert np.array_equal(x, v)


def test_feast_fixture_matrix_solve(O):
    """the 2x2 matrix of feast/tests/test-feast.hs:25 ([[2,-1],[-1,2]], eigenvalues 1 and 3):
    FEAST's inner step is (ze*I - A) \\ b; check a shifted real solve against the closed form"""
    A = O.fromTriples(2, 2, [(0, 0, 2.0), (0, 1, -1.0), (1, 0, -1.0), (1, 1, 2.0)])
    b = np.array([1.0, 0.0])
    x, _ = O.linear_solve(A, b)
    assert O.count_not_close(x, np.array([2.0 / 3.0, 1.0 / 3.0])) == 0
    # eigenpairs: A v = lambda v for lambda in {1, 3}
    for lam, v in ((1.0, [1.0, 1.0]), (3.0, [1.0, -1.0])):
        assert np.array_equal(O.mulV(A, np.array(v)), lam * np.array(v))


def test_solve_vs_scipy_and_transposed(O):
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(2)
    n = 300
    k = 3000
    A = O.compress(n, n, np.concatenate([rng.integers(0, n, k), np.arange(n)]),
                   np.concatenate([rng.integers(0, n, k), np.arange(n)]),
                   np.concatenate([rng.normal(size=k), np.full(n, 20.0)]))
    xs = rng.uniform(0.5, 1.5, n)
    S_ = csc_tuple_to_scipy(A)
    for sys_, M in ((0, S_), (1, S_.T.tocsc())):
        b = M @ xs
        x, st_ = O.linear_solve(A, b, sys=sys_)
        assert st_ == 0
        assert O.count_not_close(x, xs, 1e-10) == 0
        assert O.count_not_close(x, spla.spsolve(M.tocsc(), b), 1e-10) == 0


def test_singular_warning(O):
    A = O.fromTriples(2, 2, [(0, 0, 1.0), (1, 0, 1.0)])
    _, st_ = O.linear_solve(A, np.ones(2))
    assert st_ == 1  # UMFPACK_WARNING_singular_matrix: positive, not fatal (Umfpack.hs:101)


# -- synthetic workloads (SURVEY.md §8d closed forms) ----------------------------------------------------------------------
def test_poisson_closed_forms(O):
    n = 100
    rp, ci, v = O.gen_poisson2d_csr(n)
    assert rp[-1] == 5 * n * n - 4 * n == 49600
    y = np.zeros(n * n)
    O.csr_gaxpy32(rp, ci, v, np.ones(n * n), y)
    ix, iy = np.meshgrid(np.arange(n), np.arange(n))
    nb = (ix > 0).astype(float) + (ix < n - 1) + (iy > 0) + (iy < n - 1)
    assert np.array_equal(y, (4 - nb).ravel())
    m = 12
    rp, ci, v = O.gen_poisson3d_csr(m)
    assert rp[-1] == 7 * m ** 3 - 6 * m ** 2
    assert 7 * 200 ** 3 - 6 * 200 ** 2 == 55_760_000


def test_random_generator_invariants(O):
    n, K = 5000, 20
    rp, ci, v = O.gen_random_csr(n, K)
    assert np.all(np.diff(rp) <= K) and np.all(np.diff(rp) >= 1)
    for r in range(0, n, 97):
        c = ci[rp[r]:rp[r + 1]]
        assert np.all(np.diff(c) > 0) and c.min() >= 0 and c.max() < n
    assert np.all(v >= 0.5)  # sums of values in [0.5, 1.5): no cancellation
    # row blocks regenerate identically (what every rank of a multi-GPU run relies on)
    rp2, ci2, v2 = O.gen_random_csr(n, K, row0=1234, row1=2345)
    assert np.array_equal(ci2, ci[rp[1234]:rp[2345]]) and np.array_equal(v2, v[rp[1234]:rp[2345]])
    x = O.gen_vector(n)
    assert np.array_equal(O.gen_vector(n, j0=100, j1=200), x[100:200]) and x.min() >= 0.5 and x.max() < 1.5


def test_banded_generator(O):
    n = 4000
    rp, ci, v = O.gen_banded_csr(n)
    r = 2000
    offs = ci[rp[r]:rp[r + 1]] - r
    assert len(offs) == 20 and offs[0] == -1000 and offs[-1] == 1000 and np.all(np.diff(offs) > 0)
    assert rp[1] - rp[0] == 10  # row 0: only non-negative offsets survive


def test_rmat_generator(O):
    r, c, v = O.gen_rmat_coo(10, 5000)
    assert r.min() >= 0 and r.max() < 1024 and c.max() < 1024
    r2, c2, v2 = O.gen_rmat_coo(10, 1000, e0=2000)
    assert np.array_equal(r2, r[2000:3000]) and np.array_equal(c2, c[2000:3000]) and np.array_equal(v2, v[2000:3000])
    # Erdos-Renyi parameters: rows roughly uniform
    assert abs(np.mean(r) - 511.5) < 20
    rs, _, _ = O.gen_rmat_coo(10, 5000, abc=(0.57, 0.19, 0.19))
    assert np.mean(rs) < 400  # Graph500 skew towards low ids


def test_oracle_kronecker_and_take_diag_against_dense():
    """numpy restatement of kronecker / takeDiag (Sparse.hs:597-648) against np.kron"""
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    for (ar, ac, ak), (br, bc, bk) in (((4, 3, 7), (2, 5, 6)), ((1, 1, 1), (6, 6, 20)), ((3, 3, 0), (2, 2, 3))):
        A = O.compress(ar, ac, rng.integers(0, ar, ak), rng.integers(0, ac, ak), rng.integers(-4, 5, ak).astype(float))
        B = O.compress(br, bc, rng.integers(0, br, bk), rng.integers(0, bc, bk), rng.integers(-4, 5, bk).astype(float))
        K = O.kronecker(A, B)
        assert O.check_matrix(K) == 0

        def dense(m):
            out = np.zeros((m[0], m[1]))
            out[m[3], np.repeat(np.arange(m[1]), np.diff(m[2]))] = m[4]
            return out

        assert np.array_equal(dense(K), np.kron(dense(A), dense(B)))
        assert np.array_equal(O.take_diag(K), np.diag(dense(K))[:min(K[0], K[1])])


def test_lin_complex_restatement_against_dense(O):
    """orc_lin_z (glin at Complex Double, Sparse.hs:401-431): against dense numpy arithmetic, union pattern with
    cancellations kept, and equal to the real restatement when everything is real"""
    rng = np.random.default_rng(8)
    for nr, nc, k in ((4, 3, 6), (30, 20, 150)):
        def rnd():
            r, c = rng.integers(0, nr, k), rng.integers(0, nc, k)
            re = O.compress(nr, nc, r, c, rng.integers(-3, 4, k).astype(float))
            im = O.compress(nr, nc, r, c, rng.integers(-3, 4, k).astype(float))
            return (nr, nc, re[2], re[3], re[4] + 1j * im[4])
        A, B = rnd(), rnd()
        for alpha, beta in ((-1.0, 0.5 + 2j), (1j, 1.0), (2.0, -3.0)):
            Cm = O.lin_z(alpha, A, beta, B)
            assert O.check_matrix((Cm[0], Cm[1], Cm[2], Cm[3], np.real(Cm[4]))) == 0

            def dense(m):
                d = np.zeros((m[0], m[1]), dtype=complex)
                cols = np.repeat(np.arange(m[1]), np.diff(m[2]))
                d[m[3], cols] = m[4]
                return d
            assert np.array_equal(dense(Cm), alpha * dense(A) + beta * dense(B))  # small integers: exact
            union = (dense(A) != 0) | (dense(B) != 0)
            assert int(Cm[2][-1]) >= int(union.sum())  # explicit zeros of the operands stay, too
        real = lambda m: (m[0], m[1], m[2], m[3], np.ascontiguousarray(np.real(m[4])))
        Cr = O.lin(2.0, real(A), -0.5, real(B))
        Cz = O.lin_z(2.0, real(A), -0.5, real(B))
        assert np.array_equal(Cr[2], Cz[2]) and np.array_equal(Cr[3], Cz[3])
        assert np.array_equal(Cr[4], np.real(Cz[4])) and not np.any(np.imag(Cz[4]))


def test_mm_complex_restatement_against_dense(O):
    """orc_mm_z (mm at Complex Double, Sparse.hs:691-702): exact against dense numpy on small integers, equal to the
    real restatement on real data, format invariants"""
    rng = np.random.default_rng(15)

    def rnd(nr, nc, k):
        r, c = rng.integers(0, nr, k), rng.integers(0, nc, k)
        re = O.compress(nr, nc, r, c, rng.integers(-3, 4, k).astype(float))
        im = O.compress(nr, nc, r, c, rng.integers(-3, 4, k).astype(float))
        return (nr, nc, re[2], re[3], re[4] + 1j * im[4])

    def dense(m):
        d = np.zeros((m[0], m[1]), dtype=complex)
        d[m[3], np.repeat(np.arange(m[1]), np.diff(m[2]))] = m[4]
        return d
    for m, n, p, k in ((3, 4, 2, 8), (25, 30, 20, 200)):
        A, B = rnd(m, n, k), rnd(n, p, k)
        Cm = O.mm_z(A, B)
        assert O.check_matrix((Cm[0], Cm[1], Cm[2], Cm[3], np.real(Cm[4]))) == 0
        assert np.array_equal(dense(Cm), dense(A) @ dense(B))
        real = lambda t: (t[0], t[1], t[2], t[3], np.ascontiguousarray(np.real(t[4])))
        Cr, Cz = O.mm(real(A), real(B)), O.mm_z(real(A), real(B))
        assert np.array_equal(Cr[2], Cz[2]) and np.array_equal(Cr[3], Cz[3]) and np.array_equal(Cr[4], np.real(Cz[4]))
    try:
        O.mm_z(rnd(3, 4, 5), rnd(3, 4, 5))
        raise AssertionError("inner dimension mismatch not reported")
    except O.OracleError:
        pass
