"""The umfpack_di_* ABI on the GPU (suitesparse/src/Numeric/LinearAlgebra/Umfpack.hs).
The reference pins exactly one thing: `ident <\\> v == v` (suitesparse/tests/test-umfpack.hs:16-19,
exact equality).  Everything else is checked on the solution: manufactured solutions, residuals,
the CPU oracle and scipy (UMFPACK's pivot order is an un-pinned implementation detail)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import csc_tuple_to_scipy, mat_to_tuple, tuple_to_mat

pytestmark = pytest.mark.gpu


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.lists(st.floats(-1e6, 1e6, allow_nan=False), min_size=1, max_size=60))
def test_ident_solve_exact(gpu, pkg, v):
    v = np.array(v)
    assert np.array_equal(pkg.umfpack.solve(pkg.ident(len(v)), v), v)  # prop_linSolveId


def test_feast_fixture_matrix(gpu, pkg, O):
    A = pkg.fromTriples(2, 2, [(0, 0, 2), (0, 1, -1), (1, 0, -1), (1, 1, 2)])
    x = pkg.umfpack.solve(A, np.array([1.0, 0.0]))
    assert O.count_not_close(x, np.array([2.0 / 3.0, 1.0 / 3.0])) == 0


@pytest.mark.parametrize("force_pivot", ["0", "1"])
@pytest.mark.parametrize("m", [5, 30, 100])
def test_poisson2d_manufactured(gpu, pkg, O, m, force_pivot, monkeypatch):
    # Poisson matrices are column diagonally dominant: "0" takes the blocked no-interchange
    # factorisation, "1" forces the partial-pivoting band LU; both must give the solution
    monkeypatch.setenv("SPL_LU_FORCE_PIVOT", force_pivot)
    n = m * m
    rp, ci, v = O.gen_poisson2d_csr(m)
    A = pkg.Matrix(n, n, rp, ci, v)
    xs = O.gen_vector(n)
    b = O.mulV(mat_to_tuple(A), xs)
    fact = pkg.umfpack.factor(A, pkg.umfpack.analyze(A))
    x = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackNormal, A, b)
    assert O.count_not_close(x, xs, 1e-10) == 0
    xt = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackTrans, A, b)  # symmetric: same answer
    assert O.count_not_close(xt, xs, 1e-10) == 0
    xo, _ = O.linear_solve(mat_to_tuple(A), b)
    assert O.count_not_close(x, xo, 1e-10) == 0


@pytest.mark.parametrize("force_pivot", ["0", "1"])
def test_poisson3d_wide_band_path(gpu, pkg, O, force_pivot, monkeypatch):
    """m = 14: band half-width 196 -> the one-kernel-pair-per-column path when pivoting is forced"""
    monkeypatch.setenv("SPL_LU_FORCE_PIVOT", force_pivot)
    m = 14
    n = m ** 3
    rp, ci, v = O.gen_poisson3d_csr(m)
    A = pkg.Matrix(n, n, rp, ci, v)
    xs = O.gen_vector(n)
    b = O.mulV(mat_to_tuple(A), xs)
    x = pkg.umfpack.solve(A, b)
    assert O.count_not_close(x, xs, 1e-10) == 0


def test_unsymmetric_needs_pivoting(gpu, pkg, O):
    """zero diagonal + unsymmetric values: partial pivoting inside the band is required"""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(5)
    n = 400
    k = 2400
    rows = np.concatenate([rng.integers(0, n, k), (np.arange(n) + 1) % n])
    cols = np.concatenate([rng.integers(0, n, k), np.arange(n)])
    vals = np.concatenate([rng.normal(size=k), np.full(n, 7.0)])
    A = O.compress(n, n, rows, cols, vals)
    keep = A[3] != np.repeat(np.arange(n), np.diff(A[2]))  # drop the diagonal entirely
    # rebuild without diagonal entries
    cols_e = np.repeat(np.arange(n), np.diff(A[2]))[keep]
    A = O.compress(n, n, A[3][keep], cols_e, A[4][keep])
    M = tuple_to_mat(pkg, A)
    S = csc_tuple_to_scipy(A)
    xs = rng.uniform(0.5, 1.5, n)
    for mode, op in ((pkg.umfpack.UmfpackNormal, S), (pkg.umfpack.UmfpackTrans, S.T.tocsc())):
        b = op @ xs
        fact = pkg.umfpack.factor(M, pkg.umfpack.analyze(M))
        x = pkg.umfpack.linearSolve_(fact, mode, M, b)
        r = np.max(np.abs(op @ x - b)) / (np.max(np.abs(b)) + np.max(np.abs(x)))
        assert r < 1e-12
        ref = spla.spsolve(op.tocsc(), b)
        assert np.max(np.abs(x - ref)) / np.max(np.abs(ref)) < 1e-8


def test_same_analysis_many_factorisations_many_rhs(gpu, pkg, O):
    """the FEAST usage pattern (Feast.hs:210-218): one analyze, several same-pattern factors,
    several right-hand sides each"""
    m = 20
    n = m * m
    rp, ci, v = O.gen_poisson2d_csr(m)
    A = pkg.Matrix(n, n, rp, ci, v)
    an = pkg.umfpack.analyze(A)
    for shift in (0.0, 1.5, -0.25):
        As = pkg.Matrix(n, n, rp, ci, np.where(v == 4.0, v + shift, v))
        fact = pkg.umfpack.factor(As, an)
        for seed in (1, 2, 3):
            xs = O.gen_vector(n, seed=seed)
            b = O.mulV(mat_to_tuple(As), xs)
            assert O.count_not_close(pkg.umfpack.linearSolve_(fact, 0, As, b), xs, 1e-10) == 0
    xs = [O.gen_vector(n, seed=s) for s in (7, 8)]
    bs = [O.mulV(mat_to_tuple(A), x) for x in xs]
    for x, xo in zip(pkg.umfpack.linearSolve(A, bs), xs):
        assert O.count_not_close(x, xo, 1e-10) == 0


def test_status_codes(gpu, pkg):
    import ctypes as C
    L = pkg.umfpack._declare()
    # singular: positive warning, not an exception (Umfpack.hs:101 throws only on < 0)
    S = pkg.fromTriples(2, 2, [(0, 0, 1.0), (1, 0, 1.0)])
    f = pkg.umfpack.factor(S, pkg.umfpack.analyze(S))
    assert f.status == 1
    # different pattern
    A = pkg.ident(3)
    an = pkg.umfpack.analyze(A)
    B = pkg.fromTriples(3, 3, [(0, 0, 1.0), (1, 1, 1.0), (2, 2, 1.0), (0, 2, 1.0)])
    with pytest.raises(pkg.umfpack.UmfpackError):
        pkg.umfpack.factor(B, an)
    # bad handles (rectangular matrices: test_rectangular_matrices_are_analysed_factored_and_refused_by_solve)
    x = np.zeros(3)
    ap = (C.c_int * 4)(0, 1, 2, 3)
    ai = (C.c_int * 3)(0, 1, 2)
    ax = (C.c_double * 3)(1, 1, 1)
    assert L.umfpack_di_solve(0, ap, ai, ax, pkg._ffi.p_f64(x), pkg._ffi.p_f64(x), None, None, None) == -3
    h = C.c_void_p()
    L.umfpack_di_free_numeric(C.byref(h))  # NULL: no-op
    L.umfpack_di_free_symbolic(C.byref(h))


def test_rectangular_matrices_are_analysed_factored_and_refused_by_solve(gpu, pkg):
    """UMFPACK analyses and factors rectangular matrices and refuses to solve with them (UMFPACK_ERROR_invalid_system,
    "the matrix is not square"); the reference's binding passes n_row and n_col through (Umfpack.hs:60-69) and throws
    on negative statuses only.  Here the statuses follow UMFPACK's: symbolic 0, numeric 0 for full structural rank over
    the non-zero entries and the singular-matrix warning (+1) otherwise, different_pattern as for square matrices, solve
    -13 in every form; real and complex; nothing is factored (round 4; rounds 1 - 3 refused at the analysis)."""
    import ctypes as C
    U = pkg.umfpack
    wide = pkg.fromTriples(2, 3, [(0, 0, 1.0), (1, 1, 2.0), (0, 2, 3.0)])          # rank 2 = min(2, 3)
    tall = pkg.fromTriples(4, 2, [(0, 0, 1.0), (3, 0, 2.0), (3, 1, 3.0)])          # rank 2: column 1 must take row 3, column 0 row 0
    short = pkg.fromTriples(3, 2, [(2, 0, 1.0), (2, 1, 1.0)])                      # both columns only reach row 2: rank 1
    zeroed = pkg.fromTriples(2, 3, [(0, 0, 0.0), (1, 1, 2.0), (0, 2, 0.0)])        # stored zeros are no pivots: rank 1
    # structurally regular, numerically rank 1 — UMFPACK meets an exactly zero pivot and warns; small matrices
    # are eliminated on the host for their numerical rank (real and complex)
    ones = pkg.fromTriples(3, 2, [(i, j, 1.0) for j in range(2) for i in range(3)])
    zones = pkg.fromTriples(2, 3, [(i, j, (1.0 + 2.0j) * (j + 1)) for j in range(3) for i in range(2)])  # rows equal: rank 1
    for mat, want in ((wide, 0), (tall, 0), (short, 1), (zeroed, 1), (ones, 1), (zones, 1)):
        an = U.analyze(mat)
        f = U.factor(mat, an)
        assert f.status == want
        with pytest.raises(U.UmfpackError) as e:
            U.linearSolve_(f, U.UmfpackNormal, mat, np.ones(mat.nrows, dtype=complex if mat.is_complex else float))
        assert e.value.status == -13
        with pytest.raises(U.UmfpackError):
            U.linearSolveMany_(f, U.UmfpackTrans, mat, [np.ones(mat.nrows, dtype=complex if mat.is_complex else float)] * 2)
    # the pattern is checked as for square matrices
    other = pkg.fromTriples(2, 3, [(0, 0, 1.0), (1, 1, 2.0), (1, 2, 3.0)])
    with pytest.raises(U.UmfpackError) as e:
        U.factor(other, U.analyze(wide))
    assert e.value.status == -11
    # complex (umfpack_zi_*), a longer chain of augmenting paths: a 40 x 50 band whose columns j reach rows j - 1 and j
    n_row, n_col = 40, 50
    tri = [(max(j - 1, 0), j, 1.0 + 1.0j) for j in range(n_col) if max(j - 1, 0) < n_row]
    tri += [(j, j, 2.0 - 1.0j) for j in range(min(n_row, n_col)) if j > 0]
    Z = pkg.fromTriples(n_row, n_col, sorted(set(tri), key=lambda t: (t[1], t[0])))
    assert Z.is_complex
    fz = U.factor(Z, U.analyze(Z))
    assert fz.status == 0  # rank 40 = min(40, 50)
    with pytest.raises(U.UmfpackError) as e:
        U.linearSolve_(fz, U.UmfpackNormal, Z, np.ones(n_row, dtype=complex))
    assert e.value.status == -13
    # the device-pointer form refuses as well
    L = U._declare()
    nr, nc, ap, ai, ax = wide._tuple32()
    f = U.factor(wide, U.analyze(wide))
    assert L.spl_umfpack_di_solve_many_dev(0, pkg._ffi.p_i32(ap), pkg._ffi.p_i32(ai), pkg._ffi.p_f64(ax), 1, None, None, f.value) == -13
    # Info[UMFPACK_STATUS] carries the status on the error returns of the solves as well, real and complex
    info = np.full(90, 7.0)
    xb = np.ones(3)
    assert L.umfpack_di_solve(0, pkg._ffi.p_i32(ap), pkg._ffi.p_i32(ai), pkg._ffi.p_f64(ax), pkg._ffi.p_f64(xb), pkg._ffi.p_f64(xb),
                              f.value, None, pkg._ffi.p_f64(info)) == -13 and info[0] == -13.0
    sq = pkg.fromTriples(2, 2, [(0, 0, 2.0), (1, 1, 4.0)])
    fs = U.factor(sq, U.analyze(sq))
    _nr, _nc, sp_, si_, sx_ = sq._tuple32()
    info[:] = 7.0
    assert L.umfpack_di_solve(5, pkg._ffi.p_i32(sp_), pkg._ffi.p_i32(si_), pkg._ffi.p_f64(sx_), pkg._ffi.p_f64(xb), pkg._ffi.p_f64(xb),
                              fs.value, None, pkg._ffi.p_f64(info)) == -13 and info[0] == -13.0  # no such system
    info[:] = 7.0
    assert L.umfpack_di_solve(0, pkg._ffi.p_i32(sp_), pkg._ffi.p_i32(si_), pkg._ffi.p_f64(sx_), pkg._ffi.p_f64(xb), pkg._ffi.p_f64(np.array([2.0, 4.0, 0.0])),
                              fs.value, None, pkg._ffi.p_f64(info)) == 0 and info[0] == 0.0 and xb[0] == 1.0 and xb[1] == 1.0


def test_nopiv_path_unsymmetric_dominant_and_transposed(gpu, pkg, O):
    """column diagonally dominant but unsymmetric, unequal bandwidths: exercises every masked edge of
    the blocked factorisation and all four blocked solves (L, U, U^T, L^T)"""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(17)
    for n, lo, hi in ((70, 3, 9), (500, 40, 7), (1000, 1, 65), (333, 100, 100)):
        rows, cols, vals = [], [], []
        for j in range(n):
            for i in range(max(0, j - hi), min(n, j + lo + 1)):
                if i != j and rng.random() < 0.5:
                    rows.append(i); cols.append(j); vals.append(rng.normal())
        A0 = O.compress(n, n, rows, cols, vals)
        colsum = np.zeros(n)
        np.add.at(colsum, np.repeat(np.arange(n), np.diff(A0[2])), np.abs(A0[4]))
        A = O.compress(n, n, rows + list(range(n)), cols + list(range(n)), vals + list(colsum + 1.0))
        M = tuple_to_mat(pkg, A)
        S = csc_tuple_to_scipy(A)
        xs = rng.uniform(0.5, 1.5, n)
        fact = pkg.umfpack.factor(M, pkg.umfpack.analyze(M))
        for mode, op in ((0, S), (1, S.T.tocsc())):
            b = op @ xs
            x = pkg.umfpack.linearSolve_(fact, mode, M, b)
            assert O.count_not_close(x, xs, 1e-10) == 0, (n, lo, hi, mode)
            assert O.count_not_close(x, spla.spsolve(op.tocsc(), b), 1e-10) == 0


def test_speculative_no_interchange_path_spd(gpu, pkg, O):
    """symmetric positive definite but NOT diagonally dominant (B^T B of a banded B): elimination
    without interchanges is stable, the speculative path is taken and kept"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(23)
    n = 600
    Bm = sp.diags([rng.uniform(0.5, 1.5, n - abs(d)) for d in (-7, -1, 0, 1, 3)], (-7, -1, 0, 1, 3), format="csc")
    S = (Bm.T @ Bm + 0.05 * sp.identity(n)).tocsc()
    S.sort_indices()
    assert np.any(np.abs(S.diagonal()) < np.asarray(abs(S).sum(axis=0)).ravel() - np.abs(S.diagonal()))
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    xs = rng.uniform(0.5, 1.5, n)
    fact = pkg.umfpack.factor(M, pkg.umfpack.analyze(M))
    assert fact.path in (2, 4)  # band or multifrontal (SPL_LU_METHOD), as a speculation
    for mode in (0, 1):
        b = S @ xs  # symmetric: same system both ways
        x = pkg.umfpack.linearSolve_(fact, mode, M, b)
        assert np.max(np.abs(S @ x - b)) / (np.max(np.abs(b)) + np.max(np.abs(x))) < 1e-13
        assert np.max(np.abs(x - spla.spsolve(S, b))) / np.max(np.abs(xs)) < 1e-8
    assert fact.path in (2, 4)  # the speculation held


def test_speculation_that_fails_is_replaced_by_pivoting(gpu, pkg, O, monkeypatch):
    """tiny (not zero) diagonal: the no-interchange factors exist but are useless; the solve
    notices through the backward error, refactors with partial pivoting and answers correctly"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(29)
    n = 300
    # 2x2 diagonal blocks [[1e-14, 3], [3, 1e-14]] (well conditioned, but useless as pivots in
    # the given order) plus a weak band coupling
    off = np.zeros(n - 1)
    off[0::2] = 3.0
    S = sp.diags([off, np.full(n, 1e-14), off, rng.uniform(-0.1, 0.1, n - 2), rng.uniform(-0.1, 0.1, n - 5)],
                 (-1, 0, 1, -2, 5), format="csc")
    S.sort_indices()
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    xs = rng.uniform(0.5, 1.5, n)
    b = S @ xs
    monkeypatch.setenv("SPL_LU_STATIC_PIVOT", "0")  # (the static-pivoting stage has its own tests below)
    fact = pkg.umfpack.factor(M, pkg.umfpack.analyze(M))
    assert fact.path in (2, 4)
    x = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackNormal, M, b)
    assert fact.path == 0  # replaced
    ref = spla.spsolve(S, b)
    assert np.max(np.abs(S @ x - b)) / (np.max(np.abs(b)) + np.max(np.abs(x))) < 1e-12
    assert np.max(np.abs(x - ref)) / np.max(np.abs(ref)) < 1e-8
    xt = pkg.umfpack.linearSolve_(fact, pkg.umfpack.UmfpackTrans, M, S.T @ xs)
    assert np.max(np.abs(S.T @ xt - S.T @ xs)) / np.max(np.abs(S.T @ xs)) < 1e-12
    # SPL_LU_FORCE_PIVOT=0: no speculation, this matrix goes straight to partial pivoting
    monkeypatch.setenv("SPL_LU_FORCE_PIVOT", "0")
    assert pkg.umfpack.factor(M, pkg.umfpack.analyze(M)).path == 0


@pytest.mark.parametrize("k", [1, 2, 4, 7])
@pytest.mark.parametrize("force_pivot", ["0", "1"])
def test_batched_linear_solve_matches_one_at_a_time(gpu, pkg, O, k, force_pivot, monkeypatch):
    """linearSolveMany_ (all right-hand sides through the factors together, groups of 4 in the
    blocked solves) == map linearSolve_ (Umfpack.hs:103-108), both systems, both LU paths"""
    monkeypatch.setenv("SPL_LU_FORCE_PIVOT", force_pivot)
    m = 23
    n = m * m
    rp, ci, v = O.gen_poisson2d_csr(m)
    v = v.copy()
    v[ci == np.repeat(np.arange(n), np.diff(rp))] += 0.25  # unsymmetric-ish scaling keeps dominance
    A = pkg.Matrix(n, n, rp, ci, v)
    rng = np.random.default_rng(k)
    bs = [rng.uniform(0.5, 1.5, n) for _ in range(k)]
    U = pkg.umfpack
    fact = U.factor(A, U.analyze(A))
    for mode in (U.UmfpackNormal, U.UmfpackTrans):
        many = U.linearSolveMany_(fact, mode, A, bs)
        assert len(many) == k
        for b, x in zip(bs, many):
            one = U.linearSolve_(fact, mode, A, b)
            assert O.count_not_close(x, one, 1e-12) == 0
    assert U.linearSolveMany_(fact, U.UmfpackNormal, A, []) == []
    xs = U.linearSolve(A, bs)  # the reference's API: factor once, every right-hand side
    for b, x in zip(bs, xs):
        assert O.count_not_close(x, U.linearSolve_(fact, U.UmfpackNormal, A, b), 1e-12) == 0


@pytest.mark.parametrize("method", ["band", "mf"])
@pytest.mark.parametrize("k,zero_cols", [(11, (0, 3, 4, 9)), (16, (1,)), (9, (0, 1, 2, 3, 4, 5, 6, 8)), (3, (2,))])
def test_refinement_takes_only_unfinished_columns_through_the_factors(gpu, pkg, O, method, k, zero_cols, monkeypatch):
    """a batch in which some columns are done after the first solve (zero right-hand sides: backward error 0) while the
    others take refinement steps: those still refining are gathered side by side for the step (one, a part of a group of
    eight, more than a group) and must come out exactly as when each is solved alone; the finished ones stay untouched"""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", method)
    m = 40
    n = m * m
    rp, ci, v = O.gen_poisson2d_csr(m)
    v = v.copy()
    v[ci == np.repeat(np.arange(n), np.diff(rp))] -= 1.37  # indefinite shift: the first solve is not at rounding level
    A = pkg.Matrix(n, n, rp, ci, v)
    S = sp.csr_matrix((v, ci, rp), shape=(n, n))
    rng = np.random.default_rng(100 * k + len(zero_cols))
    bs = [np.zeros(n) if c in zero_cols else rng.normal(size=n) * 10.0 ** rng.integers(-3, 4) for c in range(k)]
    U = pkg.umfpack
    fact = U.factor(A, U.analyze(A))
    for mode in (U.UmfpackNormal, U.UmfpackTrans):
        many = U.linearSolveMany_(fact, mode, A, bs)
        for c, (b, x) in enumerate(zip(bs, many)):
            if c in zero_cols:
                assert not np.any(x)
                continue
            one = U.linearSolve_(fact, mode, A, b)
            # (an indefinite matrix, and kernels for 8 columns that add in another order than those for one: agreement
            # to the conditioning, backward errors at rounding level for both)
            assert O.count_not_close(x, one, 1e-9) == 0
            op = S if mode == U.UmfpackNormal else S.T
            assert np.max(np.abs(op @ x - b)) / (np.max(np.abs(b)) + 6 * np.max(np.abs(x))) < 1e-13


# ---- multifrontal path (SPL_LU_METHOD=mf forces it on small matrices; large meshes take it by themselves)
def _grid_matrix(pkg, O, kind, m):
    if kind == "2d":
        rp, ci, v = O.gen_poisson2d_csr(m)
        n = m * m
    else:
        rp, ci, v = O.gen_poisson3d_csr(m)
        n = m ** 3
    return n, pkg.Matrix(n, n, rp, ci, v)


@pytest.mark.parametrize("kind,m", [("3d", 20), ("2d", 90)])
def test_blocked_diagonal_blocks_agree_with_the_unblocked_form(gpu, pkg, O, kind, m, monkeypatch):
    """round 4: the 64 x 64 diagonal blocks of the fronts (and of the band) are factored by panels of 16 and inverted by
    the block recurrence on the matrix cores; SPL_LU_DIAG=plain keeps the unblocked form of rounds 1 - 3.  Different
    associations of the same sums: the solutions agree to rounding level, each with a backward error below eps."""
    n, A = _grid_matrix(pkg, O, kind, m)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    xs = np.random.default_rng(m).uniform(0.5, 1.5, n)
    b = np.asarray(S @ xs).ravel()
    sols = {}
    for method in ("mf", "band"):
        monkeypatch.setenv("SPL_LU_METHOD", method)
        for form in ("blocked", "plain"):
            if form == "plain":
                monkeypatch.setenv("SPL_LU_DIAG", "plain")
            else:
                monkeypatch.delenv("SPL_LU_DIAG", raising=False)
            fact = U.factor(A, U.analyze(A))
            sols[method, form] = U.linearSolve_(fact, U.UmfpackNormal, A, b)
            assert fact.solve_report["backward_error"] < 2.3e-16
        assert np.max(np.abs(sols[method, "blocked"] - sols[method, "plain"]) / np.abs(sols[method, "plain"])) < 1e-12
    assert np.max(np.abs(sols["mf", "blocked"] - xs) / xs) < 1e-11


@pytest.mark.parametrize("unsym", [False, True])
def test_pipelined_steps_and_split_boundary_product_give_the_same_solution(gpu, pkg, O, unsym, monkeypatch):
    """round 4, csrc/multifrontal.hip: (1) the super-block steps of the large fronts as a software pipeline over launches
    (lead groups of step k beside the bulk row updates of step k - 1, the next super block's update handed over in a
    carry array) subtract the same numbers in the same order as one launch per step: the solutions are the same bits
    (SPL_MF_PIPE=0); (2) the boundary rows' share of the forward pass as a streaming product of its own sums the same
    products in another association than the steps over all rows did (SPL_MF_SPLIT_FWD=0): the same solution to
    rounding level.  A 3-D mesh whose top fronts take several super-block steps, symmetric (L D L^T) and unsymmetric
    (both systems: the transposed kernels too); the solve report says two walks each."""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_MF_BIGSOLVE", "64")  # many-workgroup solves from 64 rows on: more fronts, more steps
    monkeypatch.setenv("SPL_MF_CHAIN", "0")  # the substitution steps themselves (round 5: chains take their place by default)
    m = 30
    n, A = _grid_matrix(pkg, O, "3d", m)
    rng = np.random.default_rng(3)
    if unsym:
        S0 = csc_tuple_to_scipy(mat_to_tuple(A)).tocoo()
        v = S0.data * rng.uniform(0.8, 1.2, S0.nnz)
        v[S0.row == S0.col] = 7.0  # dominant: no interchanges, plain LU on the tree
        S1 = sp.csc_matrix((v, (S0.row, S0.col)), shape=S0.shape)
        S1.sort_indices()
        A = pkg.Matrix(n, n, S1.indptr.astype(np.int32), S1.indices.astype(np.int32), S1.data)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    xs = rng.uniform(0.5, 1.5, n)
    fact = U.factor(A, U.analyze(A))
    assert fact.path in (3, 4)
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
        b = np.asarray(op @ xs).ravel()
        x = U.linearSolve_(fact, mode, A, b)
        rep = fact.solve_report
        assert rep["walks"] == 1 + rep["ir_attempted"] <= 3 and rep["backward_error"] < 2.3e-16
        assert rep["walk_bytes"] > 16 * n
        monkeypatch.setenv("SPL_MF_PIPE", "0")
        x_plain = U.linearSolve_(fact, mode, A, b)
        monkeypatch.setenv("SPL_MF_SPLIT_FWD", "0")
        x_old = U.linearSolve_(fact, mode, A, b)
        monkeypatch.delenv("SPL_MF_PIPE")
        monkeypatch.delenv("SPL_MF_SPLIT_FWD")
        assert np.array_equal(x, x_plain)
        assert np.max(np.abs(x - x_old) / np.abs(x_old)) < 1e-13
        assert np.max(np.abs(x - xs) / xs) < 1e-12


@pytest.mark.parametrize("kind", ["symmetric", "unsymmetric", "pivoted", "complex"])
@pytest.mark.parametrize("span", ["512", "256", "1024"])
def test_chains_of_matrix_vector_products_give_the_same_solution(gpu, pkg, O, kind, span, monkeypatch):
    """round 5, csrc/mf_chain.hpp: the pivot blocks of the large fronts as chains of matrix-vector products (explicit
    inverses T_k of blocks of 512 pivots and M_k = T_k L_{k,k-1}, built by the first solve of A x = b on the matrix cores)
    against the substitution steps they replace (SPL_MF_CHAIN=0): the same solution to rounding level, backward errors
    below eps, for L D L^T fronts, plain LU fronts, fronts with interchanges inside their 64 x 64 blocks and native
    complex fronts; blocks of 256 and of 1 024 pivots as well.  A 3-D mesh whose root front has 1 296 pivots (three blocks, the last
    one partial) above levels of fronts with one block; the report says what the chains take."""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    m = 36
    n, A = _grid_matrix(pkg, O, "3d", m)
    rng = np.random.default_rng(11)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    if kind in ("unsymmetric", "pivoted"):
        S0 = S.tocoo()
        v = S0.data * rng.uniform(0.8, 1.2, S0.nnz)
        # dominant: plain LU without interchanges; "pivoted": a weak diagonal, threshold pivoting inside the blocks
        v[S0.row == S0.col] = 7.0 if kind == "unsymmetric" else rng.uniform(0.5, 1.5, n)
        S = sp.csc_matrix((v, (S0.row, S0.col)), shape=S0.shape)
    elif kind == "complex":
        monkeypatch.setenv("SPL_ZI_NATIVE", "1")
        S = sp.csc_matrix((3.0 + 0.5j) * sp.identity(n) - S)
    S.sort_indices()
    A = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    U = pkg.umfpack
    xs = rng.uniform(0.5, 1.5, n) + (1j * rng.uniform(-1, 1, n) if kind == "complex" else 0.0)
    b = np.asarray(S @ xs).ravel()
    an = U.analyze(A)
    monkeypatch.setenv("SPL_MF_CHAIN", "0")
    f0 = U.factor(A, an)
    x0 = U.linearSolve_(f0, U.UmfpackNormal, A, b)
    r0 = f0.solve_report
    assert f0.path in (3, 4) and r0["chain_span"] == 0 and r0["chain_bytes"] == 0
    St = sp.csc_matrix(S.conj().T)
    bt = np.asarray(St @ xs).ravel()
    xt0 = U.linearSolve_(f0, U.UmfpackTrans, A, bt)  # (the switch is read when a set of chains would be built)
    assert f0.solve_report["chain_bytes"] == 0
    monkeypatch.setenv("SPL_MF_CHAIN", span)
    f1 = U.factor(A, an)
    x1 = U.linearSolve_(f1, U.UmfpackNormal, A, b)
    r1 = f1.solve_report
    assert f1.path == f0.path and f1.stats == f0.stats
    if kind == "complex":
        assert f1.stats["complex_fronts"] == 1
    assert r1["chain_span"] == int(span) and r1["chain_bytes"] > 0 and r1["chain_build_ms"] > 0
    assert r1["walk_bytes"] == r0["walk_bytes"]
    lim = 1e-13 if kind == "pivoted" else 2.3e-16  # (block pivoting is a speculation checked against 1e-13)
    assert r0["backward_error"] <= lim and r1["backward_error"] <= lim
    assert np.max(np.abs(x1 - x0)) <= 1e-12 * np.max(np.abs(x0))
    assert np.max(np.abs(x1 - xs) / np.abs(xs)) < 1e-10
    # the chains belong to the factors: a second solve reuses them; the transposed system (A^H x = b for complex
    # matrices) builds a set of its own at its first solve — the same construction on U^H and L^H
    x2 = U.linearSolve_(f1, U.UmfpackNormal, A, b)
    assert np.array_equal(x1, x2) and f1.solve_report["chain_build_ms"] == r1["chain_build_ms"]
    xt1 = U.linearSolve_(f1, U.UmfpackTrans, A, bt)
    rt = f1.solve_report
    assert rt["backward_error"] <= lim
    assert np.max(np.abs(xt1 - xt0)) <= 1e-12 * np.max(np.abs(xt0))
    assert np.max(np.abs(xt1 - xs) / np.abs(xs)) < 1e-10
    if kind in ("unsymmetric", "pivoted"):  # (symmetric factors serve A^T x = b with the untransposed kernels)
        assert rt["chain_bytes"] == 2 * r1["chain_bytes"] and rt["chain_build_ms"] > r1["chain_build_ms"]


@pytest.mark.parametrize("kind", ["symmetric", "unsymmetric", "complex"])
def test_chains_with_eight_and_sixteen_columns_give_the_same_solutions(gpu, pkg, O, kind, monkeypatch):
    """round 5, chain_lead_multi (csrc/mf_chain.hpp): batched solves take 8 real columns (16 for 8 complex ones) through
    the tree together; their pivot-block steps are the same matrix-vector products with u staged through LDS a segment
    of 512 columns at a time.  SPL_MF_CHAIN_MULTI=0 keeps the substitution steps for them (the chains then serve single
    right-hand sides only): the same solutions to rounding level, column by column, backward errors below eps; 11
    columns (two groups, the second padded); host and device right-hand sides."""
    import scipy.sparse as sp
    import torch
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    m = 36
    n, A = _grid_matrix(pkg, O, "3d", m)
    rng = np.random.default_rng(13)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    if kind == "unsymmetric":
        S0 = S.tocoo()
        v = S0.data * rng.uniform(0.8, 1.2, S0.nnz)
        v[S0.row == S0.col] = 7.0
        S = sp.csc_matrix((v, (S0.row, S0.col)), shape=S0.shape)
    elif kind == "complex":
        monkeypatch.setenv("SPL_ZI_NATIVE", "1")
        S = sp.csc_matrix((3.0 + 0.5j) * sp.identity(n) - S)
    S.sort_indices()
    A = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    U = pkg.umfpack
    k = 11
    xs = [rng.uniform(0.5, 1.5, n) + (1j * rng.uniform(-1, 1, n) if kind == "complex" else 0.0) for _ in range(k)]
    bs = [np.asarray(S @ x).ravel() for x in xs]
    an = U.analyze(A)
    monkeypatch.setenv("SPL_MF_CHAIN_MULTI", "0")
    f0 = U.factor(A, an)
    X0 = U.linearSolveMany_(f0, U.UmfpackNormal, A, bs)
    assert f0.solve_report["chain_span"] == 512 and f0.solve_report["backward_error"] <= 2.3e-16
    St = sp.csc_matrix(S.conj().T)
    bts = [np.asarray(St @ x).ravel() for x in xs]
    T0 = U.linearSolveMany_(f0, U.UmfpackTrans, A, bts)
    monkeypatch.delenv("SPL_MF_CHAIN_MULTI")
    f1 = U.factor(A, an)
    X1 = U.linearSolveMany_(f1, U.UmfpackNormal, A, bs)
    assert f1.solve_report["chain_span"] == 512 and f1.solve_report["backward_error"] <= 2.3e-16
    Xd = U.linearSolveManyDevice_(f1, U.UmfpackNormal, A, torch.from_numpy(np.stack(bs)).cuda()).cpu().numpy()
    one = U.linearSolve_(f1, U.UmfpackNormal, A, bs[3])
    for j in range(k):
        assert np.max(np.abs(X1[j] - X0[j])) <= 1e-12 * np.max(np.abs(X0[j]))
        assert np.max(np.abs(X1[j] - xs[j]) / np.abs(xs[j])) < 1e-10
        assert np.array_equal(Xd[j], X1[j])
    assert np.max(np.abs(one - X1[3])) <= 1e-12 * np.max(np.abs(one))
    # ... and the transposed systems (their own chains, the transposed products of the bulk groups 256 columns at a time)
    T1 = U.linearSolveMany_(f1, U.UmfpackTrans, A, bts)
    assert f1.solve_report["backward_error"] <= 2.3e-16
    for j in range(k):
        assert np.max(np.abs(T1[j] - T0[j])) <= 1e-12 * np.max(np.abs(T0[j]))
        assert np.max(np.abs(T1[j] - xs[j]) / np.abs(xs[j])) < 1e-10


@pytest.mark.parametrize("unsym", [False, True])
def test_pair_chain_beside_the_bulk_pass_gives_the_same_solution(gpu, pkg, O, unsym, monkeypatch):
    """round 4, factor_loop (csrc/dense_lu_kernels.hpp): in large fronts the panel solves of the next pair of block steps
    run beside the bulk pass of the pair before (an L-shaped pass brings the next block column / row up to date, the
    second one receives three panels at once in the following pair).  SPL_LU_LOOKAHEAD=1 sends every front of at least
    five blocks through that code (the product starts at 80 tiles beyond the next pair), =0 switches it off: the same
    products in other groupings of K — solutions agree to rounding level, backward errors below eps; symmetric
    (L D L^T) and unsymmetric fronts, both systems."""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_MF_MIDMAX", "256")  # fronts above 256 rows take the per-front pipeline
    m = 34
    n, A = _grid_matrix(pkg, O, "3d", m)
    rng = np.random.default_rng(5)
    if unsym:
        S0 = csc_tuple_to_scipy(mat_to_tuple(A)).tocoo()
        v = S0.data * rng.uniform(0.8, 1.2, S0.nnz)
        v[S0.row == S0.col] = 7.0
        S1 = sp.csc_matrix((v, (S0.row, S0.col)), shape=S0.shape)
        S1.sort_indices()
        A = pkg.Matrix(n, n, S1.indptr.astype(np.int32), S1.indices.astype(np.int32), S1.data)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    xs = rng.uniform(0.5, 1.5, n)
    an = U.analyze(A)
    sols = {}
    for ahead in ("1", "0"):
        monkeypatch.setenv("SPL_LU_LOOKAHEAD", ahead)
        fact = U.factor(A, an)
        assert fact.path in (3, 4)
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
            b = np.asarray(op @ xs).ravel()
            sols[ahead, mode] = U.linearSolve_(fact, mode, A, b)
            assert fact.solve_report["backward_error"] < 2.3e-16
    for mode in (U.UmfpackNormal, U.UmfpackTrans):
        assert np.max(np.abs(sols["1", mode] - sols["0", mode]) / np.abs(sols["0", mode])) < 1e-12
        assert np.max(np.abs(sols["1", mode] - xs) / xs) < 1e-11


@pytest.mark.parametrize("kind,m", [("3d", 24), ("2d", 170), ("2d", 100)])
def test_paired_steps_of_the_lockstep_fronts_give_the_same_solution(gpu, pkg, O, kind, m, monkeypatch):
    """round 4, mid_update_kernel (csrc/multifrontal.hip): the medium fronts of a level advance in lockstep, and two
    FULL pivot blocks in a row are taken as a pair — an L-shaped pass after the first, one pass over the rest of the
    window with both panels (K = 128) after the second; a short second block does not pair (its L-shaped tiles would
    reach into that window: the first version of this did, and solved nothing).  SPL_MF_MIDPAIR=0: one pass per block
    as before.  Same products in other groupings: solutions agree to rounding level.  Limits lowered so that the top
    separators of these small meshes (100 - 580 pivots: one to nine blocks, the last one short) are medium fronts."""
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_MF_SMALL", "64")
    monkeypatch.setenv("SPL_MF_MIDMAX", "4096")
    n, A = _grid_matrix(pkg, O, kind, m)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    xs = np.random.default_rng(m).uniform(0.5, 1.5, n)
    b = np.asarray(S @ xs).ravel()
    an = U.analyze(A)
    sols = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("SPL_MF_MIDPAIR", pair)
        fact = U.factor(A, an)
        sols[pair] = U.linearSolve_(fact, U.UmfpackNormal, A, b)
        assert fact.solve_report["backward_error"] < 2.3e-16
    assert np.max(np.abs(sols["1"] - sols["0"]) / np.abs(sols["0"])) < 1e-12
    assert np.max(np.abs(sols["1"] - xs) / xs) < 1e-11


@pytest.mark.parametrize("cut", [None, "2"])
@pytest.mark.parametrize("kind,m", [("3d", 26), ("2d", 150)])
def test_assembly_beside_the_factorisation_gives_the_same_factors(gpu, pkg, O, kind, m, cut, monkeypatch):
    """round 4, csrc/multifrontal.hip: while a level is factored a second stream moves the panels of the level below to
    the arena, zeroes the freed region and scatters the entries of A of the level above into it; only the extend-add
    stays between the levels.  The same kernels on the same data in another interleaving: the solution has the same bits
    as with everything on one stream (SPL_MF_OVERLAP=0), also when the tree is cut into subtrees (SPL_MF_CUT), with the
    size limits lowered so that a small tree runs every class of front."""
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_MF_SMALL", "64")
    monkeypatch.setenv("SPL_MF_MIDMAX", "512")
    if cut:
        monkeypatch.setenv("SPL_MF_CUT", cut)
    n, A = _grid_matrix(pkg, O, kind, m)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    xs = np.random.default_rng(m).uniform(0.5, 1.5, n)
    b = np.asarray(S @ xs).ravel()
    an = U.analyze(A)
    x = [U.linearSolve_(U.factor(A, an), U.UmfpackNormal, A, b) for _ in range(2)]
    monkeypatch.setenv("SPL_MF_OVERLAP", "0")
    x_one_stream = U.linearSolve_(U.factor(A, an), U.UmfpackNormal, A, b)
    assert np.array_equal(x[0], x_one_stream) and np.array_equal(x[1], x_one_stream)
    assert np.max(np.abs(x[0] - xs) / xs) < 1e-11


@pytest.mark.parametrize("limits", ["default", "small"])
@pytest.mark.parametrize("kind,m", [("2d", 7), ("2d", 45), ("2d", 130), ("3d", 9), ("3d", 22)])
def test_multifrontal_matches_band_and_oracle(gpu, pkg, O, kind, m, limits, monkeypatch):
    """nested-dissection multifrontal factors == band factors == the CPU oracle, both systems,
    several right-hand sides at once; sizes from a single front to thousands.  "small" lowers the
    size limits of the front classes so that these small trees also run the code of the large ones:
    per-front multi-launch factorisation above 512 (lockstep from 64 up to there) and many-workgroup
    solves above 64"""
    if limits == "small":
        monkeypatch.setenv("SPL_MF_SMALL", "64")
        monkeypatch.setenv("SPL_MF_MIDMAX", "512")
        monkeypatch.setenv("SPL_MF_BIGSOLVE", "64")
    n, A = _grid_matrix(pkg, O, kind, m)
    U = pkg.umfpack
    rng = np.random.default_rng(m)
    bs = [rng.uniform(0.5, 1.5, n) for _ in range(3)]
    monkeypatch.setenv("SPL_LU_METHOD", "band")
    fb = U.factor(A, U.analyze(A))
    assert fb.path == 1
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    fm = U.factor(A, U.analyze(A))
    assert fm.path == 3
    sb, sm = fb.stats, fm.stats
    assert (sb["path"], sb["n"], sb["fronts"]) == (1, n, 0) and (sm["path"], sm["n"]) == (3, n)
    assert 0 < sb["kl"] == sb["ku"] <= (m if kind == "2d" else m * m)  # symmetric pattern, RCM no worse than natural
    assert sb["flops"] == 2.0 * n * sb["kl"] * sb["ku"] and sb["device_bytes"] >= 8.0 * n * (2 * sb["kl"] + 1)
    assert sm["fronts"] >= 1 and 0 < sm["flops"] and sm["device_bytes"] > 0
    for mode in (U.UmfpackNormal, U.UmfpackTrans):
        xb = U.linearSolveMany_(fb, mode, A, bs)
        xm = U.linearSolveMany_(fm, mode, A, bs)
        for p, q in zip(xb, xm):
            assert O.count_not_close(p, q, 1e-10) == 0
    if n <= 3000:
        xo, _ = O.linear_solve(mat_to_tuple(A), bs[0])
        assert O.count_not_close(U.linearSolve_(fm, U.UmfpackNormal, A, bs[0]), xo, 1e-10) == 0


def test_multifrontal_unsymmetric_values_and_fallback(gpu, pkg, O, monkeypatch):
    """unsymmetric values on a mesh pattern (the tree is built on A + A^T); then a matrix whose
    no-interchange factors are useless: the speculation is replaced by band partial pivoting"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    U = pkg.umfpack
    rng = np.random.default_rng(41)
    m = 40
    n = m * m
    rp, ci, v = O.gen_poisson2d_csr(m)
    v = v * rng.uniform(0.5, 1.0, len(v))
    diag = ci == np.repeat(np.arange(n), np.diff(rp))
    v[diag] = 4.5  # column sums of |off-diagonal| <= 4: dominant, unsymmetric
    A = pkg.Matrix(n, n, rp, ci, v)
    S = sp.csc_matrix((v, ci, rp), shape=(n, n))
    xs = rng.uniform(0.5, 1.5, n)
    fact = U.factor(A, U.analyze(A))
    assert fact.path == 3
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, S.T.tocsc())):
        x = U.linearSolve_(fact, mode, A, op @ xs)
        assert O.count_not_close(x, xs, 1e-10) == 0
    # 2x2 blocks [[1e-14, 3], [3, 1e-14]] + weak coupling: needs interchanges
    off = np.zeros(n - 1)
    off[0::2] = 3.0
    B = sp.diags([off, np.full(n, 1e-14), off, rng.uniform(-0.1, 0.1, n - m)], (-1, 0, 1, m), format="csc")
    B.sort_indices()
    M = pkg.Matrix(n, n, B.indptr, B.indices, B.data)
    # round 3: threshold pivoting inside the diagonal blocks of the fronts takes the 3s by itself — the speculation holds
    fact = U.factor(M, U.analyze(M))
    assert fact.path == 4 and fact.stats["block_pivoting"] == 1
    for mode, op in ((U.UmfpackNormal, B), (U.UmfpackTrans, B.T.tocsc())):
        x = U.linearSolve_(fact, mode, M, op @ xs)
        assert fact.path == 4 and np.max(np.abs(x - xs)) / np.max(np.abs(xs)) < 1e-10
    many = U.linearSolveMany_(fact, U.UmfpackNormal, M, [B @ xs, 2.0 * (B @ xs)])
    assert np.max(np.abs(many[1] - 2.0 * xs)) / np.max(np.abs(xs)) < 1e-10
    # without it (SPL_LU_BLOCK_PIVOT=0), the stages of round 2:
    monkeypatch.setenv("SPL_LU_BLOCK_PIVOT", "0")
    fact = U.factor(M, U.analyze(M))
    assert fact.path == 4 and fact.stats["block_pivoting"] == 0
    x = U.linearSolve_(fact, U.UmfpackNormal, M, B @ xs)
    assert fact.path == 5  # static pivoting: the transversal takes the 3s, the factors stay on the tree
    assert np.max(np.abs(x - xs)) / np.max(np.abs(xs)) < 1e-10
    xt = U.linearSolve_(fact, U.UmfpackTrans, M, B.T @ xs)
    assert fact.path == 5 and np.max(np.abs(xt - xs)) / np.max(np.abs(xs)) < 1e-10
    many = U.linearSolveMany_(fact, U.UmfpackNormal, M, [B @ xs, 2.0 * (B @ xs)])
    assert np.max(np.abs(many[1] - 2.0 * xs)) / np.max(np.abs(xs)) < 1e-10
    # without that stage (SPL_LU_STATIC_PIVOT=0): band factorisation with partial pivoting, as before
    monkeypatch.setenv("SPL_LU_STATIC_PIVOT", "0")
    fact = U.factor(M, U.analyze(M))
    assert fact.path == 4
    x = U.linearSolve_(fact, U.UmfpackNormal, M, B @ xs)
    assert fact.path == 0
    assert np.max(np.abs(x - xs)) / np.max(np.abs(xs)) < 1e-8
    assert np.max(np.abs(x - spla.spsolve(B, B @ xs))) / np.max(np.abs(xs)) < 1e-8


def test_multifrontal_random_patterns_against_oracle(gpu, pkg, O, monkeypatch):
    """random unsymmetric sparse matrices made diagonally dominant (isolated vertices, several
    components, dense rows): the multifrontal path against the CPU oracle's LU, both systems"""
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    U = pkg.umfpack
    rng = np.random.default_rng(77)
    for n, k in ((1, 0), (2, 1), (60, 150), (333, 900), (700, 5000), (1500, 4000)):
        rows = rng.integers(0, n, k)
        cols = rng.integers(0, n, k)
        if n >= 300:  # a dense-ish row and column
            rows = np.concatenate([rows, np.full(n // 3, 7), rng.integers(0, n, n // 3)])
            cols = np.concatenate([cols, rng.integers(0, n, n // 3), np.full(n // 3, 11)])
        vals = rng.normal(size=len(rows))
        off = rows != cols
        A0 = O.compress(n, n, rows[off], cols[off], vals[off])
        colsum = np.zeros(n)
        np.add.at(colsum, np.repeat(np.arange(n), np.diff(A0[2])), np.abs(A0[4]))
        A = O.compress(n, n, np.concatenate([A0[3], np.arange(n)]),
                       np.concatenate([np.repeat(np.arange(n), np.diff(A0[2])), np.arange(n)]),
                       np.concatenate([A0[4], colsum + 1.0]))
        M = tuple_to_mat(pkg, A)
        fact = U.factor(M, U.analyze(M))
        assert fact.path == 3
        xs = rng.uniform(0.5, 1.5, n)
        S = csc_tuple_to_scipy(A)
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, S.T.tocsc())):
            b = np.asarray(op @ xs).ravel()
            x = U.linearSolve_(fact, mode, M, b)
            assert O.count_not_close(x, xs, 1e-10) == 0, (n, mode)
        xo, _ = O.linear_solve(A, np.asarray(S @ xs).ravel())
        assert O.count_not_close(U.linearSolve_(fact, U.UmfpackNormal, M, np.asarray(S @ xs).ravel()), xo, 1e-10) == 0


def test_arrow_matrix_takes_the_tree(gpu, pkg, O):
    """a dense row and column around a sparse rest (n = 30 000): as a band it is n wide, and a nested
    dissection that cannot separate it would make one dense front; the hub becomes a separator of its
    own and the system is solved on the tree in milliseconds"""
    import scipy.sparse as sp
    n = 30000
    rng = np.random.default_rng(11)
    A = sp.diags([-np.ones(n - 1), -np.ones(n - 1)], (-1, 1)).tolil()
    A[0, 1:] = rng.uniform(-1.0, -0.1, n - 1)
    A[1:, 0] = rng.uniform(-1.0, -0.1, (n - 1, 1))
    A = sp.csc_matrix(A)
    A = sp.csc_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=0)).ravel() + 1.0))  # dominant by columns
    A.sort_indices()
    M = pkg.Matrix(n, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data)
    xs = rng.uniform(0.5, 1.5, n)
    fact = pkg.umfpack.factor(M, pkg.umfpack.analyze(M))
    assert fact.path == 3 and fact.stats["device_bytes"] < 1e9
    for mode, op in ((pkg.umfpack.UmfpackNormal, A), (pkg.umfpack.UmfpackTrans, A.T)):
        x = pkg.umfpack.linearSolve_(fact, mode, M, op @ xs)
        assert O.count_not_close(x, xs, 1e-10) == 0


def test_numeric_rejects_other_row_indices(gpu, pkg):
    """same column counts, different row indices: UMFPACK_ERROR_different_pattern (-11), not a scatter with
    a stale ordering (only Ap used to be compared)"""
    import ctypes as C
    U = pkg.umfpack
    L = U._declare()
    ap = np.array([0, 2, 4, 6], dtype=np.int32)
    ai = np.array([0, 1, 1, 2, 0, 2], dtype=np.int32)
    ax = np.array([4.0, 1.0, 4.0, 1.0, 1.0, 4.0])
    sym = C.c_void_p()
    assert L.umfpack_di_symbolic(3, 3, U.p_i32(ap), U.p_i32(ai), U.p_f64(ax), C.byref(sym), None, None) == 0
    num = C.c_void_p()
    ai2 = np.array([0, 2, 0, 1, 1, 2], dtype=np.int32)  # same counts per column, other rows
    assert L.umfpack_di_numeric(U.p_i32(ap), U.p_i32(ai2), U.p_f64(ax), sym, C.byref(num), None, None) == -11
    assert not num.value
    assert L.umfpack_di_numeric(U.p_i32(ap), U.p_i32(ai), U.p_f64(ax), sym, C.byref(num), None, None) == 0
    L.umfpack_di_free_numeric(C.byref(num))
    L.umfpack_di_free_symbolic(C.byref(sym))


def test_zi_numeric_rejects_other_pattern_and_vouching_does_not_leak(gpu, pkg):
    """the complex numeric phase compares the COMPLEX pattern with the one it analysed (-11 on other rows or other
    column counts) and only then lets the embedding skip its own, four times longer comparison; that licence must
    not outlive the call: a real numeric call right after still checks its pattern"""
    import ctypes as C
    U = pkg.umfpack
    L = U._declare()
    ap = np.array([0, 2, 4, 6], dtype=np.int32)
    ai = np.array([0, 1, 1, 2, 0, 2], dtype=np.int32)
    az = np.array([4.0, 1.0, 1.0, 0.5, 4.0, -1.0, 1.0, 0.0, 1.0, 2.0, 4.0, 0.0])  # packed (re, im)
    sym = C.c_void_p()
    assert L.umfpack_zi_symbolic(3, 3, U.p_i32(ap), U.p_i32(ai), U.p_f64(az), None, C.byref(sym), None, None) == 0
    num = C.c_void_p()
    ai2 = np.array([0, 2, 0, 1, 1, 2], dtype=np.int32)
    assert L.umfpack_zi_numeric(U.p_i32(ap), U.p_i32(ai2), U.p_f64(az), None, sym, C.byref(num), None, None) == -11
    assert not num.value
    ap3 = np.array([0, 1, 4, 6], dtype=np.int32)
    assert L.umfpack_zi_numeric(U.p_i32(ap3), U.p_i32(ai), U.p_f64(az), None, sym, C.byref(num), None, None) == -11
    assert L.umfpack_zi_numeric(U.p_i32(ap), U.p_i32(ai), U.p_f64(az), None, sym, C.byref(num), None, None) == 0
    L.umfpack_zi_free_numeric(C.byref(num))
    L.umfpack_zi_free_symbolic(C.byref(sym))
    # the real path right after: still checked
    ax = np.array([4.0, 1.0, 4.0, 1.0, 1.0, 4.0])
    assert L.umfpack_di_symbolic(3, 3, U.p_i32(ap), U.p_i32(ai), U.p_f64(ax), C.byref(sym), None, None) == 0
    assert L.umfpack_di_numeric(U.p_i32(ap), U.p_i32(ai2), U.p_f64(ax), sym, C.byref(num), None, None) == -11
    L.umfpack_di_free_symbolic(C.byref(sym))


def test_linear_solve_checks_rhs_length_and_type(gpu, pkg):
    U = pkg.umfpack
    A = pkg.ident(5)
    fact = U.factor(A, U.analyze(A))
    with pytest.raises(U.UmfpackError):
        U.linearSolve_(fact, U.UmfpackNormal, A, np.ones(4))
    Z = pkg.Matrix(5, 5, A.pointers, A.indices, A.values.astype(np.complex128))
    with pytest.raises(U.UmfpackError):
        U.linearSolve_(fact, U.UmfpackNormal, Z, np.ones(5, dtype=np.complex128))


def _backward_error(S, x, b):
    """componentwise backward error max_i |r_i| / (|A| |x| + |b|)_i (what umfpack_di_solve drives below 1e-13)"""
    r = np.abs(S @ x - b)
    den = abs(S) @ np.abs(x) + np.abs(b)
    return float(np.max(r / np.where(den > 0, den, 1.0)))


@pytest.mark.parametrize("m,dim,tiny_diag", [(58, 3, True), (300, 2, True), (58, 3, False)])
def test_static_pivoting_random_unsymmetric_mesh(gpu, pkg, O, m, dim, tiny_diag):
    """a random unsymmetric, non-dominant matrix with a mesh pattern (values uniform in [-1, 1]; 3-D:
    195 112 unknowns).  With a diagonal of 1e-12 the factors without interchanges are useless, the
    static-pivoting stage refactors on the tree (path 5) and both systems are solved to a backward error
    <= 1e-13 — where the band fallback of round 1 did not fit the HBM and solve returned
    UMFPACK_ERROR_out_of_memory.  With a random diagonal the plain speculation may already hold (path 4):
    either way the answer is backward stable."""
    import scipy.sparse as sp
    rng = np.random.default_rng(m)
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    I = sp.identity(m)
    P = (sp.kron(I, T) + sp.kron(T, I)) if dim == 2 else (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I))
    S = sp.csc_matrix(P)
    S.data = rng.uniform(-1.0, 1.0, S.nnz)
    if tiny_diag:
        S.setdiag(1e-12 * rng.uniform(0.5, 1.0, S.shape[0]))
    S = sp.csc_matrix(S)
    S.sort_indices()
    n = S.shape[0]
    M = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    U = pkg.umfpack
    fact = U.factor(M, U.analyze(M))
    xs = rng.uniform(0.5, 1.5, n)
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
        b = np.asarray(op @ xs).ravel()
        x = U.linearSolve_(fact, mode, M, b)
        assert fact.path == 5 if tiny_diag else fact.path in (4, 5), fact.path
        assert _backward_error(op, x, b) <= 1e-13


def test_rows_in_random_order_end_on_static_pivoting(gpu, pkg, monkeypatch):
    """a mesh matrix whose rows arrive in random order: the pattern of A + A^T has no separators, the tree of the
    speculation can grow beyond the device (tools/fuzz_lu_scale.py, family perm2d: 195 364 unknowns used to return
    UMFPACK_ERROR_out_of_memory from umfpack_di_numeric where SuperLU solves).  numeric now hands such a matrix to
    static pivoting, whose transversal undoes the permutation.  The refusal of the speculation is simulated here
    (SPL_LU_TEST_SPECULATION_OOM) so that the test stays small."""
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    m = 280  # large enough for the refinement with the static-pivoting factors to stall: the FGMRES polish takes over
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    P = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocoo()
    v = 10.0 ** rng.uniform(-3, 3, P.nnz) * rng.choice([-1.0, 1.0], P.nnz)
    perm = rng.permutation(m * m)
    S = sp.csc_matrix((v, (perm[P.row], P.col)), shape=(m * m, m * m))
    S.sort_indices()
    n = S.shape[0]
    M = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    U = pkg.umfpack
    monkeypatch.setenv("SPL_LU_TEST_SPECULATION_OOM", "1")
    fact = U.factor(M, U.analyze(M))
    assert fact.path == 5
    xs = rng.uniform(0.5, 1.5, n)
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
        b = np.asarray(op @ xs).ravel()
        x = U.linearSolve_(fact, mode, M, b)
        assert _backward_error(op, x, b) <= 1e-13


def test_stalled_static_pivoting_is_polished_by_fgmres(gpu, pkg, monkeypatch):
    """299 209 unknowns of the perm2d family of tools/fuzz_lu_scale.py (5-point mesh, values 10^U(-3, 3) with random
    signs, rows in random order): the speculation does not fit, static pivoting factors, its refinement stops at a
    backward error of 5e-10 and the pivoted band does not fit — this used to be status -1.  The FGMRES polish
    preconditioned by the factors held takes the backward error to rounding level; without it (SPL_LU_GMRES=0) the
    refined solution is still returned (<= 1e-9)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    m = 547
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
    P = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)) + sp.identity(m * m)).tocoo()
    v = 10.0 ** rng.uniform(-3, 3, P.nnz) * rng.choice([-1.0, 1.0], P.nnz)
    perm = rng.permutation(m * m)
    S = sp.csc_matrix((v, (perm[P.row], P.col)), shape=(m * m, m * m))
    S.sort_indices()
    n = S.shape[0]
    M = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    U = pkg.umfpack
    xs = rng.uniform(0.5, 1.5, n)
    b = np.asarray(S @ xs).ravel()
    fact = U.factor(M, U.analyze(M))
    # (round 4: residuals accumulated in twice the working precision — refinement with them no longer stalls on this
    # matrix; the plainly evaluated residual, SPL_LU_RESIDUAL=plain, reproduces the stall the polish was written for)
    x = U.linearSolve_(fact, U.UmfpackNormal, M, b)
    assert fact.path == 5 and _backward_error(S, x, b) <= 1e-13
    assert fact.solve_report["backward_error"] <= 1e-13
    monkeypatch.setenv("SPL_LU_RESIDUAL", "plain")
    x = U.linearSolve_(fact, U.UmfpackNormal, M, b)
    assert fact.path == 5 and _backward_error(S, x, b) <= 1e-13
    monkeypatch.setenv("SPL_LU_GMRES", "0")
    x0 = U.linearSolve_(fact, U.UmfpackNormal, M, b)
    assert 1e-13 < _backward_error(S, x0, b) <= 1e-9
    # what the object reports is the error of what it delivered, measured its own way
    assert 0.1 * _backward_error(S, x0, b) <= fact.solve_report["backward_error"] <= 10.0 * _backward_error(S, x0, b)
    monkeypatch.delenv("SPL_LU_RESIDUAL")
    x1 = U.linearSolve_(fact, U.UmfpackNormal, M, b)
    assert _backward_error(S, x1, b) <= 1e-9


def test_solve_many_device_pointers_match_host(gpu, pkg, O):
    """spl_umfpack_{di,zi}_solve_many_dev: right-hand sides and solutions in HBM (torch tensors) — the same
    numbers as the host-array entry points, both systems, real and complex"""
    import scipy.sparse as sp
    import torch
    rng = np.random.default_rng(77)
    m = 24
    T = sp.diags([-np.ones(m - 1), 2.5 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    S = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)))
    S.data = S.data * rng.uniform(0.8, 1.2, S.nnz)
    S.sort_indices()
    n = S.shape[0]
    U = pkg.umfpack
    for cplx in (False, True):
        vals = S.data + (0.3j * rng.uniform(-1, 1, S.nnz) if cplx else 0.0)
        M = pkg.Matrix(n, n, S.indptr, S.indices, vals)
        fact = U.factor(M, U.analyze(M))
        B = rng.normal(size=(5, n)) + (1j * rng.normal(size=(5, n)) if cplx else 0.0)
        for mode in (U.UmfpackNormal, U.UmfpackTrans):
            host = U.linearSolveMany_(fact, mode, M, [B[c] for c in range(5)])
            devX = U.linearSolveManyDevice_(fact, mode, M, torch.from_numpy(np.ascontiguousarray(B)).cuda())
            assert devX.is_cuda and devX.shape == (5, n)
            assert np.array_equal(devX.cpu().numpy(), np.stack(host))
    with pytest.raises(U.UmfpackError):
        U.linearSolveManyDevice_(fact, U.UmfpackNormal, M, torch.zeros((2, n + 1), dtype=torch.complex128, device="cuda"))
    with pytest.raises(U.UmfpackError):
        U.linearSolveManyDevice_(fact, U.UmfpackNormal, M, torch.zeros((2, n), dtype=torch.float64, device="cuda"))


def test_static_pivoting_complex_mesh(gpu, pkg):
    """the `zi` path with a useless diagonal: the solve refactors the EMBEDDING the object holds (its own device
    copy, not the caller's complex arrays) with static pivoting and stays on the tree; both systems, packed and
    device-pointer right-hand sides"""
    import scipy.sparse as sp
    import torch
    rng = np.random.default_rng(5)
    m = 40
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    S = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)), dtype=np.complex128)
    S.data = rng.uniform(-1, 1, S.nnz) + 1j * rng.uniform(-1, 1, S.nnz)
    S.setdiag(1e-12 * (rng.uniform(0.5, 1.0, S.shape[0]) + 0j))
    S = sp.csc_matrix(S)
    S.sort_indices()
    n = S.shape[0]
    M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    U = pkg.umfpack
    fact = U.factor(M, U.analyze(M))
    xs = rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(-1, 1, n)
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.conj().T))):
        b = np.asarray(op @ xs).ravel()
        x = U.linearSolve_(fact, mode, M, b)
        assert fact.path in (0, 5), fact.path
        assert _backward_error(op, x, b) <= 1e-12
        xd = U.linearSolveManyDevice_(fact, mode, M, torch.from_numpy(b[None, :].copy()).cuda()).cpu().numpy()[0]
        assert _backward_error(op, xd, b) <= 1e-12


# ---- symmetric matrices: L D L^T on the same fronts (csrc/dense_lu_kernels.hpp Band::sym)
@pytest.mark.parametrize("limits", ["default", "small"])
@pytest.mark.parametrize("kind,m", [("2d", 45), ("2d", 130), ("3d", 22), ("indef", 90)])
def test_symmetric_matrix_is_factored_as_ldlt(gpu, pkg, O, kind, m, limits, monkeypatch):
    """A == A^T exactly: the tree runs in its symmetric mode (U12 = D L21^T instead of a second triangular solve, the
    trailing update only on and below the diagonal, extend-add mirrors the lower triangle) — half the flops in the
    statistics, and the solutions of both systems those of the plain LU of the same fronts (SPL_LU_SYMMETRIC=0) to
    1e-10.  "indef": a shifted Laplacian with eigenvalues of both signs (symmetric indefinite: D has both signs; the
    speculation without interchanges is checked by its backward error as for any matrix).  "small": the size classes
    lowered so that lockstep, multi-launch and many-workgroup code all run on these trees."""
    import scipy.sparse as sp
    if limits == "small":
        monkeypatch.setenv("SPL_MF_SMALL", "64")
        monkeypatch.setenv("SPL_MF_MIDMAX", "512")
        monkeypatch.setenv("SPL_MF_BIGSOLVE", "64")
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    rng = np.random.default_rng(m)
    if kind == "indef":
        T = sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
        W = sp.triu(sp.random(m * m, m * m, density=2.0 / (m * m), random_state=3, data_rvs=lambda k: rng.uniform(-0.3, 0.3, k)), 1)
        S = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)) - 1.2345 * sp.identity(m * m) + W + W.T)
        S.sort_indices()
        n = m * m
        A = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    else:
        n, A = _grid_matrix(pkg, O, kind, m)
        S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    bs = [rng.uniform(0.5, 1.5, n) for _ in range(2)]
    fs = U.factor(A, U.analyze(A))
    monkeypatch.setenv("SPL_LU_SYMMETRIC", "0")
    fl = U.factor(A, U.analyze(A))
    monkeypatch.delenv("SPL_LU_SYMMETRIC")
    assert fs.path == fl.path == (4 if kind == "indef" else 3)  # 4: not diagonally dominant, the factors are a speculation
    assert fs.stats["flops"] == 0.5 * fl.stats["flops"] and fs.stats["fronts"] == fl.stats["fronts"]
    for mode in (U.UmfpackNormal, U.UmfpackTrans):
        xs = U.linearSolveMany_(fs, mode, A, bs)
        xl = U.linearSolveMany_(fl, mode, A, bs)
        for p, q, b in zip(xs, xl, bs):
            assert O.count_not_close(p, q, 1e-10) == 0
            assert _backward_error(S, p, b) <= 1e-13


def test_almost_symmetric_matrix_takes_plain_lu(gpu, pkg, O, monkeypatch):
    """one value off by one ulp, or one entry present on one side only: not symmetric, the plain LU runs (full flops)
    and solves both systems"""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    m = 40
    n, A0 = _grid_matrix(pkg, O, "2d", m)
    S0 = csc_tuple_to_scipy(mat_to_tuple(A0)).tocsc()
    U = pkg.umfpack
    full = None
    for variant in ("sym", "ulp", "pattern"):
        S = S0.copy()
        if variant == "ulp":
            k = S.indptr[7]  # entry (6, 7): off the diagonal
            assert S.indices[k] == 6
            S.data[k] = np.nextafter(S.data[k], 0.0)
        elif variant == "pattern":
            S = sp.csc_matrix(S + sp.csc_matrix(([0.125], ([3], [n - 5])), shape=(n, n)))
        S.sort_indices()
        A = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
        f = U.factor(A, U.analyze(A))
        assert f.path == 3
        monkeypatch.setenv("SPL_LU_SYMMETRIC", "0")
        full = U.factor(A, U.analyze(A)).stats["flops"]  # the same tree as plain LU
        monkeypatch.delenv("SPL_LU_SYMMETRIC")
        assert f.stats["flops"] == (0.5 * full if variant == "sym" else full)
        xs = np.random.default_rng(5).uniform(0.5, 1.5, n)
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
            b = np.asarray(op @ xs).ravel()
            assert _backward_error(op, U.linearSolve_(f, mode, A, b), b) <= 1e-13


def test_band_ordering_is_computed_on_demand(gpu, pkg, O, monkeypatch):
    """from 1e5 unknowns on, a structurally symmetric pattern whose tree wins against the best band its level structure
    allows is analysed without a band ordering (csrc/umfpack.hip, symbolic_common); a factorisation that needs the
    band after all — here forced: partial pivoting — computes reverse Cuthill-McKee then, from the pattern the object
    holds.  2-D Poisson 320^2 = 102 400 unknowns."""
    m = 320
    n, A = _grid_matrix(pkg, O, "2d", m)
    U = pkg.umfpack
    an = U.analyze(A)
    f = U.factor(A, an)
    assert f.path == 3 and f.stats["kl"] == 0
    monkeypatch.setenv("SPL_LU_FORCE_PIVOT", "1")
    fb = U.factor(A, an)  # the same analysis object
    st = fb.stats
    assert st["path"] == 0 and 0 < st["kl"] == st["ku"] <= m
    xs = np.random.default_rng(2).uniform(0.5, 1.5, n)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    b = np.asarray(S @ xs).ravel()
    for fac in (f, fb):
        assert _backward_error(S, U.linearSolve_(fac, U.UmfpackNormal, A, b), b) <= 1e-13


def test_analysis_thread_uses_the_callers_device(gpu, pkg, O, monkeypatch):
    """the nested dissection runs on a thread of its own, and HIP's current device is per thread (0 in a new
    one) — the level service must follow the device the CALLER selected.  Needs two devices: with device 1 current, the
    slab of the service (at least 1 GiB, kept by the library's pool afterwards) must not appear on device 0."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_ND_GPU_MIN", "2000")
    n, A = _grid_matrix(pkg, O, "3d", 36)
    U = pkg.umfpack
    pkg._ffi.release_cached_memory()
    prev = torch.cuda.current_device()
    try:
        torch.cuda.set_device(1)
        free0 = torch.cuda.mem_get_info(0)[0]
        free1 = torch.cuda.mem_get_info(1)[0]
        an = U.analyze(A)
        assert free0 - torch.cuda.mem_get_info(0)[0] < (1 << 29), "the analysis allocated on device 0"
        assert free1 - torch.cuda.mem_get_info(1)[0] >= (1 << 30), "the level service's slab is not on the caller's device"
        del an
    finally:
        torch.cuda.set_device(prev)
        pkg._ffi.release_cached_memory()


@pytest.mark.parametrize("kind", ["3d", "2d", "two_components", "unsymmetric_pattern"])
def test_level_structures_on_the_gpu(gpu, pkg, O, kind, monkeypatch):
    """the nested dissection's level structures of large regions built on the device (csrc/nd_levels.hip; by default
    from 10^6 vertices on, here from 2 000): the tree is a valid one (the factorisation solves both systems), of the
    host's quality (other roots, other cuts: flops within 25 %; 2 % fewer at config C5), and does not depend on the order the device's atomics produce (two analyses:
    the same flops, the same solution bit for bit).  A disconnected graph falls back to the host traversal."""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    rng = np.random.default_rng(8)
    if kind == "3d":
        n, A = _grid_matrix(pkg, O, "3d", 36)
    elif kind == "2d":
        n, A = _grid_matrix(pkg, O, "2d", 260)
    else:
        m = 150
        T = sp.diags([-np.ones(m - 1), 4.0 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
        K = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(sp.diags([-np.ones(m - 1), -np.ones(m - 1)], (-1, 1)), sp.identity(m)))
        if kind == "two_components":
            S0 = sp.block_diag([K, 2.0 * K], format="csc")
        else:  # entries dropped on one side only: the pattern is not symmetric, the graph is that of A + A^T
            K = K.tocoo()
            keep = (K.row <= K.col) | (rng.uniform(size=K.nnz) < 0.7)
            S0 = sp.csc_matrix((K.data[keep], (K.row[keep], K.col[keep])), shape=K.shape)
        S0.sort_indices()
        n = S0.shape[0]
        A = pkg.Matrix(n, n, S0.indptr.astype(np.int32), S0.indices.astype(np.int32), S0.data)
    S = csc_tuple_to_scipy(mat_to_tuple(A))
    U = pkg.umfpack
    xs = rng.uniform(0.5, 1.5, n)
    monkeypatch.setenv("SPL_ND_GPU_MIN", "0")
    host = U.factor(A, U.analyze(A))
    monkeypatch.setenv("SPL_ND_GPU_MIN", "2000")
    results = []
    for rep in range(2):
        # the boundary lists of the fronts are made on the device too (bitmaps over the ancestors' pivots): the first
        # analysis makes them on both sides and fails unless they agree entry for entry
        monkeypatch.setenv("SPL_ND_BOUNDARIES", "check" if rep == 0 else "")
        f = U.factor(A, U.analyze(A))
        assert f.path in (3, 4)
        sols = []
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
            b = np.asarray(op @ xs).ravel()
            x = U.linearSolve_(f, mode, A, b)
            assert _backward_error(op, x, b) <= 1e-13
            sols.append(x)
        results.append((f.stats["flops"], f.stats["fronts"], sols))
    assert results[0][0] == results[1][0] and results[0][1] == results[1][1]
    assert all(np.array_equal(p, q) for p, q in zip(results[0][2], results[1][2]))
    assert results[0][0] <= 1.25 * host.stats["flops"]


@pytest.mark.parametrize("kind", ["arrow", "complex", "3d_host_lists"])
def test_boundary_lists_on_the_gpu(gpu, pkg, O, kind, monkeypatch):
    """csrc/nd_levels.hip, boundaries(): trees the mesh cases above do not make — hubs peeled one at a time (an arrow
    matrix: a chain of one-pivot separators), the tree of a complex matrix (ordered on the small graph, expanded
    afterwards) — with SPL_ND_BOUNDARIES=check: the analysis fails unless the device's lists equal the host's; and
    SPL_ND_BOUNDARIES=host gives the same factorisation as the default."""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_ND_GPU_MIN", "2000")
    rng = np.random.default_rng(15)
    U = pkg.umfpack
    if kind == "arrow":
        m = 70
        n0, A0 = _grid_matrix(pkg, O, "2d", m)
        S0 = csc_tuple_to_scipy(mat_to_tuple(A0)).tolil()
        for h in (0, 1, 2):  # three dense rows and columns
            S0[h, :] = -0.01
            S0[:, h] = -0.01
            S0[h, h] = 2.0 * n0
        S0 = sp.csc_matrix(S0)
        S0.sort_indices()
        A = pkg.Matrix(n0, n0, S0.indptr.astype(np.int32), S0.indices.astype(np.int32), S0.data)
    elif kind == "complex":
        n0, A0 = _grid_matrix(pkg, O, "3d", 22)
        S0 = csc_tuple_to_scipy(mat_to_tuple(A0))
        S0 = sp.csc_matrix((3.0 + 0.5j) * sp.identity(n0) - S0)
        S0.sort_indices()
        A = pkg.Matrix(n0, n0, S0.indptr.astype(np.int32), S0.indices.astype(np.int32), S0.data)
    else:
        n0, A = _grid_matrix(pkg, O, "3d", 30)
        S0 = csc_tuple_to_scipy(mat_to_tuple(A))
    xs = rng.uniform(0.5, 1.5, n0) + (1j * rng.uniform(0.5, 1.5, n0) if kind == "complex" else 0.0)
    b = np.asarray(S0 @ xs).ravel()
    monkeypatch.setenv("SPL_ND_BOUNDARIES", "check")
    f = U.factor(A, U.analyze(A))
    x = U.linearSolve_(f, U.UmfpackNormal, A, b)
    assert _backward_error(S0, x, b) <= 1e-13
    if kind == "3d_host_lists":
        monkeypatch.setenv("SPL_ND_BOUNDARIES", "host")
        g = U.factor(A, U.analyze(A))
        assert g.stats["flops"] == f.stats["flops"] and g.stats["fronts"] == f.stats["fronts"]
        assert np.array_equal(U.linearSolve_(g, U.UmfpackNormal, A, b), x)


# ---- threshold pivoting inside the diagonal blocks of the fronts (csrc/dense_lu_kernels.hpp, Band::piv) ------------
def test_block_pivoting_symmetric_retry_and_row_scales(gpu, pkg, monkeypatch):
    """(1) a SYMMETRIC matrix of 2 x 2 blocks [[1e-14, 3], [3, 1e-14]] with weak coupling: its L D L^T speculation (no
    interchanges) is rejected by the first solve, the same tree is factored again as LU with block pivoting (all the
    flops of the tree, block_pivoting = 1) and holds — no static pivoting, no host matching.  (2) an unsymmetric mesh
    matrix whose rows and columns are scaled by 10^U(-6, 6): candidates are compared after UMFPACK's row scaling, the
    speculation holds at once; both systems to a backward error at rounding level."""
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    U = pkg.umfpack
    rng = np.random.default_rng(77)
    m = 48
    n = m * m
    off = np.zeros(n - 1)
    off[0::2] = 3.0
    far = rng.uniform(-0.1, 0.1, n - m)
    B = sp.diags([far, off, np.full(n, 1e-14), off, far], (-m, -1, 0, 1, m), format="csc")
    B.sort_indices()
    assert abs(B - B.T).max() == 0
    M = pkg.Matrix(n, n, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data)
    fact = U.factor(M, U.analyze(M))
    full = None
    st = fact.stats
    assert st["path"] == 4 and st["block_pivoting"] == 0  # symmetric: L D L^T first
    half = st["flops"]
    xs = rng.uniform(0.5, 1.5, n)
    for mode, op in ((U.UmfpackNormal, B), (U.UmfpackTrans, B.T.tocsc())):
        b = np.asarray(op @ xs).ravel()
        x = U.linearSolve_(fact, mode, M, b)
        assert _backward_error(op, x, b) <= 1e-13
    st = fact.stats
    assert st["path"] == 4 and st["block_pivoting"] == 1 and st["flops"] == 2.0 * half
    # (2)
    T = sp.diags([np.ones(20), np.ones(21), np.ones(20)], (-1, 0, 1))
    I = sp.identity(21)
    P = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocoo()
    n2 = 21 ** 3
    v = rng.uniform(-1.0, 1.0, P.nnz)
    dr, dc = 10.0 ** rng.uniform(-6, 6, n2), 10.0 ** rng.uniform(-6, 6, n2)
    S = sp.csc_matrix((v * dr[P.row] * dc[P.col], (P.row, P.col)), shape=(n2, n2))
    S.sort_indices()
    M2 = pkg.Matrix(n2, n2, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    f2 = U.factor(M2, U.analyze(M2))
    assert f2.stats["block_pivoting"] == 1
    x2 = rng.uniform(0.5, 1.5, n2)
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.T))):
        b = np.asarray(op @ x2).ravel()
        assert _backward_error(op, U.linearSolve_(f2, mode, M2, b), b) <= 1e-13
    assert f2.path in (4, 5)


def test_umfpack_entry_points_from_several_threads(gpu, pkg, O, monkeypatch):
    """`safe` foreign calls may arrive concurrently from several OS threads (only FEAST serialises,
    Feast.hs:134): analyze / factor / solve of different matrices — a symmetric mesh (L D L^T), an unsymmetric one that
    is not dominant (block pivoting), a complex shift on native complex fronts, the GPU level service — from four
    threads at once, three rounds each, give what the same calls give one after the other"""
    import threading
    import scipy.sparse as sp
    monkeypatch.setenv("SPL_LU_METHOD", "mf")
    monkeypatch.setenv("SPL_ZI_NATIVE", "1")
    monkeypatch.setenv("SPL_ND_GPU_MIN", "3000")
    U = pkg.umfpack
    rng = np.random.default_rng(99)
    cases = []
    n1, A1 = _grid_matrix(pkg, O, "2d", 70)
    cases.append((A1, csc_tuple_to_scipy(mat_to_tuple(A1)), rng.uniform(0.5, 1.5, n1)))
    n2, A2 = _grid_matrix(pkg, O, "3d", 18)
    cases.append((A2, csc_tuple_to_scipy(mat_to_tuple(A2)), rng.uniform(0.5, 1.5, n2)))
    m = 50
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    P = sp.csc_matrix(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)))
    P.data = rng.uniform(-1.0, 1.0, P.nnz)
    P.setdiag(rng.uniform(2.0, 3.0, m * m))  # not dominant (column sums up to 4), fine with block pivoting
    P = sp.csc_matrix(P)
    P.sort_indices()
    cases.append((pkg.Matrix(m * m, m * m, P.indptr.astype(np.int32), P.indices.astype(np.int32), P.data), P,
                  rng.uniform(0.5, 1.5, m * m)))
    K = csc_tuple_to_scipy(mat_to_tuple(A1))
    Z = sp.csc_matrix((2.5 + 0.7j) * sp.identity(n1) - K)
    Z.sort_indices()
    cases.append((pkg.Matrix(n1, n1, Z.indptr, Z.indices, Z.data), Z, rng.uniform(0.5, 1.5, n1) + 1j * rng.uniform(0.5, 1.5, n1)))

    def run(case):
        A, S, xs = case
        b = np.asarray(S @ xs).ravel()
        f = U.factor(A, U.analyze(A))
        x = U.linearSolve_(f, U.UmfpackNormal, A, b)
        return x, _backward_error(S, x, b)

    serial = [run(c) for c in cases]
    assert all(be <= 1e-13 for _, be in serial)
    results, errors = {}, []

    def work(t):
        try:
            for it in range(3):
                results[(t, it)] = run(cases[t])
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for (t, it), (x, be) in results.items():
        assert be <= 1e-13 and np.array_equal(x, serial[t][0]), (t, it, be)
