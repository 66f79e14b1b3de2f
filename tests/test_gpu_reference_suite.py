"""sparse-linear/tests/Sparse.hs, item for item, against the PRODUCT (HIP path through the
C ABI) with the reference's own names.  Element type: integer-valued doubles (the
reference uses Int), so `==` is exact.  Every result is also compared with the oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import arbdim, arbitrary_triples, arbval, mat_to_tuple, tuples_equal

pytestmark = pytest.mark.gpu
S = dict(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))


def check_matrix(O, M):
    """checkMatrix (tests/Test/LinearAlgebra.hs:40-67)"""
    return O.check_matrix(mat_to_tuple(M)) == 0


@st.composite
def arb_matrix_args(draw):
    nr, nc = draw(arbdim), draw(arbdim)
    return nr, nc, draw(arbitrary_triples(nr, nc))


# describe "fromTriples" (checkMatrix arbitrary)
@settings(**S)
@given(arb_matrix_args())
def test_fromTriples(gpu, pkg, O, a):
    nr, nc, triples = a
    M = pkg.fromTriples(nr, nc, triples)
    assert check_matrix(O, M)
    assert tuples_equal(mat_to_tuple(M), O.fromTriples(nr, nc, triples))


def test_compress_errors(gpu, pkg):
    with pytest.raises(pkg.SparseError, match=r"compress: row index out of bounds \(0,2\) at 1"):
        pkg.fromTriples(2, 2, [(0, 0, 1.0), (2, 0, 1.0)])
    with pytest.raises(pkg.SparseError, match=r"compress: column index out of bounds \(0,2\) at 0"):
        pkg.fromTriples(2, 2, [(0, -1, 1.0), (5, 5, 1.0)][:1])
    with pytest.raises(pkg.SparseError, match="row and column array lengths differ"):
        pkg.compress(2, 2, [0], [0, 1], [1.0, 2.0])
    with pytest.raises(pkg.SparseError, match="row and value array lengths differ"):
        pkg.compress(2, 2, [0], [0], [1.0, 2.0])
    # rows are checked before columns (Sparse.hs:196-212)
    with pytest.raises(pkg.SparseError, match="row index"):
        pkg.fromTriples(2, 2, [(0, 9, 1.0), (7, 0, 1.0)])


def test_compress_large_with_many_duplicates(gpu, pkg, O):
    rng = np.random.default_rng(0)
    for nr, nc, k in ((50, 40, 30000), (3000, 2000, 200000), (5, 100000, 300000), (100000, 3, 250000)):
        r, c = rng.integers(0, nr, k), rng.integers(0, nc, k)
        v = rng.integers(-3, 4, k).astype(float)  # exact sums whatever the order
        M = pkg.compress(nr, nc, r, c, v)
        assert tuples_equal(mat_to_tuple(M), O.compress(nr, nc, r, c, v))
    # non-exact values: duplicates are summed in input order, like the oracle's stable sort
    r, c = rng.integers(0, 30, 5000), rng.integers(0, 30, 5000)
    v = rng.normal(size=5000)
    assert tuples_equal(mat_to_tuple(pkg.compress(30, 30, r, c, v)), O.compress(30, 30, r, c, v))
    Z = pkg.compress(4, 3, [], [], [])
    assert Z == pkg.zeros(4, 3)


# describe "kronecker"
@settings(**S)
@given(arbdim, arbdim)
def test_kronecker_identities(gpu, pkg, m, n):
    assert pkg.kronecker(pkg.ident(m), pkg.ident(n)) == pkg.ident(m * n)


@settings(**S)
@given(arb_matrix_args(), arb_matrix_args())
def test_kronecker_format(gpu, pkg, O, a, b):
    A, B = pkg.fromTriples(*a), pkg.fromTriples(*b)
    K = pkg.kronecker(A, B)
    assert check_matrix(O, K)
    assert np.array_equal(pkg.pack(K), np.kron(pkg.pack(A), pkg.pack(B)))
    assert tuples_equal(mat_to_tuple(K), O.kronecker(mat_to_tuple(A), mat_to_tuple(B)))  # device == restatement
    assert np.array_equal(pkg.takeDiag(K), O.take_diag(mat_to_tuple(K)))


# describe "diag"
@settings(**S)
@given(st.lists(arbval, min_size=1, max_size=30))
def test_diag(gpu, pkg, O, v):
    v = np.array(v)
    assert np.array_equal(pkg.takeDiag(pkg.diag(v)), v)
    assert check_matrix(O, pkg.diag(v))


# describe "mulV"
@settings(**S)
@given(st.lists(arbval, min_size=1, max_size=200))
def test_mulV_ident(gpu, pkg, v):
    v = np.array(v)
    assert np.array_equal(pkg.mulV(pkg.ident(len(v)), v), v)


# describe "addition"
@settings(**S)
@given(arb_matrix_args())
def test_add_ident_inv(gpu, pkg, O, a):
    A = pkg.fromTriples(*a)
    assert A + pkg.zeros(A.nrows, A.ncols) == A
    assert A - A == pkg.cmap(lambda v: v * 0, A)


@settings(**S)
@given(st.data())
def test_add_commute_assoc_format(gpu, pkg, O, data):
    nr, nc = data.draw(arbdim), data.draw(arbdim)
    A, B, C = (pkg.fromTriples(nr, nc, data.draw(arbitrary_triples(nr, nc))) for _ in range(3))
    assert A + B == B + A
    assert A + (B + C) == (A + B) + C
    assert check_matrix(O, A + B)
    assert tuples_equal(mat_to_tuple(A + B), O.add(mat_to_tuple(A), mat_to_tuple(B)))


def test_lin_matches_oracle_on_floats(gpu, pkg, O):
    rng = np.random.default_rng(4)
    for nr, nc, k in ((1, 1, 1), (30, 20, 200), (500, 400, 30000), (4, 5000, 9000)):
        A = pkg.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), rng.normal(size=k))
        B = pkg.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), rng.normal(size=k))
        for al, be in ((1.0, 1.0), (1.0, -1.0), (-1.0, 0.37), (2.5, 0.0)):
            assert tuples_equal(mat_to_tuple(pkg.lin(al, A, be, B)), O.lin(al, mat_to_tuple(A), be, mat_to_tuple(B)))
    with pytest.raises(pkg.SparseError, match="glin: row number mismatch"):
        pkg.ident(3) + pkg.zeros(4, 3)
    with pytest.raises(pkg.SparseError, match="glin: column number mismatch"):
        pkg.ident(3) + pkg.zeros(3, 4)


# describe "transpose"
@settings(**S)
@given(st.lists(arbval, min_size=1, max_size=30))
def test_transpose_diag(gpu, pkg, v):
    d = pkg.diag(np.array(v))
    assert pkg.transpose(d) == d


# describe "ctrans" (real-valued fixtures; sigma_y is complex: SURVEY.md §8f rank 3)
def test_ctrans_fixtures(gpu, pkg):
    m = pkg.fromTriples(2, 2, [(0, 0, 2), (0, 1, -1), (1, 0, -1), (1, 1, 2)])
    assert m == pkg.ctrans(m) and pkg.hermitian(m)
    sx = pkg.fromTriples(2, 2, [(0, 1, 1), (1, 0, 1)])
    assert sx == pkg.ctrans(sx)
    assert not pkg.hermitian(pkg.fromTriples(2, 2, [(0, 1, 1)]))


# describe "mul"
@settings(**S)
@given(arb_matrix_args())
def test_mul_identities(gpu, pkg, a):
    A = pkg.fromTriples(*a)
    assert pkg.ident(A.nrows) * A == A
    assert A * pkg.ident(A.ncols) == A


@settings(**S)
@given(st.data())
def test_mul_assoc_format(gpu, pkg, O, data):
    m, n, p, q = (data.draw(arbdim) for _ in range(4))
    A = pkg.fromTriples(m, n, data.draw(arbitrary_triples(m, n)))
    B = pkg.fromTriples(n, p, data.draw(arbitrary_triples(n, p)))
    C = pkg.fromTriples(p, q, data.draw(arbitrary_triples(p, q)))
    assert (A * B) * C == A * (B * C)
    assert check_matrix(O, A * B)
    assert tuples_equal(mat_to_tuple(A * B), O.mm(mat_to_tuple(A), mat_to_tuple(B), literal=True))


def test_mm_cancellation_and_errors(gpu, pkg):
    a = pkg.fromTriples(1, 2, [(0, 0, 1.0), (0, 1, -1.0)])
    b = pkg.fromTriples(2, 1, [(0, 0, 1.0), (1, 0, 1.0)])
    c = a * b
    assert c.pointers.tolist() == [0, 1] and c.indices.tolist() == [0] and c.values.tolist() == [0.0]
    with pytest.raises(pkg.SparseError, match="mm: inner dimension mismatch"):
        a * a


# describe "fromBlocksDiag"
@settings(**S)
@given(arbdim, arbdim)
def test_fromBlocksDiag_identity(gpu, pkg, m, n):
    assembled = pkg.fromBlocksDiag([[pkg.ident(m), pkg.ident(n)], [None, None]])
    assert assembled == pkg.ident(m + n)


@settings(**S)
@given(st.data())
def test_fromBlocksDiag_symmetric_blockwise(gpu, pkg, data):
    nr, nc = data.draw(arbdim), data.draw(arbdim)
    MN = pkg.fromTriples(nr, nc, data.draw(arbitrary_triples(nr, nc)))
    M = pkg.fromTriples(nr, nr, data.draw(arbitrary_triples(nr, nr)))
    N = pkg.fromTriples(nc, nc, data.draw(arbitrary_triples(nc, nc)))
    symM, symN = M + pkg.ctrans(M), N + pkg.ctrans(N)
    assembled = pkg.fromBlocksDiag([[symM, symN], [MN, pkg.ctrans(MN)]])
    assert assembled == pkg.ctrans(assembled)


@settings(**S)
@given(st.data())
def test_fromBlocksDiag_format(gpu, pkg, O, data):
    n = data.draw(st.integers(1, 4))
    mats = [pkg.fromTriples(*data.draw(arb_matrix_args())) for _ in range(n)]
    D = pkg.fromBlocksDiag([mats] + [[None] * n for _ in range(n - 1)])
    assert check_matrix(O, D)
    assert D == pkg.blockDiag(mats)
    assert D.nrows == sum(m.nrows for m in mats) and D.ncols == sum(m.ncols for m in mats)


# describe "Data.Matrix.Sparse.Foreign"
@settings(**S)
@given(arb_matrix_args())
def test_fromForeign_withConstMatrix(gpu, pkg, a):
    A = pkg.fromTriples(*a)
    assert pkg.withConstMatrix(A, lambda *t: pkg.fromForeign(True, *t)) == A


def test_kronecker_large_and_edges(gpu, pkg, O):
    """the reference's own model problem at size: kronecker (ident n) T + kronecker T (ident n) has
    5 n^2 - 4 n entries; plus empty operands, rectangular operands and complex values"""
    n = 300
    T = pkg.fromTriples(n, n, [(i, i, 2.0) for i in range(n)] + [(i, i + 1, -1.0) for i in range(n - 1)] +
                        [(i + 1, i, -1.0) for i in range(n - 1)])
    A = pkg.kronecker(pkg.ident(n), T) + pkg.kronecker(T, pkg.ident(n))
    assert A.nrows == n * n and int(A.pointers[-1]) == 5 * n * n - 4 * n
    assert np.array_equal(pkg.takeDiag(A), np.full(n * n, 4.0))
    assert check_matrix(O, A)
    Z = pkg.kronecker(pkg.zeros(3, 2), T)
    assert Z == pkg.zeros(3 * n, 2 * n)
    R = pkg.fromTriples(2, 3, [(0, 2, 1.5), (1, 0, -2.0)])
    K = pkg.kronecker(R, R)
    assert (K.nrows, K.ncols) == (4, 9) and np.array_equal(pkg.pack(K), np.kron(pkg.pack(R), pkg.pack(R)))
    Cm = pkg.Matrix(2, 2, [0, 1, 2], [0, 1], np.array([1 + 2j, 3 - 1j]))
    Kc = pkg.kronecker(Cm, Cm)
    assert np.array_equal(pkg.pack(Kc), np.kron(pkg.pack(Cm), pkg.pack(Cm)))
    assert np.array_equal(pkg.takeDiag(Cm), np.array([1 + 2j, 3 - 1j]))


# ---- hcat / vcat / fromBlocks / fromBlocksDiag on the device (spl_assemble_blocks) against the oracle's
# restatement of Sparse.hs:500-595 ----------------------------------------------------------------------
def _rand_mat(pkg, O, rng, nr, nc, k, cplx=False):
    A = O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), rng.integers(-9, 10, k).astype(float))
    vals = A[4] + 1j * rng.integers(-9, 10, len(A[4])) if cplx else A[4]
    return pkg.Matrix(nc, nr, A[2], A[3], vals)


def _same(M, t):
    return (M.nrows, M.ncols) == (t[0], t[1]) and np.array_equal(M.pointers, t[2]) and np.array_equal(M.indices, t[3]) \
        and np.array_equal(M.values, t[4])


@pytest.mark.parametrize("cplx", [False, True])
def test_hcat_vcat_match_oracle(gpu, pkg, O, cplx):
    rng = np.random.default_rng(11 + cplx)
    for _ in range(6):
        nr = int(rng.integers(1, 40))
        mats = [_rand_mat(pkg, O, rng, nr, int(rng.integers(1, 30)), int(rng.integers(0, 80)), cplx and i % 2 == 0)
                for i in range(int(rng.integers(1, 6)))]
        H = pkg.hcat(mats)
        assert _same(H, O.hcat([mat_to_tuple(m) for m in mats]))
        assert O.check_matrix((H.nrows, H.ncols, H.pointers, H.indices, np.abs(H.values))) == 0
        nc = int(rng.integers(1, 40))
        mats = [_rand_mat(pkg, O, rng, int(rng.integers(1, 30)), nc, int(rng.integers(0, 80)), cplx and i % 2 == 1)
                for i in range(int(rng.integers(1, 6)))]
        V = pkg.vcat(mats)
        assert _same(V, O.vcat([mat_to_tuple(m) for m in mats]))
        assert O.check_matrix((V.nrows, V.ncols, V.pointers, V.indices, np.abs(V.values))) == 0
    A, B = _rand_mat(pkg, O, rng, 7, 5, 20), _rand_mat(pkg, O, rng, 7, 3, 9)
    assert pkg.hjoin(A, B) == pkg.hcat([A, B]) and pkg.transpose(pkg.hjoin(A, B)) == pkg.vjoin(pkg.transpose(A), pkg.transpose(B))
    with pytest.raises(pkg.SparseError):
        pkg.hcat([])
    with pytest.raises(pkg.SparseError):
        pkg.hcat([A, pkg.transpose(A)])  # nrows mismatch
    with pytest.raises(pkg.SparseError):
        pkg.vcat([A, B])  # ncols mismatch


def test_fromBlocks_matches_oracle(gpu, pkg, O):
    rng = np.random.default_rng(5)
    hs, ws = [4, 9, 1, 6], [3, 8, 5]
    blocks = [[None if rng.random() < 0.3 else _rand_mat(pkg, O, rng, h, w, int(rng.integers(0, 30))) for w in ws] for h in hs]
    for r in range(len(hs)):  # every block row and column keeps one block: dimensions stay specified
        if all(b is None for b in blocks[r]):
            blocks[r][0] = _rand_mat(pkg, O, rng, hs[r], ws[0], 5)
    for c in range(len(ws)):
        if all(blocks[r][c] is None for r in range(len(hs))):
            blocks[0][c] = _rand_mat(pkg, O, rng, hs[0], ws[c], 5)
    M = pkg.fromBlocks(blocks)
    ref = O.fromBlocks([[None if b is None else mat_to_tuple(b) for b in row] for row in blocks])
    assert _same(M, ref) and (M.nrows, M.ncols) == (sum(hs), sum(ws))
    # == vcat . map hcat with explicit zero blocks (Sparse.hs:564)
    full = pkg.vcat([pkg.hcat([b if b is not None else pkg.zeros(hs[r], ws[c]) for c, b in enumerate(row)])
                     for r, row in enumerate(blocks)])
    assert M == full
    D = pkg.fromBlocksDiag([[blocks[0][0], blocks[1][1], blocks[2][2]], [blocks[0][1], blocks[1][2], None], [None, None, None]])
    refD = O.fromBlocksDiag([[mat_to_tuple(blocks[0][0]) if blocks[0][0] is not None else None,
                              mat_to_tuple(blocks[1][1]) if blocks[1][1] is not None else None,
                              mat_to_tuple(blocks[2][2]) if blocks[2][2] is not None else None],
                             [mat_to_tuple(blocks[0][1]) if blocks[0][1] is not None else None,
                              mat_to_tuple(blocks[1][2]) if blocks[1][2] is not None else None, None],
                             [None, None, None]]) if all(blocks[i][i] is not None for i in range(3)) else None
    if refD is not None:
        assert _same(D, refD)
    with pytest.raises(pkg.SparseError):
        pkg.fromBlocks([[None, blocks[0][1]], [None, blocks[1][1]]])  # underspecified widths
    with pytest.raises(pkg.SparseError):
        pkg.fromBlocks([[pkg.ident(2), pkg.ident(3)], [pkg.ident(3), pkg.ident(3)]])  # incompatible widths / heights
