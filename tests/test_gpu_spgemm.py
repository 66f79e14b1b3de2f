"""SpGEMM (mm, Sparse.hs:691-702) on the HIP path: structure bit-exact, values bit-exact
(the kernels keep the reference's ascending-k accumulation order), every size bin."""
import os

import numpy as np
import pytest

from helpers import csc_tuple_to_scipy, mat_to_tuple, tuple_to_mat, tuples_equal

pytestmark = pytest.mark.gpu


def rand_csc(O, rng, nr, nc, k, ints=False):
    v = rng.integers(-4, 5, k).astype(float) if ints else rng.uniform(0.5, 1.5, k)
    return O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), v)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (40, 30, 50, 300), (300, 300, 300, 3000),
                                   (2000, 1500, 1800, 30000), (200, 5000, 100, 8000)])
def test_mm_matches_oracle_bitwise(gpu, pkg, O, shape):
    rng = np.random.default_rng(sum(shape))
    m, n, p, k = shape
    A, B = rand_csc(O, rng, m, n, k), rand_csc(O, rng, n, p, k)
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    Co = O.mm(A, B)
    assert tuples_equal(mat_to_tuple(C), Co)
    assert O.check_matrix(mat_to_tuple(C)) == 0


def test_mm_all_bins(gpu, pkg, O):
    """columns of B engineered so that every accumulator bin is used: a few products (LDS
    table per wavefront), a few thousand (LDS table per workgroup), tens of thousands
    (dense accumulator in HBM)"""
    rng = np.random.default_rng(9)
    n = 6000
    # A: ~40 per column, plus three dense-ish columns
    rows = [rng.integers(0, n, 40 * n)]
    cols = [np.repeat(np.arange(n), 40)]
    for c in (7, 1000, 4242):
        rows.append(rng.choice(n, 3000, replace=False))
        cols.append(np.full(3000, c))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = O.compress(n, n, rows, cols, rng.uniform(0.5, 1.5, len(rows)))
    # B: column 0 empty, column 1 tiny, column 2 medium (~60 entries -> ~2400 products),
    # column 3 heavy (hits the dense columns of A), the rest ~3 entries
    brow = [rng.integers(0, n, 3 * n), np.array([5]), rng.choice(n, 60, replace=False),
            np.concatenate([[7, 1000, 4242], rng.choice(n, 300, replace=False)])]
    bcol = [rng.integers(4, n, 3 * n), np.array([1]), np.full(60, 2), np.full(303, 3)]
    brow, bcol = np.concatenate(brow), np.concatenate(bcol)
    B = O.compress(n, n, brow, bcol, rng.uniform(0.5, 1.5, len(brow)))
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    Co = O.mm(A, B)
    assert np.array_equal(C.pointers, Co[2]) and np.array_equal(C.indices, Co[3])
    assert np.array_equal(C.values, Co[4])
    lens = np.diff(Co[2])
    assert lens[0] == 0 and lens[3] > 4096 and 256 < lens[2] <= 4096


def test_mm_vs_scipy_independent(gpu, pkg, O):
    rng = np.random.default_rng(21)
    A, B = rand_csc(O, rng, 400, 300, 5000, ints=True), rand_csc(O, rng, 300, 500, 5000, ints=True)
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    ref = (csc_tuple_to_scipy(A) @ csc_tuple_to_scipy(B)).toarray()
    assert np.array_equal(pkg.pack(C), ref)


def test_mm_A_times_A_random_generator(gpu, pkg, O):
    """A*A on a synthetic random matrix (the C4 shape in miniature)"""
    n = 20000
    rp, ci, v = O.gen_random_csr(n, 16)
    A = O.csr_to_csc_tuple(n, n, rp, ci, v)
    M = tuple_to_mat(pkg, A)
    C = M * M
    assert tuples_equal(mat_to_tuple(C), O.mm(A, A))


def test_mm_unsorted_input_columns(gpu, pkg, O):
    """fromForeign-style inputs with unsorted columns are tolerated (sorted on upload)"""
    A = pkg.Matrix(2, 3, [0, 2, 3], [2, 0, 1], [1.0, 2.0, 3.0])  # column 0 has rows (2, 0)
    As = pkg.Matrix(2, 3, [0, 2, 3], [0, 2, 1], [2.0, 1.0, 3.0])
    B = pkg.ident(2)
    assert A * B == As


def test_device_resident_spgemm_rmat(gpu, pkg, O):
    """handle-based A*A on a device-generated R-MAT graph (C4 in miniature), all three
    quadrant settings of SURVEY.md §8d; the generator, compress and SpGEMM all run in HBM"""
    for scale, ef, abc in ((10, 8, (0.25, 0.25, 0.25)), (12, 16, (0.45, 0.22, 0.22)), (11, 32, (0.57, 0.19, 0.19))):
        n = 1 << scale
        H = pkg.DeviceMatrix.rmat(scale, ef, abc)
        rp, ci, v = H.export_csr()
        r, c, vals = O.gen_rmat_coo(scale, n * ef, abc)
        A_csc = O.compress(n, n, r, c, vals)
        A_csr = O.transpose(A_csc)  # CSR(A) = CSC(A^T)
        assert np.array_equal(rp, A_csr[2]) and np.array_equal(ci, A_csr[3]) and np.array_equal(v, A_csr[4])
        HC, products = H.spgemm(H)
        crp, cci, cv = HC.export_csr()
        Co = O.mm(A_csc, A_csc)
        Ct = O.transpose(Co)
        assert np.array_equal(crp, Ct[2]) and np.array_equal(cci, Ct[3]) and np.array_equal(cv, Ct[4])
        lens = np.diff(A_csc[2])
        assert products == int(np.sum(lens[A_csc[3]]))  # sum over entries (k, j) of |A[:,k]|


@pytest.mark.parametrize("nparts", [1, 3, 8])
def test_colblock_spgemm_hip_blocks_concatenate_to_mm(gpu, pkg, O, nparts):
    """§8e: the column blocks of B each rank would multiply on its GPU (here one after the other
    on one GPU, HIP SpGEMM as the local operator) concatenate to exactly A*B"""
    rng = np.random.default_rng(31 + nparts)
    A, B = rand_csc(O, rng, 900, 800, 9000), rand_csc(O, rng, 800, 1100, 7000)
    Am, Bm = tuple_to_mat(pkg, A), tuple_to_mat(pkg, B)
    blocks = []
    for p in range(nparts):
        op = pkg.dist.ColBlockSpGEMM(Am, Bm, p, nparts, pkg.mm)
        blocks.append(op.step())
        assert blocks[-1].ncols == op.bounds[p + 1] - op.bounds[p]
    C = pkg.hcat(blocks)
    assert tuples_equal(mat_to_tuple(C), O.mm(A, B))


@pytest.mark.parametrize("shape_name", ["small", "large"])
def test_ordered_form_matches_oracle(gpu, pkg, O, monkeypatch, shape_name):
    """SPL_SPGEMM_ORDERED=1: every column written at its final offset through the look-back chain (no
    compaction), counting sort over row buckets, bucket-overflow fallback to the merge tree, heavy columns
    copied from scratch slots — same structure and values, bit for bit (Sparse.hs:691-702)"""
    monkeypatch.setenv("SPL_SPGEMM_ORDERED", "1")
    monkeypatch.setenv("SPL_SPGEMM_ORDERED_SHAPE", shape_name)  # 1536 / 96 (four workgroups per CU) or 2048 / 128
    rng = np.random.default_rng(17)
    for shape in ((40, 30, 50, 300), (2000, 1500, 1800, 30000), (200, 5000, 100, 8000), (300, 300, 300, 3000)):
        m, n, p, k = shape
        A, B = rand_csc(O, rng, m, n, k), rand_csc(O, rng, n, p, k)
        C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
        assert tuples_equal(mat_to_tuple(C), O.mm(A, B))
    # rows crowded into a narrow range (one bucket overflows) + columns of every class
    n = 6000
    rows = np.concatenate([rng.integers(0, 40, 30 * n), rng.integers(0, n, 10 * n)])
    cols = np.concatenate([np.repeat(np.arange(n), 30), np.repeat(np.arange(n), 10)])
    A = O.compress(n, n, rows, cols, rng.uniform(0.5, 1.5, len(rows)))
    brow = np.concatenate([rng.integers(0, n, 3 * n), rng.choice(n, 60, replace=False), rng.choice(n, 300, replace=False)])
    bcol = np.concatenate([rng.integers(4, n, 3 * n), np.full(60, 2), np.full(300, 3)])
    B = O.compress(n, n, brow, bcol, rng.uniform(0.5, 1.5, len(brow)))
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    assert tuples_equal(mat_to_tuple(C), O.mm(A, B))
    # very sparse A (one entry per column) times columns of B with 130 ... 300 entries: few products, but more
    # entries of B than the ordered kernel's own classes take: these columns go through the listed kernels
    n = 4000
    A = O.compress(n, n, rng.permutation(n), np.arange(n), rng.uniform(0.5, 1.5, n))
    brow = np.concatenate([rng.choice(n, k, replace=False) for k in (130, 200, 300, 5, 64, 128, 129)])
    bcol = np.concatenate([np.full(k, c) for c, k in enumerate((130, 200, 300, 5, 64, 128, 129))])
    B = O.compress(n, 7, brow, bcol, rng.uniform(0.5, 1.5, len(brow)))
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    assert tuples_equal(mat_to_tuple(C), O.mm(A, B))
    for scale, ef, abc in ((12, 16, (0.45, 0.22, 0.22)), (13, 32, (0.25, 0.25, 0.25))):
        H = pkg.DeviceMatrix.rmat(scale, ef, abc)
        HC, products = H.spgemm(H)
        crp, cci, cv = HC.export_csr()
        rp, ci, v = H.export_csr()
        nn = 1 << scale
        At = (nn, nn, rp, ci.astype(np.int64), v)
        Cs = O.mm(At, At)
        assert np.array_equal(crp, Cs[2]) and np.array_equal(cci, Cs[3]) and np.array_equal(cv, Cs[4])


@pytest.mark.parametrize("nrows", [(1 << 21) - 1, 1 << 21, (1 << 21) + 1])
def test_ordered_form_last_row_last_tiebreak(gpu, pkg, O, monkeypatch, nrows):
    """the ordered form packs key = row << 11 | t and uses 0xffffffff as "no product"; with 2^21 rows
    a column of exactly 2048 products whose last product hits the last row would produce that very key.  The
    ordered form is now refused from 2^21 rows on; around the bound, forced on with the 2048-product shape, such
    a column must come out right (and C keep its format invariants)"""
    monkeypatch.setenv("SPL_SPGEMM_ORDERED", "1")
    monkeypatch.setenv("SPL_SPGEMM_ORDERED_SHAPE", "large")
    rng = np.random.default_rng(3)
    # A: nrows x 2; column 0 has 2047 entries, column 1 one entry in the last row.  B: 2 x 3, column 0 selects
    # both (2048 products, the last with t = 2047 in row nrows - 1), columns 1, 2 ordinary
    r0 = np.sort(rng.choice(nrows - 1, 2047, replace=False))
    A = O.compress(nrows, 2, np.concatenate([r0, [nrows - 1]]), np.concatenate([np.zeros(2047, dtype=np.int64), [1]]),
                   rng.uniform(0.5, 1.5, 2048))
    B = O.compress(2, 3, np.array([0, 1, 0, 1]), np.array([0, 0, 1, 2]), rng.uniform(0.5, 1.5, 4))
    C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
    Co = O.mm(A, B)
    assert tuples_equal(mat_to_tuple(C), Co)
    assert np.diff(Co[2])[0] == 2048 and Co[3][2047] == nrows - 1
    assert O.check_matrix(mat_to_tuple(C)) == 0


@pytest.mark.parametrize("form", [{}, {"SPL_SPGEMM_ORDERED": "1"}, {"SPL_SPGEMM_ORDERED": "0"}, {"SPL_SPGEMM_TWO_PASS": "1"}])
def test_long_columns_of_b_with_few_products(gpu, pkg, O, monkeypatch, form):
    """a column of B with hundreds to thousands of entries that selects mostly EMPTY columns of A has few products:
    it must not go to the one-wavefront kernel, whose LDS stages at most 256 entries of B (found by
    tools/fuzz_spgemm.py: 2049 x 4000 with 1 187 entries times a 4000 x 1 column of 1 899 entries hung the kernel)"""
    for k, v in form.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(29)
    for m, n, ka, nbs in ((2049, 4000, 1187, (1899,)), (3000, 4000, 100, (2000, 257, 3000, 5)), (500, 9000, 40, (8000, 300))):
        A = O.compress(m, n, rng.integers(0, m, ka), rng.integers(0, n, ka), rng.normal(size=ka))
        brow = np.concatenate([rng.choice(n, k, replace=False) for k in nbs])
        bcol = np.concatenate([np.full(k, c) for c, k in enumerate(nbs)])
        B = O.compress(n, len(nbs), brow, bcol, rng.normal(size=len(brow)))
        C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
        assert tuples_equal(mat_to_tuple(C), O.mm(A, B))



@pytest.mark.parametrize("m", [70001, 4097, 16])
def test_dense_accumulator_columns_any_row_count(gpu, pkg, O, m):
    """columns beyond every LDS bin (tens of thousands of products, or thousands of entries of B) go through the
    dense accumulator in HBM, whose flags are swept 16 rows per thread: row counts that are not multiples of 16,
    rows touched at the very end of the range, several dense columns per workgroup slot, both single-pass forms"""
    rng = np.random.default_rng(m)
    n = 3000
    ka = min(40 * n, m * n // 2)
    rows, cols = rng.integers(0, m, ka), rng.integers(0, n, ka)
    rows[:50] = m - 1  # the last row, in many columns
    A = O.compress(m, n, rows, cols, rng.normal(size=ka))
    nbs = (2500, 2100, 30, 2999, 1)
    brow = np.concatenate([rng.choice(n, k, replace=False) for k in nbs])
    bcol = np.concatenate([np.full(k, c) for c, k in enumerate(nbs)])
    B = O.compress(n, len(nbs), brow, bcol, rng.normal(size=len(brow)))
    ref = O.mm(A, B)
    for env in ({}, {"SPL_SPGEMM_ORDERED": "1"}, {"SPL_SPGEMM_TWO_PASS": "1"}):
        for k in ("SPL_SPGEMM_ORDERED", "SPL_SPGEMM_TWO_PASS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            C = pkg.mm(tuple_to_mat(pkg, A), tuple_to_mat(pkg, B))
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert tuples_equal(mat_to_tuple(C), ref), env
