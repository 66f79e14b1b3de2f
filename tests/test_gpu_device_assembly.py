"""Device-resident forms of lin / transpose / compress (handle in, handle out: spl_matrix_lin,
spl_matrix_to_complex, spl_matrix_transpose, spl_matrix_compress_dev) against the oracle's restatements of
Sparse.hs:184-280 (compress), :301-329 (transpose), :401-431 (lin): structure and values bit for bit.  The tiled
`lin` kernel (csrc/assemble.hip) keeps the per-column walk of the reference, so its results equal those of the
thread-per-column kernel it replaces (SPL_LIN_TILED=0) and of orc_lin / orc_lin_z."""
import numpy as np
import pytest

from helpers import mat_to_tuple, tuple_to_mat, tuples_equal

pytestmark = pytest.mark.gpu


def rand_csc(O, rng, nr, nc, k, cplx=False):
    v = rng.uniform(0.5, 1.5, k)
    if cplx:
        v = v + 1j * rng.uniform(-1.0, 1.0, k)
    return O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), v) if not cplx else \
        _compress_c(O, nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), v)


def _compress_c(O, nr, nc, r, c, v):
    re = O.compress(nr, nc, r, c, v.real.copy())
    im = O.compress(nr, nc, r, c, v.imag.copy())
    return (nr, nc, re[2], re[3], re[4] + 1j * im[4])


def handle_to_csc_tuple(H):
    """(nrows, ncols, colptr, rowidx, values) of a whole-matrix handle: its CSR arrays are the CSC arrays of the
    transpose, so transpose back on the host side with a stable sort (order inside columns preserved)"""
    inf = H.info()
    rp, ci, v = H.export_csr()
    nr, nc = inf["nrows_local"], inf["ncols"]
    rows = np.repeat(np.arange(nr, dtype=np.int64), np.diff(rp))
    order = np.argsort(ci, kind="stable")
    cp = np.concatenate([[0], np.cumsum(np.bincount(ci, minlength=nc))]).astype(np.int64)
    return (nr, nc, cp, rows[order], v[order])


_LIN_SHAPES = [(1, 1, 1), (37, 129, 400), (5000, 4000, 90_000), (300, 128 * 7, 30_000)]


# every shape through both kernel sets; the large case once, through the tiled kernels
@pytest.mark.parametrize("shape,tiled", [(sh, t) for t in ("1", "0") for sh in _LIN_SHAPES] + [((200_000, 150_000, 3_000_000), "1")])
def test_lin_on_handles_matches_oracle_bitwise(gpu, pkg, O, monkeypatch, shape, tiled):
    monkeypatch.setenv("SPL_LIN_TILED", tiled)
    nr, nc, k = shape
    rng = np.random.default_rng(nr + nc)
    A, B = rand_csc(O, rng, nr, nc, k), rand_csc(O, rng, nr, nc, k // 2 + 1)
    HA = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, A))
    HB = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, B))
    big = nr >= 100_000
    for al, be in (((-1.0, 2.5),) if big else ((1.0, 1.0), (-1.0, 2.5), (0.0, 1.0))):
        HC = HA.lin(al, HB, be)
        Co = O.lin(al, A, be, B)
        assert tuples_equal(handle_to_csc_tuple(HC), Co)
        assert O.check_matrix(handle_to_csc_tuple(HC)) == 0
        HC.free()


def test_lin_long_columns_fall_back_inside_the_tiled_kernel(gpu, pkg, O):
    """a tile of 128 major slices holding more than the LDS image (3072 entries) walks global memory instead"""
    rng = np.random.default_rng(4)
    nr, nc = 300, 20_000  # handles are row-major: rows of A are the slices lin walks; 300 rows x ~2000 entries
    A, B = rand_csc(O, rng, nr, nc, 600_000), rand_csc(O, rng, nr, nc, 500_000)
    HA, HB = (pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, M)) for M in (A, B))
    HC = HA.lin(2.0, HB, -3.0)
    assert tuples_equal(handle_to_csc_tuple(HC), O.lin(2.0, A, -3.0, B))


def test_lin_complex_scalars_on_promoted_handles(gpu, pkg, O):
    """ze * B - A with a complex ze on real matrices (Feast.hs:214-216): promote with to_complex (cmap (:+ 0)),
    then lin with complex scalars — bit for bit orc_lin_z"""
    rng = np.random.default_rng(8)
    nr = nc = 3000
    A, B = rand_csc(O, rng, nr, nc, 40_000), rand_csc(O, rng, nr, nc, 9_000)
    HA = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, A)).to_complex()
    HB = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, B)).to_complex()
    assert HA.is_complex and HB.is_complex
    ze = 1.25 + 0.75j
    HC = HB.lin(ze, HA, -1.0)
    Co = O.lin_z(ze, B, -1.0, A)
    got = handle_to_csc_tuple(HC)
    assert np.array_equal(got[2], Co[2]) and np.array_equal(got[3], Co[3])
    assert np.array_equal(got[4].view(np.float64), np.asarray(Co[4], dtype=np.complex128).view(np.float64))
    with pytest.raises(Exception):
        pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, A)).lin(1j, pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, B)), 1.0)


def test_lin_shape_mismatch(gpu, pkg, O):
    rng = np.random.default_rng(1)
    HA = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, rand_csc(O, rng, 10, 12, 30)))
    HB = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, rand_csc(O, rng, 12, 10, 30)))
    with pytest.raises(Exception):
        HA.lin(1.0, HB, 1.0)


@pytest.mark.parametrize("shape", [(1, 1, 1), (37, 129, 400), (6000, 2500, 120_000)])
def test_transpose_on_handles(gpu, pkg, O, shape):
    nr, nc, k = shape
    rng = np.random.default_rng(k)
    A = rand_csc(O, rng, nr, nc, k)
    H = pkg.DeviceMatrix.from_csc(tuple_to_mat(pkg, A))
    HT = H.transpose()
    assert tuples_equal(handle_to_csc_tuple(HT), O.transpose(A))
    HTT = HT.transpose()
    assert tuples_equal(handle_to_csc_tuple(HTT), A)  # transpose is an involution (tests/Sparse.hs:56-61)


@pytest.mark.parametrize("shape", [(1, 1, 1), (50, 70, 2000), (4000, 3000, 200_000)])
def test_compress_from_device_triples(gpu, pkg, O, shape):
    torch = gpu
    nr, nc, k = shape
    rng = np.random.default_rng(k + 1)
    r, c = rng.integers(0, nr, k).astype(np.int32), rng.integers(0, nc, k).astype(np.int32)
    v = rng.integers(-9, 10, k).astype(float)  # exactly representable: duplicate sums do not depend on their order
    dr, dc, dv = torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    H = pkg.DeviceMatrix.compress_dev(nr, nc, k, dr.data_ptr(), dc.data_ptr(), dv.data_ptr())
    assert tuples_equal(handle_to_csc_tuple(H), O.compress(nr, nc, r, c, v))
    # the handle multiplies like the matrix
    xh = rng.integers(-3, 4, nc).astype(float)
    y = torch.zeros(nr, dtype=torch.float64, device="cuda")
    H.spmv_dev(torch.from_numpy(xh).cuda().data_ptr(), y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), O.mulV(O.compress(nr, nc, r, c, v), xh))


def test_compress_dev_bounds_rows_before_columns(gpu, pkg):
    torch = gpu
    r = torch.tensor([0, 5, 1], dtype=torch.int32, device="cuda")   # row 5 out of bounds (position 1)
    c = torch.tensor([9, 0, 0], dtype=torch.int32, device="cuda")   # column 9 out of bounds (position 0)
    v = torch.ones(3, dtype=torch.float64, device="cuda")
    with pytest.raises(Exception) as e:
        pkg.DeviceMatrix.compress_dev(3, 3, 3, r.data_ptr(), c.data_ptr(), v.data_ptr())
    assert "bounds" in str(e.value).lower() or "-" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nbits", [(0, 33), (1, 33), (63, 7), (64, 8), (4096, 16), (4097, 33), (70001, 9), (1 << 20, 33),
                                     (3_000_001, 41), (1 << 20, 64)])
def test_radix_sort_u64_matches_numpy(gpu, pkg, n, nbits):
    """csrc/radix_sort.hip (the hand-written sort behind the nested dissection's level structures: no vendor primitive
    in the product): keys with random low `nbits` bits and one common pattern above them come back ascending"""
    import torch
    rng = np.random.default_rng(n + nbits)
    lo = rng.integers(0, 1 << min(nbits, 62), n, dtype=np.uint64) if nbits < 64 else rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    hi = np.uint64(0) if nbits >= 62 else np.uint64(0x2A) << np.uint64(nbits)
    keys = (lo | hi).astype(np.uint64)
    if n > 10:  # duplicates and an already sorted stretch
        keys[: n // 8] = keys[n // 8: 2 * (n // 8)]
        keys[-(n // 4):] = np.sort(keys[-(n // 4):])
    d = torch.from_numpy(keys.view(np.int64).copy()).cuda() if n else torch.empty(0, dtype=torch.int64, device="cuda")
    st = pkg._ffi.lib().spl_debug_sort_u64(d.data_ptr() if n else None, n, nbits, None)
    assert st == 0
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), np.sort(keys))
