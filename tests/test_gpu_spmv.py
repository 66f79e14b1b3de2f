"""GPU parity of the SpMV path (axpy_/mulV/axpy, Sparse.hs:433-471) through the C ABI."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

from helpers import arbitrary_dims_triples, arbval, tuple_to_mat
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu
GPU_SETTINGS = dict(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))


def test_ident_mulV(gpu, pkg):
    # tests/Sparse.hs:41-47: ident `mulV` v == v
    for m in (1, 2, 63, 64, 65, 257, 1000):
        v = np.arange(1, m + 1, dtype=np.float64) * 0.5 - 3
        assert np.array_equal(pkg.mulV(pkg.ident(m), v), v)


@settings(**GPU_SETTINGS)
@given(arbitrary_dims_triples(), st.data())
def test_mulV_matches_oracle_bitwise(gpu, pkg, O, dt, data):
    nr, nc, triples = dt
    A = O.fromTriples(nr, nc, triples)  # oracle-built matrix; device compress is tested elsewhere
    x = np.array(data.draw(st.lists(arbval, min_size=nc, max_size=nc)))
    y = pkg.mulV(tuple_to_mat(pkg, A), x)
    assert np.array_equal(y, O.mulV(A, x))


@settings(**GPU_SETTINGS)
@given(arbitrary_dims_triples(), st.data())
def test_axpy_matches_oracle_bitwise(gpu, pkg, O, dt, data):
    nr, nc, triples = dt
    A = O.fromTriples(nr, nc, triples)
    x = np.array(data.draw(st.lists(arbval, min_size=nc, max_size=nc)))
    y0 = np.array(data.draw(st.lists(arbval, min_size=nr, max_size=nr)))
    M = tuple_to_mat(pkg, A)
    assert np.array_equal(pkg.axpy(M, x, y0), O.axpy(A, x, y0))
    y = y0.copy()
    pkg.axpy_(M, x, y)
    yo = y0.copy()
    O.axpy_(A, x, yo)
    assert np.array_equal(y, yo)


def test_dimension_errors(gpu, pkg):
    A = pkg.ident(4)
    with pytest.raises(pkg.SparseError, match="axpy_: column dimension"):
        pkg.mulV(A, np.ones(5))
    with pytest.raises(pkg.SparseError, match="axpy_: row dimension"):
        pkg.axpy_(A, np.ones(4), np.ones(3))


def test_poisson2d_c1(gpu, pkg, O):
    """config C1: 1e4 x 1e4 5-point Poisson built the reference way
    (kronecker (ident n) T + kronecker T (ident n)), closed-form answers"""
    n = 100
    T = pkg.Matrix(n, n, *_tridiag(n))
    A = pkg.kronecker(pkg.ident(n), T) + pkg.kronecker(T, pkg.ident(n))
    assert pkg.nonZero(A) == 5 * n * n - 4 * n == 49600
    rp, ci, v = O.gen_poisson2d_csr(n)
    assert np.array_equal(A.pointers, rp) and np.array_equal(A.indices, ci) and np.array_equal(A.values, v)
    ones = np.ones(n * n)
    y = pkg.mulV(A, ones)
    ix, iy = np.meshgrid(np.arange(n), np.arange(n))
    nb = (ix > 0).astype(float) + (ix < n - 1) + (iy > 0) + (iy < n - 1)
    assert np.array_equal(y, (4 - nb).ravel())
    h = np.pi / (n + 1)
    vec = np.outer(np.sin((np.arange(n) + 1) * h), np.sin((np.arange(n) + 1) * h)).ravel()
    lam = 4 - 4 * np.cos(h)
    assert np.max(np.abs(pkg.mulV(A, vec) - lam * vec)) < 1e-13
    assert np.array_equal(pkg.mulV(A, vec), O.mulV((n * n, n * n, A.pointers, A.indices, A.values), vec))


def _tridiag(n):
    ptr = [0]
    idx, val = [], []
    for c in range(n):
        for r, x in ((c - 1, -1.0), (c, 2.0), (c + 1, -1.0)):
            if 0 <= r < n:
                idx.append(r)
                val.append(x)
        ptr.append(len(idx))
    return np.array(ptr), np.array(idx), np.array(val)


@pytest.mark.parametrize("kind,n,K", [("random", 5000, 20), ("random", 70000, 20), ("random", 300, 64),
                                      ("banded", 5000, 20), ("banded", 100000, 20),
                                      ("poisson2d", 37, 0), ("poisson3d", 13, 0)])
def test_synthetic_generators_match_oracle(gpu, pkg, O, kind, n, K):
    H = pkg.DeviceMatrix.synthetic(kind, n, K if K else 20)
    rp, ci, v = H.export_csr()
    if kind == "random":
        orp, oci, ov = O.gen_random_csr(n, K)
    elif kind == "banded":
        orp, oci, ov = O.gen_banded_csr(n)
    elif kind == "poisson2d":
        orp, oci, ov = O.gen_poisson2d_csr(n)
    else:
        orp, oci, ov = O.gen_poisson3d_csr(n)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(v, ov)


def test_synthetic_row_block(gpu, pkg, O):
    n = 10000
    H = pkg.DeviceMatrix.synthetic("random", n, 20, row0=2500, row1=5000)
    rp, ci, v = H.export_csr()
    orp, oci, ov = O.gen_random_csr(n, 20, row0=2500, row1=5000)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(v, ov)
    assert H.info()["row0"] == 2500 and H.info()["nrows_local"] == 2500


@pytest.mark.parametrize("variant", range(8))
@pytest.mark.parametrize("kind,n", [("random", 100003), ("banded", 50021), ("poisson3d", 23)])
def test_device_spmv_variants(gpu, pkg, O, kind, n, variant):
    torch = gpu
    H = pkg.DeviceMatrix.synthetic(kind, n, 20)
    H.set_variant(variant)
    N = H.info()["nrows_global"]
    rp, ci, v = H.export_csr()
    x = torch.empty(N, dtype=torch.float64, device="cuda")
    y = torch.full((N,), 7.0, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    pkg._ffi.check("vec", pkg._ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, N, x.data_ptr(), s))
    xh = O.gen_vector(N)
    assert np.array_equal(x.cpu().numpy(), xh)
    H.spmv_dev(x.data_ptr(), y.data_ptr(), accumulate=False, stream=s)
    torch.cuda.synchronize()
    yo = np.zeros(N)
    O.csr_gaxpy32(rp, ci, v, xh, yo)
    yg = y.cpu().numpy()
    if variant == 7:  # sub-wavefront kernel: tree order, tolerance only
        assert O.count_not_close(yg, yo, 1e-10) == 0
    else:  # streaming kernels keep the reference's summation order
        assert np.array_equal(yg, yo)
    # accumulate: y <- A x + y
    H.spmv_dev(x.data_ptr(), y.data_ptr(), accumulate=True, stream=s)
    torch.cuda.synchronize()
    yo2 = yo.copy()
    O.csr_gaxpy32(rp, ci, v, xh, yo2)
    if variant == 7:
        assert O.count_not_close(y.cpu().numpy(), yo2, 1e-10) == 0
    else:
        assert np.array_equal(y.cpu().numpy(), yo2)


def test_ragged_and_long_rows(gpu, pkg, O):
    """empty rows, one very long row (whole-chunk wavefront reduction), a medium row"""
    rng = np.random.default_rng(7)
    n = 3000
    lens = rng.integers(0, 6, size=n)
    lens[17] = 2600   # longer than several LDS chunks
    lens[18] = 0
    lens[1500] = 700
    lens[n - 1] = 1300
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]) if lens.sum() else []
    vals = rng.uniform(0.5, 1.5, size=len(rows))
    A = O.compress(n, n, rows, cols, vals)
    x = rng.uniform(0.5, 1.5, size=n)
    M = tuple_to_mat(pkg, A)
    for variant in range(8):
        M.device_handle().set_variant(variant)
        y = pkg.mulV(M, x)
        yo = O.mulV(A, x)
        assert O.count_not_close(y, yo, 1e-10) == 0
        short = lens < 256
        if variant != 7:
            assert np.array_equal(y[short], yo[short])


def test_empty_shapes(gpu, pkg):
    Z = pkg.zeros(5, 3)
    assert np.array_equal(pkg.mulV(Z, np.ones(3)), np.zeros(5))
    Z0 = pkg.zeros(0, 4)
    assert len(pkg.mulV(Z0, np.ones(4))) == 0


def test_invalid_matrix_rejected(gpu, pkg):
    bad = pkg.Matrix(2, 2, [0, 1, 2], [0, 5], [1.0, 1.0])  # row index 5 out of range
    with pytest.raises(pkg.SparseLinearError, match="invalid matrix"):
        pkg.mulV(bad, np.ones(2))
    bad2 = pkg.Matrix(2, 2, [0, 2, 1], [0, 1], [1.0, 1.0])  # pointers not monotone / wrong total
    with pytest.raises(pkg.SparseLinearError, match="invalid matrix"):
        pkg.mulV(bad2, np.ones(2))


def test_mulVT_and_mulM(gpu, pkg, O):
    rng = np.random.default_rng(3)
    A = O.compress(40, 30, rng.integers(0, 40, 300), rng.integers(0, 30, 300), rng.integers(-5, 6, 300).astype(float))
    M = tuple_to_mat(pkg, A)
    x = rng.integers(-4, 5, 40).astype(float)
    assert np.array_equal(pkg.mulVT(M, x), O.mulV(O.transpose(A), x))
    B = rng.integers(-3, 4, (30, 7)).astype(float)
    assert np.array_equal(pkg.mulM(M, B), O.mulM(A, B))


def test_transpose_matches_oracle(gpu, pkg, O):
    rng = np.random.default_rng(5)
    for nr, nc, k in ((1, 1, 1), (7, 3, 15), (50, 80, 900), (300, 200, 20000), (2000, 1500, 40000)):
        A = O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), rng.uniform(-1, 1, k))
        T = pkg.transpose(tuple_to_mat(pkg, A))
        To = O.transpose(A)
        assert (T.nrows, T.ncols) == (To[0], To[1])
        assert np.array_equal(T.pointers, To[2]) and np.array_equal(T.indices, To[3])
        assert np.array_equal(T.values, To[4])
    d = pkg.diag(np.arange(1.0, 9.0))
    assert pkg.transpose(d) == d  # tests/Sparse.hs:56-59


def test_transpose_long_rows(gpu, pkg, O):
    """a dense row / column exercises the LDS and the global-memory segment sorts"""
    n = 9000
    rows = np.concatenate([np.full(n, 3), np.arange(n), np.full(5000, 77)])
    cols = np.concatenate([np.arange(n), np.full(n, 5), np.arange(5000) + 100])
    vals = np.arange(len(rows), dtype=float) % 13 + 1
    A = O.compress(n, n, rows, cols, vals)
    T = pkg.transpose(tuple_to_mat(pkg, A))
    To = O.transpose(A)
    assert np.array_equal(T.pointers, To[2]) and np.array_equal(T.indices, To[3]) and np.array_equal(T.values, To[4])


def test_rowblock_partition(gpu, pkg, O):
    rng = np.random.default_rng(11)
    n = 5000
    k = 60000
    A = O.compress(n, n, rng.integers(0, n, k), rng.integers(0, n, k), rng.uniform(0.5, 1.5, k))
    x = rng.uniform(0.5, 1.5, n)
    yo = O.mulV(A, x)
    M = tuple_to_mat(pkg, A)
    for nparts in (2, 3, 8):
        parts = []
        covered = 0
        for p in range(nparts):
            H = pkg.DeviceMatrix.from_csc(M, part=p, nparts=nparts)
            inf = H.info()
            assert inf["row0"] == covered
            covered += inf["nrows_local"]
            parts.append(H.mulv(x))
        assert covered == n
        assert np.array_equal(np.concatenate(parts), yo)  # row sums never cross ranks: bit-identical


@pytest.mark.parametrize("k", [1, 3, 8, 16, 33, 70])
def test_fused_spmm_matches_per_column_axpy(gpu, pkg, O, k):
    """mulM (Sparse.hs:473-498): the fused sparse x dense kernel equals one axpy_ per column, bit for bit"""
    torch = gpu
    rng = np.random.default_rng(k)
    nr, nc, nz = 3000, 2500, 40000
    A = O.compress(nr, nc, rng.integers(0, nr, nz), rng.integers(0, nc, nz), rng.normal(size=nz))
    B = rng.normal(size=(nc, k))
    M = tuple_to_mat(pkg, A)
    C = pkg.mulM(M, B)
    assert np.array_equal(C, O.mulM(A, B))
    # device-resident form, accumulate
    H = M.device_handle()
    dB = torch.from_numpy(B).cuda()
    C0 = rng.normal(size=(nr, k))
    dC = torch.from_numpy(C0.copy()).cuda()
    H.spmm_dev(dB.data_ptr(), dC.data_ptr(), k, accumulate=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = np.stack([O.axpy(A, B[:, j].copy(), C0[:, j].copy()) for j in range(k)], axis=1)
    assert np.array_equal(dC.cpu().numpy(), ref)


@pytest.mark.parametrize("kind,n", [("random", 100003), ("banded", 50021), ("poisson3d", 23), ("poisson2d", 300)])
def test_sliced_ell_image_bitwise(gpu, pkg, O, kind, n):
    """variant 15 (csrc/spmv_sell.hip): one lane per row, sequential fold -> bit-identical for every row"""
    torch = gpu
    H = pkg.DeviceMatrix.synthetic(kind, n, 20)
    H.set_variant(15)
    assert H.info()["blocked_rows"] == -64
    N = H.info()["nrows_global"]
    rp, ci, v = H.export_csr()
    xh = O.gen_vector(N)
    x = torch.from_numpy(xh).cuda()
    y = torch.from_numpy(O.gen_vector(N, seed=5)).cuda()
    yo = y.cpu().numpy().copy()
    s = torch.cuda.current_stream().cuda_stream
    H.spmv_dev(x.data_ptr(), y.data_ptr(), accumulate=True, stream=s)
    torch.cuda.synchronize()
    O.csr_gaxpy32(rp, ci, v, xh, yo)
    assert np.array_equal(y.cpu().numpy(), yo)
    H.spmv_dev(x.data_ptr(), y.data_ptr(), accumulate=False, stream=s)
    torch.cuda.synchronize()
    yo2 = np.zeros(N)
    O.csr_gaxpy32(rp, ci, v, xh, yo2)
    assert np.array_equal(y.cpu().numpy(), yo2)


def test_sliced_ell_ragged_and_auto(gpu, pkg, O):
    rng = np.random.default_rng(23)
    n = 20000
    lens = rng.integers(0, 9, size=n)
    lens[100] = 700
    lens[n - 1] = 0
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens])
    A = O.compress(n, n, rows, cols, rng.normal(size=len(rows)))
    x = rng.normal(size=n)
    M = tuple_to_mat(pkg, A)
    H = pkg.DeviceMatrix.from_csc(M)
    H.set_variant(15)
    assert np.array_equal(H.mulv(x), O.mulV(A, x))  # long row included: still sequential, still exact
    # optimize() picks the sliced-ELL image for a large stencil matrix and keeps the bits
    P = pkg.DeviceMatrix.synthetic("poisson3d", 40)
    y0 = P.mulv(O.gen_vector(64000))
    P.optimize()
    assert P.info()["blocked_rows"] == -64
    assert np.array_equal(P.mulv(O.gen_vector(64000)), y0)


def test_abi_is_reentrant_across_threads(gpu, pkg, O):
    """the reference makes `safe` foreign calls that may arrive concurrently from several OS
    threads: same handle from 4 threads, and per-thread handles, all at once"""
    import threading
    rng = np.random.default_rng(31)
    n = 20000
    A = O.compress(n, n, rng.integers(0, n, 200000), rng.integers(0, n, 200000), rng.normal(size=200000))
    M = tuple_to_mat(pkg, A)
    shared = M.device_handle()
    xs = [rng.normal(size=n) for _ in range(4)]
    refs = [O.mulV(A, x) for x in xs]
    out, errs = [None] * 4, []

    def work(t):
        try:
            own = pkg.DeviceMatrix.from_csc(M)          # concurrent create / transpose on the device
            for _ in range(5):
                y1 = shared.mulv(xs[t])                 # concurrent SpMV on one handle
                y2 = own.mulv(xs[t])
                T = pkg.transpose(M)                    # concurrent one-shot calls
                assert np.array_equal(y1, refs[t]) and np.array_equal(y2, refs[t])
                assert T.nrows == n
            out[t] = True
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs and all(out)
