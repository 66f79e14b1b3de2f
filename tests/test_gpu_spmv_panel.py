"""The order-free SpMV mode (csrc/spmv_panel.hip): workgroup-wide, column-sorted panels whose
products are added with LDS atomics in whatever order the hardware executes them.  Contract:
north_star's 1e-10 relative on values against the reference order (Sparse.hs:447-451) — checked
here against the oracle with the reference's closeness predicate (feast/tests/test-feast.hs:17-19)
and, much tighter, against a bound of a few ulps of sum |a x|; on exactly representable data
(integers) the order cannot matter and the results must be bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(torch, H, x):
    n = H.info()["nrows_local"]
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.mark.parametrize("n,K,P,w,unroll,form", [
    # form 1 / 2: one chunk per load instruction, 1 / 2 index blocks per phase
    (50_003, 20, 3000, 12, 4, 1), (50_003, 20, 3000, 12, 6, 2), (200_000, 20, 20479, 14, 0, 2),
    (131_072, 7, 8192, 17, 12, 2), (70_001, 40, 5000, 10, 10, 2), (1000, 3, 64, 4, 4, 1),
    (300_000, 20, 19000, 13, 8, 2),
    # form 4 / 5: paired storage (16-byte value loads), 1 / 2 index blocks per phase; unroll counts pairs
    (50_003, 20, 3000, 12, 2, 4), (300_000, 20, 19000, 13, 4, 5), (131_072, 7, 8192, 17, 3, 4),
    (200_000, 20, 20479, 14, 6, 5), (1000, 3, 64, 4, 2, 4), (200_000, 20, 20479, 14, 0, 0),
    # phases far longer than the register pipeline: the un-pipelined tail loop
    (70_001, 40, 5000, 13, 3, 5), (70_001, 40, 5000, 13, 4, 1),
    # form 6 / 7: ring form (loader wavefronts hand units to gather wavefronts), 1 / 2 index blocks per phase;
    # unroll counts the units a loader keeps in flight
    (50_003, 20, 3000, 12, 6, 6), (300_000, 20, 19000, 13, 6, 7), (131_072, 7, 8192, 17, 4, 7), (1000, 3, 64, 4, 6, 6),
    (70_001, 40, 5000, 13, 8, 7), (200_000, 20, 19700, 14, 0, 7), (64, 1, 64, 4, 5, 7), (5_000_000, 3, 19700, 17, 6, 7)])
def test_panel_matches_oracle(gpu, pkg, O, n, K, P, w, unroll, form):
    torch = gpu
    H = pkg.DeviceMatrix.synthetic("random", n, K)
    H.build_panel(P, w, unroll, form)
    H.set_variant(16)
    assert H.spmv_kernel() == 16 and H.info()["blocked_rows"] == P
    rp, ci, v = H.export_csr()
    xh = O.gen_vector(n)
    x = torch.from_numpy(xh).cuda()
    y = _run(torch, H, x)
    yo = np.zeros(n)
    O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, yo)
    assert O.count_not_close(y, yo, 1e-10) == 0
    # rounding-level: |y - yo| <= 2 * len(row) * eps * sum |a x|  (values and x are positive here)
    lens = np.diff(rp)
    assert np.all(np.abs(y - yo) <= 2.0 * np.maximum(lens, 1) * np.finfo(float).eps * np.abs(yo))
    # accumulate form: y <- A x + y
    y0 = O.gen_vector(n, seed=7)
    yd = torch.from_numpy(y0.copy()).cuda()
    H.spmv_dev(x.data_ptr(), yd.data_ptr(), accumulate=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ya = y0.copy()
    O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, ya)
    assert O.count_not_close(yd.cpu().numpy(), ya, 1e-10) == 0
    assert H.panel_errors() == 0


def test_panel_exact_on_integers(gpu, pkg, O):
    """integer-valued entries and vector: every partial sum is exact, so any order gives the same bits"""
    torch = gpu
    rng = np.random.default_rng(5)
    n, k = 40_000, 700_000
    A = O.compress(n, n, rng.integers(0, n, k), rng.integers(0, n, k), rng.integers(-9, 10, k).astype(float))
    M = pkg.Matrix(n, n, A[2], A[3], A[4])
    H = pkg.DeviceMatrix.from_csc(M)
    H.build_panel(2500, 11, 0, 0)
    H.set_variant(16)
    xh = rng.integers(-5, 6, n).astype(float)
    y = _run(torch, H, torch.from_numpy(xh).cuda())
    assert np.array_equal(y, O.mulV(A, xh))


def test_order_switch_and_optimize(gpu, pkg, O):
    """spl_matrix_set_spmv_order: the default (reference order) never uses the panel image; ORDER_FREE
    makes variant 0 use it once built; switching back restores bit-identical results"""
    torch = gpu
    n = 120_000
    H = pkg.DeviceMatrix.synthetic("random", n, 20)
    H.build_blocked(600, 12, 0)
    H.build_panel(9000, 12, 0, 0)
    assert H.spmv_kernel() == 8
    rp, ci, v = H.export_csr()
    xh = O.gen_vector(n)
    x = torch.from_numpy(xh).cuda()
    yo = np.zeros(n)
    O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, yo)
    assert np.array_equal(_run(torch, H, x), yo)
    H.set_spmv_order(H.ORDER_FREE)
    assert H.spmv_kernel() == 16
    assert O.count_not_close(_run(torch, H, x), yo, 1e-10) == 0
    H.set_spmv_order(H.ORDER_REFERENCE)
    assert H.spmv_kernel() == 8
    assert np.array_equal(_run(torch, H, x), yo)


def test_panel_empty_rows_and_ragged_edges(gpu, pkg, O):
    """rows without entries, a last panel with a few rows, a last index block with a few columns,
    an all-empty matrix"""
    torch = gpu
    rng = np.random.default_rng(3)
    nr, nc = 10_007, 4_099
    rows = rng.integers(0, nr, 30_000)
    rows = rows[(rows % 7 != 0) & (rows < nr - 5)]  # every 7th row and the last rows stay empty
    cols = rng.integers(0, nc, len(rows))
    A = O.compress(nr, nc, rows, cols, rng.uniform(0.5, 1.5, len(rows)))
    H = pkg.DeviceMatrix.from_csc(pkg.Matrix(nc, nr, A[2], A[3], A[4]))
    H.build_panel(1000, 10, 4, 2)
    H.set_variant(16)
    xh = rng.uniform(0.5, 1.5, nc)
    y = _run(torch, H, torch.from_numpy(xh).cuda())
    yo = O.mulV(A, xh)
    assert O.count_not_close(y, yo, 1e-10) == 0 and np.all(y[::7] == 0.0)
    Z = pkg.DeviceMatrix.from_csc(pkg.zeros(300, 200))
    Z.build_panel(64, 4, 4, 1)
    Z.set_variant(16)
    assert np.array_equal(_run(torch, Z, torch.ones(200, dtype=torch.float64, device="cuda")), np.zeros(300))


@pytest.mark.parametrize("slices", [2, 4, 8])
def test_panel_column_slices(gpu, pkg, O, monkeypatch, slices):
    """column slices (csrc/spmv_panel.hip): a panel's index blocks dealt to `slices` workgroups that add their parts
    of the row sums into y with atomics — y = A x and y <- A x + y to the order-free contract, exact on integers"""
    torch = gpu
    monkeypatch.setenv("SPL_PANEL_SLICES", str(slices))
    n, K = 600_000, 20
    H = pkg.DeviceMatrix.synthetic("random", n, K, row0=100_000, row1=400_000)  # a row block: 300 000 rows, all columns
    H.build_panel(9000, 12, 0, 5)
    H.set_variant(16)
    rp, ci, v = H.export_csr()
    xh = O.gen_vector(n)
    x = torch.from_numpy(xh).cuda()
    y = _run(torch, H, x)
    yo = np.zeros(300_000)
    O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, yo)
    assert O.count_not_close(y, yo, 1e-10) == 0
    lens = np.diff(rp)
    assert np.all(np.abs(y - yo) <= 2.0 * np.maximum(lens, 1) * np.finfo(float).eps * np.abs(yo))
    y0 = O.gen_vector(300_000, seed=7)
    yd = torch.from_numpy(y0.copy()).cuda()
    H.spmv_dev(x.data_ptr(), yd.data_ptr(), accumulate=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ya = y0.copy()
    O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, ya)
    assert O.count_not_close(yd.cpu().numpy(), ya, 1e-10) == 0
    # the automatic choice on the same block picks slices by itself (one generation of tall panels)
    monkeypatch.delenv("SPL_PANEL_SLICES")
    H2 = pkg.DeviceMatrix.synthetic("random", 10_000_000, K, row0=0, row1=1_250_000)
    H2.set_spmv_order(H2.ORDER_FREE)
    H2.optimize()
    assert H2.spmv_kernel() == 16 and H2.info()["blocked_rows"] > 15_000


def test_panel_column_slices_exact_on_integers(gpu, pkg, O, monkeypatch):
    torch = gpu
    monkeypatch.setenv("SPL_PANEL_SLICES", "4")
    rng = np.random.default_rng(6)
    n, k = 40_000, 700_000
    A = O.compress(n, n, rng.integers(0, n, k), rng.integers(0, n, k), rng.integers(-9, 10, k).astype(float))
    H = pkg.DeviceMatrix.from_csc(pkg.Matrix(n, n, A[2], A[3], A[4]))
    H.build_panel(2500, 8, 0, 4)
    H.set_variant(16)
    xh = rng.integers(-5, 6, n).astype(float)
    y = _run(torch, H, torch.from_numpy(xh).cuda())
    assert np.array_equal(y, O.mulV(A, xh))


@pytest.mark.parametrize("n", [150_000, 1 << 19, 2_300_000])
def test_optimize_picks_the_image_the_sweep_measured_faster(gpu, pkg, O, n):
    """round 4 (tools/probe/panel_threshold_sweep.py): a CU gathers x through the 4 MiB L2 of its own XCD, not through
    "the aggregate L2" — with order-free sums the column-sorted panels beat the CSR-stream kernel from 1 MB of x on
    (round 3 left everything below 32 MiB to the stream kernel: config C4's R-MAT matrix ran at 1.2 instead of 3.4
    TB/s).  Round 5 (tools/probe/blocked_threshold_sweep.py): shaped for the whole chip — panel height from the row
    count, x window so that a segment is about ten chunks — the reference-order blocked image beats the stream kernel
    from 4 MiB of x on (round 4 drew the line at 12 MiB from a sweep whose blocked image left most CUs idle).  Whatever
    is picked: the reference order bit for bit, the free order to 1e-10."""
    torch = gpu
    H = pkg.DeviceMatrix.synthetic("random", n, 20)
    rp, ci, v = H.export_csr()
    xh = O.gen_vector(n)
    x = torch.from_numpy(xh).cuda()
    yo = np.zeros(n)
    O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, yo)
    H.optimize()  # reference order
    assert H.spmv_kernel() == (8 if n * 8 >= (4 << 20) else 0)
    assert np.array_equal(_run(torch, H, x), yo)
    H.free()
    H = pkg.DeviceMatrix.synthetic("random", n, 20)
    H.set_spmv_order(H.ORDER_FREE)
    H.optimize()
    assert H.spmv_kernel() == 16
    assert O.count_not_close(_run(torch, H, x), yo, 1e-10) == 0
    H.free()
    # rows that share x lines keep the sliced ELL image in either order
    B = pkg.DeviceMatrix.synthetic("banded", max(n, 400_000), 20)
    B.set_spmv_order(B.ORDER_FREE)
    B.optimize()
    assert B.spmv_kernel() == 15
    B.free()
