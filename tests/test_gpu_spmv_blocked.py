"""The column-blocked SpMV image (csrc/spmv_blocked.hip): bit-identical to the oracle's
reference-order fold for every panel / block shape, accumulate mode, ragged rows, row blocks."""
import numpy as np
import pytest

from helpers import tuple_to_mat

pytestmark = pytest.mark.gpu


def _device_spmv(torch, pkg, H, x_host, y0=None):
    inf = H.info()
    x = torch.from_numpy(x_host).cuda()
    y = torch.zeros(inf["nrows_local"], dtype=torch.float64, device="cuda") if y0 is None else torch.from_numpy(y0).cuda()
    H.spmv_dev(x.data_ptr(), y.data_ptr(), accumulate=y0 is not None, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.mark.parametrize("rw,w,unroll", [(64, 8, 0), (64, 10, -2), (100, 12, 4), (1024, 14, -8), (1221, 18, 10), (2560, 16, 12),
                                         (333, 9, -4), (64, 8, 8), (1280, 17, 0), (5000, 12, -1), (77, 6, 4)])
@pytest.mark.parametrize("kind,n", [("random", 100003), ("banded", 30011), ("poisson3d", 21)])
def test_blocked_bitwise(gpu, pkg, O, kind, n, rw, w, unroll):
    torch = gpu
    H = pkg.DeviceMatrix.synthetic(kind, n, 20)
    N = H.info()["nrows_global"]
    rp, ci, v = H.export_csr()
    H.build_blocked(rw, w, unroll)
    H.set_variant(8)
    xh = O.gen_vector(N)
    yo = np.zeros(N)
    O.csr_gaxpy32(rp, ci, v, xh, yo)
    assert np.array_equal(_device_spmv(torch, pkg, H, xh), yo)
    y0 = O.gen_vector(N, seed=77)
    yo2 = y0.copy()
    O.csr_gaxpy32(rp, ci, v, xh, yo2)
    assert np.array_equal(_device_spmv(torch, pkg, H, xh, y0.copy()), yo2)


def test_blocked_ragged_rows_and_duplicated_row_runs(gpu, pkg, O):
    """long rows produce long runs of equal rows inside one 64-entry chunk and across chunks"""
    rng = np.random.default_rng(12)
    n = 5000
    lens = rng.integers(0, 5, size=n)
    lens[3] = 900
    lens[4] = 70
    lens[5] = 0
    lens[4000] = 3000
    lens[n - 1] = 200
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens])
    A = O.compress(n, n, rows, cols, rng.uniform(0.5, 1.5, len(rows)))
    x = rng.uniform(0.5, 1.5, n)
    yo = O.mulV(A, x)
    for rw, w, u in ((64, 8, 0), (256, 10, 4), (1000, 13, -4), (4096, 9, 12), (70, 5, 8)):
        M = tuple_to_mat(pkg, A)
        H = pkg.DeviceMatrix.from_csc(M)
        H.build_blocked(rw, w, u)
        H.set_variant(8)
        assert np.array_equal(H.mulv(x), yo)
        y = rng.uniform(-1, 1, n)
        yo2 = O.axpy(A, x, y)
        assert np.array_equal(H.gaxpy(x, y.copy()), yo2)


def test_blocked_row_blocks_and_rectangular(gpu, pkg, O):
    rng = np.random.default_rng(13)
    nr, nc, k = 7000, 20000, 90000
    A = O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), rng.uniform(0.5, 1.5, k))
    x = rng.uniform(0.5, 1.5, nc)
    yo = O.mulV(A, x)
    M = tuple_to_mat(pkg, A)
    parts = []
    for p in range(3):
        H = pkg.DeviceMatrix.from_csc(M, part=p, nparts=3)
        H.build_blocked(130, 11, 0)
        H.set_variant(8)
        parts.append(H.mulv(x))
    assert np.array_equal(np.concatenate(parts), yo)


def test_auto_choice(gpu, pkg, O):
    """optimize(): small or banded matrices keep the CSR-stream kernel; a large random one is blocked;
    the answer is the same bits either way"""
    H = pkg.DeviceMatrix.synthetic("random", 6_000_000, 8)   # x = 48 MB > aggregate L2, no locality
    torch = gpu
    xh = O.gen_vector(6_000_000)
    y_stream = _device_spmv(torch, pkg, H, xh)
    H.optimize()
    y_auto = _device_spmv(torch, pkg, H, xh)
    assert np.array_equal(y_stream, y_auto)
    H.set_variant(1)
    assert np.array_equal(_device_spmv(torch, pkg, H, xh), y_stream)
    rp, ci, v = O.gen_random_csr(6_000_000, 8, row0=0, row1=2000)
    yo = np.zeros(2000)
    O.csr_gaxpy32(rp, ci, v, xh, yo)
    assert np.array_equal(y_auto[:2000], yo)


@pytest.mark.parametrize("fold", [0, 1, 2])
def test_fold_order_many_entries_per_row(gpu, pkg, O, fold, monkeypatch):
    """rows with 3..40 entries inside ONE column block: both folds must reproduce the reference's
    ascending-column order bit for bit (for the ds_add_f64 fold this pins the hardware's
    same-address lane order)"""
    monkeypatch.setenv("SPL_BLOCKED_FOLD", str(fold))
    rng = np.random.default_rng(99)
    n = 20000
    lens = rng.integers(3, 41, size=n)
    rows = np.repeat(np.arange(n), lens)
    base = rng.integers(0, n - 64, size=n)
    cols = np.concatenate([np.sort(b + rng.choice(64, size=l, replace=False)) for b, l in zip(base, lens)])
    vals = rng.normal(size=len(rows)) * 10.0 ** rng.integers(-8, 8, size=len(rows))  # rounding-order sensitive
    A = O.compress(n, n, rows, cols, vals)
    x = rng.normal(size=n)
    yo = O.mulV(A, x)
    M = tuple_to_mat(pkg, A)
    for R, w, u in ((1024, 18, 10), (100, 14, 4), (1221, 16, 12)):
        H = pkg.DeviceMatrix.from_csc(M)
        H.build_blocked(R, w, u)
        H.set_variant(8)
        assert np.array_equal(H.mulv(x), yo), (fold, R, w, u)
