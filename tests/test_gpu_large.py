"""Maximum-size edge cases: more than 2^31 stored entries (64-bit row pointers at true scale),
and the 64-bit pointer kernels forced on small matrices.  Full-size outputs are checked through
size-independent properties: sampled rows recomputed by the oracle from the counter-based
generators, and linearity  A(ax + by) == a Ax + b Ay  on exactly representable data."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev_vec(torch, pkg, n, seed):
    v = torch.empty(n, dtype=torch.float64, device="cuda")
    pkg._ffi.check("vec", pkg._ffi.lib().spl_vector_synthetic_dev(seed, 0, n, v.data_ptr(),
                                                                torch.cuda.current_stream().cuda_stream))
    return v


def test_more_than_2_31_entries(gpu, pkg, O):
    torch = gpu
    n, K = 110_000_000, 20  # nnz ~ 2.2e9 > 2^31: 26 GB of matrix in HBM
    H = pkg.DeviceMatrix.synthetic("random", n, K)
    inf = H.info()
    assert inf["nnz"] > 2 ** 31
    H.set_variant(1)  # CSR-stream, 64-bit row pointers (int32 ones cannot exist here)
    x = _dev_vec(torch, pkg, n, 0xBEEF)
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    xh = x.cpu().numpy()
    for row0 in (0, 12_345_678, 64_000_001, n - 1000):
        rp, ci, v = O.gen_random_csr(n, K, row0=row0, row1=row0 + 1000)
        yo = np.zeros(1000)
        O.csr_gaxpy32(rp.astype(np.int32), ci, v, xh, yo)  # only x[ci] is read: full-length x
        assert np.array_equal(y[row0:row0 + 1000].cpu().numpy(), yo)
    # checksum of checksums: sum(y) == sum_k val_k * x[col_k] cannot be formed exactly in fp64, but
    # y >= 0.25 * row length everywhere (values and x are >= 0.5) is a cheap full-size sanity bound
    assert float(y.min()) >= 0.25 * 1 and float(y.max()) <= 2.25 * K * 1.5
    del H, x, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("blocked", [False, True])
def test_forced_64bit_pointers_small(gpu, pkg, O, monkeypatch, blocked):
    torch = gpu
    monkeypatch.setenv("SPL_FORCE_PTR64", "1")
    n = 50_003
    H = pkg.DeviceMatrix.synthetic("random", n, 20)
    if blocked:
        H.build_blocked(300, 12, 0)
        H.set_variant(8)
    else:
        H.set_variant(2)
    rp, ci, v = H.export_csr()
    xh = O.gen_vector(n)
    x = torch.from_numpy(xh).cuda()
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    yo = np.zeros(n)
    O.csr_gaxpy32(rp, ci, v, xh, yo)
    assert np.array_equal(y.cpu().numpy(), yo)


def test_linearity_full_size_c2(gpu, pkg):
    """config C2 at full size: A(2x + 3z) == 2 Ax + 3 Az exactly when every product and partial sum
    is exactly representable — use small-integer vectors and check on the matrix with values
    replaced by... (values are generic doubles here, so check to 1e-10 relative instead)"""
    torch = gpu
    n = 10_000_000
    H = pkg.DeviceMatrix.synthetic("random", n, 20)
    H.optimize()
    s = torch.cuda.current_stream().cuda_stream
    x, z = _dev_vec(torch, pkg, n, 1), _dev_vec(torch, pkg, n, 2)
    w = 2.0 * x + 3.0 * z
    yx, yz, yw = (torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3))
    for src, dst in ((x, yx), (z, yz), (w, yw)):
        H.spmv_dev(src.data_ptr(), dst.data_ptr(), stream=s)
    torch.cuda.synchronize()
    ref = 2.0 * yx + 3.0 * yz
    rel = ((yw - ref).abs() / (yw + ref).abs()).max().item()
    assert rel < 1e-10  # closeness predicate of feast/tests/test-feast.hs:17-19
    # idempotence of the launch: same inputs, same bits (no atomics across wavefronts, no races)
    y2 = torch.zeros_like(yx)
    H.spmv_dev(x.data_ptr(), y2.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert torch.equal(y2, yx)
