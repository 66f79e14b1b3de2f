#!/usr/bin/env python3
"""From which size of x on does the column-sorted panel image (order-free sums) beat the CSR-stream kernel on matrices
whose rows share no x lines?  random n x n, 20 draws per row, n = 2^17 .. 2^23 (x = 1 .. 64 MB); and R-MAT scale 20."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
ffi = pkg._ffi

def time_spmv(H):
    inf = H.info(); n, nnz = inf["nrows_local"], inf["nnz"]
    s = torch.cuda.current_stream()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    ffi.check("vec", ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), s.cuda_stream))
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(3): H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(s)
    for _ in range(20): H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    B = 12 * nnz + 4 * (n + 1) + 16 * n
    return ms, B / ms / 1e6, H.spmv_kernel()

for lg in (17, 18, 19, 20, 21, 22, 23):
    n = 1 << lg
    row = ["n=2^%d x=%.0f MB" % (lg, n * 8 / 2**20)]
    for what in ("optimize/reference", "optimize/free", "stream", "panel", "blocked"):
        H = pkg.DeviceMatrix.synthetic("random", n, 20)
        try:
            if what == "optimize/free": H.set_spmv_order(H.ORDER_FREE)
            if what.startswith("optimize"): H.optimize()
            elif what == "panel": H.build_panel(); H.set_variant(16)
            elif what == "blocked": H.build_blocked(); H.set_variant(8)
            ms, gbs, k = time_spmv(H)
            row.append("%s: k%d %.4f ms %.0f GB/s" % (what, k, ms, gbs))
        except Exception as e:
            row.append("%s: %s" % (what, e))
        H.free()
    print(" | ".join(row), flush=True)
