import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
ffi = pkg._ffi
for order in ("reference", "free"):
    H = pkg.DeviceMatrix.rmat(20, 32, (0.25, 0.25, 0.25))
    if order == "free": H.set_spmv_order(H.ORDER_FREE)
    H.optimize()
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
    print(order, "kernel", H.spmv_kernel(), "ms", round(ms, 4), "GB/s", round(B / ms / 1e6, 1), flush=True)
    H.free()
for name, build in (("panel default", lambda H: (H.build_panel(), H.set_variant(16))),
                    ("blocked default", lambda H: (H.build_blocked(), H.set_variant(8)))):
    try:
        H = pkg.DeviceMatrix.rmat(20, 32, (0.25, 0.25, 0.25))
        build(H)
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
        print(name, "kernel", H.spmv_kernel(), "ms", round(ms, 4), "GB/s", round(B / ms / 1e6, 1), inf, flush=True)
        H.free()
    except Exception as e:
        print(name, "failed:", e, flush=True)
