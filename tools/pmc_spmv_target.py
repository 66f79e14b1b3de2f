#!/usr/bin/env python3
"""target of tools/pmc_spmv_other.sh: the SpMV kernel bench_secondary.spmv_other times — `poisson3d:<m>` or `rmat:<scale>`,
sum order `reference` or `free` — optimize(), then 6 launches (the counters are averaged over the launches of the kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
kind, arg = sys.argv[1].split(":")
order = sys.argv[2] if len(sys.argv) > 2 else "reference"
H = pkg.DeviceMatrix.synthetic("poisson3d", int(arg)) if kind == "poisson3d" else pkg.DeviceMatrix.rmat(int(arg), 32, (0.25, 0.25, 0.25))
if order == "free":
    H.set_spmv_order(H.ORDER_FREE)
H.optimize()
inf = H.info()
n = inf["nrows_local"]
s = torch.cuda.current_stream()
x = torch.empty(n, dtype=torch.float64, device="cuda")
pkg._ffi.check("vec", pkg._ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), s.cuda_stream))
y = torch.zeros(n, dtype=torch.float64, device="cuda")
for _ in range(6):
    H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize()
print("SPMV_TARGET", sys.argv[1], order, "kernel", H.spmv_kernel(), "n", n, "nnz", inf["nnz"], flush=True)
