#!/usr/bin/env python3
"""What ONE rank of an N-GPU run does, measured on one GPU: the SpMV kernel on the row block
[n*r/N, n*(r+1)/N) of config C2 with the full x (no collective).  Lets the per-rank kernel time
of the 2/4/8-GPU runs be tuned on a 1-GPU box; the all-gather itself needs the real node."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
torch.cuda.set_device(0)
n = 10_000_000
x = torch.empty(n, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream()
pkg._ffi.check("v", pkg._ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), s.cuda_stream))
import argparse  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--ranks", default="1,2,4,8")
ap.add_argument("--order", default="free", choices=["free", "reference"])
ap.add_argument("--blocked", default="", help="R,w,u;R,w,u;... shapes to try instead of the automatic one")
args = ap.parse_args()
shapes = [tuple(int(t) for t in b.split(",")) for b in args.blocked.split(";") if b] or [None]
for N, shape in [(int(t), sh) for t in args.ranks.split(",") for sh in shapes]:
    r0, r1 = 0, n // N
    H = pkg.DeviceMatrix.synthetic("random", n, 20, row0=r0, row1=r1)
    if shape:
        H.build_blocked(*shape)
        H.set_variant(8)
    else:
        if args.order == "free":
            H.set_spmv_order(H.ORDER_FREE)
        H.optimize()
    y = torch.empty(r1 - r0, dtype=torch.float64, device="cuda")
    for _ in range(5):
        H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(50):
        H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    inf = H.info()
    print(json.dumps({"ranks": N, "rows": r1 - r0, "nnz": inf["nnz"], "kernel_ms": round(ms, 4),
                      "kernel": H.spmv_kernel(), "sum_order": args.order,
                      "panel_rows": inf["blocked_rows"], "cols_log2": inf["blocked_cols_log2"]}), flush=True)
    del H
