#!/usr/bin/env python3
"""What reserving CUs is worth when another kernel runs beside the SpMV (one GPU).  The row block of one rank
(--row1 rows of config C2) is multiplied while spl_debug_occupy — few workgroups of steady loads and stores, what
a collective's channel kernels look like to the dispatcher — runs on another stream; the images are laid out for
all CUs and for CUs - k.  Prints one JSON line per (competitor blocks, reserved CUs)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--row1", type=int, default=1_250_000)
    ap.add_argument("--order", default="free")
    ap.add_argument("--reps", type=int, default=40)
    args = ap.parse_args()
    import ctypes as C
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    L = pkg._ffi.lib()
    torch.cuda.set_device(0)
    n, row1 = args.n, args.row1
    main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    pkg._ffi.check("vec", L.spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), main_s.cuda_stream))
    y = torch.zeros(row1, dtype=torch.float64, device="cuda")
    buf = torch.zeros(8 << 20, dtype=torch.float64, device="cuda")   # 64 MB
    handles = {}
    for reserved in (0, 8, 16, 32):
        H = pkg.DeviceMatrix.synthetic("random", n, 20, row1=row1)
        if args.order == "free":
            H.set_spmv_order(H.ORDER_FREE)
        H.set_reserved_cus(reserved)
        H.optimize()
        handles[reserved] = H
    torch.cuda.synchronize()

    def timed(H, blocks):
        for _ in range(3):
            H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=main_s.cuda_stream)
        torch.cuda.synchronize()
        if blocks:
            pkg._ffi.check("occupy", L.spl_debug_occupy(blocks, 512, 30.0 + 0.4 * args.reps, C.c_void_p(buf.data_ptr()), buf.numel(),
                                                        C.c_void_p(side_s.cuda_stream)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        time.sleep(0.005)  # the competitor is resident before the first timed launch
        e0.record(main_s)
        for _ in range(args.reps):
            H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=main_s.cuda_stream)
        e1.record(main_s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    for blocks in (0, 8, 16, 32):
        for reserved, H in handles.items():
            if blocks == 0 and reserved not in (0, 8, 32):
                continue
            ms = timed(H, blocks)
            print(json.dumps({"rows": row1, "order": args.order, "competitor_workgroups": blocks, "reserved_cus": reserved,
                              "kernel": H.spmv_kernel(), "panel_rows": H.info().get("blocked_rows"), "ms": round(ms, 4)}), flush=True)


if __name__ == "__main__":
    main()
