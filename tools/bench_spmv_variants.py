#!/usr/bin/env python3
"""Config C2 (or --n/--draws), one process: time the SpMV kernel variants against each other on the
same device-resident matrix (HIP events on the launch stream) and check each against the
reference-order result of the column-blocked kernel.  Specs, comma separated fields:
  blocked[:R:w:unroll]            column-blocked lockstep (reference order)
  panel[:P:w:unroll:form]         column-sorted workgroup panels (order-free); form 1/2 chunk per load, 4/5 paired
  variant:<k>                     any other spl_matrix_set_variant code
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--draws", type=int, default=20)
    ap.add_argument("--matrix", default="random")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--row1", type=int, default=0, help="only rows [0,row1) (a rank's row block)")
    ap.add_argument("specs", nargs="+")
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(0)
    n = args.n
    row1 = args.row1 or n
    H = pkg.DeviceMatrix.synthetic(args.matrix, n, args.draws, row1=row1)
    nnz = H.info()["nnz"]
    B = 12 * nnz + 4 * (row1 + 1) + 8 * n + 8 * row1
    s = torch.cuda.current_stream()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    pkg._ffi.check("vec", pkg._ffi.lib().spl_vector_synthetic_dev(0xBEEF, 0, n, x.data_ptr(), s.cuda_stream))
    y = torch.zeros(row1, dtype=torch.float64, device="cuda")
    yref = None
    for spec in args.specs:
        f = spec.split(":")
        ints = [int(t) for t in f[1:]]
        try:
            if f[0] == "blocked":
                H.build_blocked(*(ints + [0, 0, 0])[:3])
                H.set_variant(8)
            elif f[0] == "panel":
                H.build_panel(*(ints + [0, 0, 0, 0])[:4])
                H.set_variant(16)
            else:
                H.set_variant(ints[0])
        except Exception as e:
            print(json.dumps({"spec": spec, "error": str(e)}), flush=True)
            continue
        for _ in range(3):
            H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(args.reps):
            H.spmv_dev(x.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out = {"spec": spec, "kernel": H.spmv_kernel(), "ms": round(ms, 4), "GBps": round(B / ms / 1e6, 1),
               "hbm_frac": round(B / ms / 1e6 / 8000.0, 4), "info": [H.info()["blocked_rows"], H.info()["blocked_cols_log2"]]}
        if yref is None:
            yref = y.clone()
        else:
            rel = ((y - yref).abs() / (y + yref).abs().clamp_min(1e-300)).max().item()
            out["max_rel_vs_first"] = rel
            out["bit_identical_to_first"] = bool(torch.equal(y, yref))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
