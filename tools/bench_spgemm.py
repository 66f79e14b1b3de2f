#!/usr/bin/env python3
"""Config C4: SpGEMM A*A for a 2^20 x 2^20 R-MAT CSR (~32 nnz/row) on one MI355X.
Reports products/s and GB/s with B = 12*(nnz(A) + products + nnz(C)), next to
the CPU oracle's touched-list restatement of mm (Sparse.hs:691-702) on a bounded column sample,
whose columns are also compared with the GPU result (structure and values, exact)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edge-factor", type=int, default=32)
    ap.add_argument("--abc", default="0.25,0.25,0.25", help="R-MAT a,b,c (d = 1-a-b-c); ER by default")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu-cols", type=int, default=4096)
    args = ap.parse_args()
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(0)
    abc = tuple(float(t) for t in args.abc.split(","))
    n = 1 << args.scale
    t = time.perf_counter()
    H = pkg.DeviceMatrix.rmat(args.scale, args.edge_factor, abc)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t
    nnzA = H.info()["nnz"]
    times = []
    HC = None
    for _ in range(args.reps):
        HC = None
        torch.cuda.synchronize()
        t = time.perf_counter()
        HC, products = H.spgemm(H)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    tt = sorted(times)[len(times) // 2]
    nnzC = HC.info()["nnz"]
    B = 12 * (nnzA + products + nnzC)
    out = {"metric": "SpGEMM A*A products/s", "value": round(products / tt / 1e9, 3), "unit": "Gproducts/s",
           "config": {"workload": "R-MAT scale %d, edge factor %d, (a,b,c)=%s" % (args.scale, args.edge_factor, abc),
                      "n": n, "nnzA": nnzA, "products": products, "nnzC": nnzC, "compression": round(products / max(nnzC, 1), 2)},
           "seconds": round(tt, 4), "GBps_algorithmic": round(B / tt / 1e9, 1), "hbm_frac": round(B / tt / 8e12, 4),
           "generate_compress_seconds": round(t_gen, 3), "dtype": "f64"}
    # CPU baseline + parity on a bounded sample: the first cpu_cols columns of C^T ... i.e. rows of C
    if args.cpu_cols > 0:
        from oracle import oracle as O
        rp, ci, v = H.export_csr()
        k = min(args.cpu_cols, n)
        # rows 0..k of C = A[0:k,:] * A ; in the reference's CSC terms: columns 0..k of C^T = A^T * (A^T)[:, 0:k]
        At = (n, n, rp, ci.astype(np.int64), v)  # CSR(A) arrays == CSC(A^T)
        Bs = (n, k, rp[:k + 1], ci[:rp[k]].astype(np.int64), v[:rp[k]])
        t = time.perf_counter()
        Cs = O.mm(At, Bs)
        t_cpu = time.perf_counter() - t
        crp, cci, cv = HC.export_csr()
        ok = (np.array_equal(crp[:k + 1], Cs[2]) and np.array_equal(cci[:crp[k]], Cs[3]) and np.array_equal(cv[:crp[k]], Cs[4]))
        lens = np.diff(rp)
        prod_s = int(np.sum(lens[ci[:rp[k]]]))
        out["cpu_baseline"] = {"value": round(prod_s / t_cpu / 1e9, 4), "unit": "Gproducts/s", "cores": 1, "kind": "port",
                               "sample": "rows 0..%d of C (%d products), oracle touched-list mm (Sparse.hs:691-702), %.2f s" % (k, prod_s, t_cpu)}
        out["parity"] = {"rows_checked": k, "structure_and_values_bit_identical": bool(ok)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
