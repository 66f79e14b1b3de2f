#!/usr/bin/env python3
"""Complex Double SpMV: the native packed-complex kernel (csrc/spmv_z.hip, 20 B per stored entry) against
the real 2n x 2n embedding it replaces (4 real entries of 12 B per complex entry), device-resident, kernel
time by HIP events; both against the oracle's complex restatement on a sample of rows."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def embedded(pkg, M):
    """the real 2n x 2n matrix with interleaved (re, im) unknowns whose action on packed complex vectors equals
    the complex matrix's: block (i, j) = [[re, -im], [im, re]] — what round 1 multiplied with"""
    import numpy as np
    p, i, x = M.pointers, M.indices, M.values
    lens = np.diff(p)
    nnz = int(p[-1])
    newp = np.concatenate([[0], np.cumsum(np.repeat(2 * lens, 2))]).astype(np.int64)
    col = np.repeat(np.arange(M.ncols), lens)
    t = np.arange(nnz) - p[col]
    idx = np.zeros(4 * nnz, dtype=np.int64)
    val = np.zeros(4 * nnz, dtype=np.float64)
    a = newp[2 * col] + 2 * t
    b = newp[2 * col + 1] + 2 * t
    idx[a], idx[a + 1], idx[b], idx[b + 1] = 2 * i, 2 * i + 1, 2 * i, 2 * i + 1
    val[a], val[a + 1], val[b], val[b + 1] = x.real, x.imag, -x.imag, x.real
    return pkg.Matrix(2 * M.ncols, 2 * M.nrows, newp, idx, val)


def main():
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    from oracle import oracle as O
    pkg = load_package()
    torch.cuda.set_device(0)
    n, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000), 20
    rp, ci, v = O.gen_random_csr(n, K)
    A = O.csr_to_csc_tuple(n, n, rp, ci, v)  # CSC of the synthetic matrix
    rng = np.random.default_rng(1)
    vz = (A[4] + 1j * rng.uniform(0.5, 1.5, len(A[4]))).astype(np.complex128)
    M = pkg.Matrix(n, n, A[2], A[3], vz)
    x = (rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(0.5, 1.5, n)).astype(np.complex128)
    s = torch.cuda.current_stream()
    dx = torch.from_numpy(x.view(np.float64).copy()).cuda()
    out = {"n": n, "nnz": int(A[2][-1])}
    for name, H, xdev, ylen in (("native", pkg.DeviceMatrix.from_csc_complex(M), dx, 2 * n),
                                ("embedding", pkg.DeviceMatrix.from_csc(embedded(pkg, M)), dx, 2 * n)):
        if name == "embedding":
            H.optimize()
        y = torch.zeros(ylen, dtype=torch.float64, device="cuda")
        for _ in range(3):
            H.spmv_dev(xdev.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            H.spmv_dev(xdev.data_ptr(), y.data_ptr(), stream=s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        yz = y.cpu().numpy().view(np.complex128)
        yo = O.mulV_z((n, n, A[2], A[3], vz), x) if name == "native" else yo_keep
        yo_keep = yo
        out[name] = {"ms": round(ms, 4), "GBps_20B_per_entry": round((20.0 * out["nnz"] + 32.0 * n) / ms / 1e6, 1),
                     "bit_identical_to_oracle": bool(np.array_equal(yz, yo)),
                     "max_rel": float(np.max(np.abs(yz - yo) / np.abs(yo)))}
        H.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
