#!/usr/bin/env python3
"""transpose / compress / lin (rows a9-a11) at scale: one-shot calls through the C ABI on a random matrix of n rows,
20 draws per row (config C2's generator), wall seconds including the PCIe transfers of the host 5-tuples, and the
algorithmic bytes each operation has to move on the device (read the operands + write the result, 12 bytes per
entry + the pointers).  Under `rocprofv3 --kernel-trace --stats -- python3 tools/bench_assembly.py` the kernel
times of the same calls are what `profiles/r02_assembly_kernel_stats.txt` lists."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from __graft_entry__ import load_package
    from oracle import oracle as O
    pkg = load_package()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    rp, ci, v = O.gen_random_csr(n, 20)
    A = O.csr_to_csc_tuple(n, n, rp, ci, v)
    M = pkg.Matrix(n, n, A[2], A[3], A[4])
    nnz = int(A[2][-1])
    rng = np.random.default_rng(0)
    out = {"n": n, "nnz": nnz}

    def timed(name, fn, bytes_):
        fn()
        t = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t
        out[name] = {"wall_s": round(dt, 4), "algorithmic_MB": round(bytes_ / 1e6, 1)}
        return r
    T = timed("transpose", lambda: pkg.transpose(M), 2 * (12 * nnz + 4 * n))
    # COO in random order with 5 % duplicates
    k = nnz + nnz // 20
    pick = rng.integers(0, nnz, k)
    cols = np.repeat(np.arange(n), np.diff(A[2]))
    rows_c, cols_c, vals_c = A[3][pick], cols[pick], A[4][pick]
    C = timed("compress", lambda: pkg.compress(n, n, rows_c, cols_c, vals_c), 16 * k + 12 * nnz)
    L = timed("lin", lambda: pkg.lin(2.0, M, -0.5, T), 2 * 12 * nnz + 12 * 2 * nnz)
    # spot checks against the oracle on a small instance happen in tests/; here only the shapes
    out["nnz_compress"], out["nnz_lin"] = int(C.pointers[-1]), int(L.pointers[-1])
    # the device-resident forms (round 3): handle in, handle out, HIP events around the calls
    import torch
    HM, HT = pkg.DeviceMatrix.from_csc(M), pkg.DeviceMatrix.from_csc(T)
    s = torch.cuda.current_stream()

    def timed_dev(name, fn, bytes_, reps=5):
        fn().free()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        hs = [fn() for _ in range(reps)]
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        for h in hs[1:]:
            h.free()
        out[name] = {"ms": round(ms, 3), "algorithmic_MB": round(bytes_ / 1e6, 1), "GBps_algorithmic": round(bytes_ / ms / 1e6, 1),
                     "hbm_frac": round(bytes_ / ms / 1e6 / 8000.0, 4)}
        return hs[0]
    HL = timed_dev("lin_on_handles", lambda: HM.lin(2.0, HT, -0.5), 2 * 12 * nnz + 12 * 2 * nnz)
    out["nnz_lin_on_handles"] = HL.info()["nnz"]
    timed_dev("transpose_on_handles", lambda: HM.transpose(), 2 * (12 * nnz + 4 * n))
    dr = torch.from_numpy(rows_c.astype(np.int32)).cuda()
    dc = torch.from_numpy(cols_c.astype(np.int32)).cuda()
    dv = torch.from_numpy(vals_c).cuda()
    HC = timed_dev("compress_on_device_triples", lambda: pkg.DeviceMatrix.compress_dev(n, n, k, dr.data_ptr(), dc.data_ptr(), dv.data_ptr()),
                   16 * k + 12 * nnz)
    out["nnz_compress_dev"] = HC.info()["nnz"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
