#!/usr/bin/env python3
"""Config C5 ladder: sparse LU factor + solve of the 3-D 7-point Poisson matrix on an m^3 grid
through the umfpack_di_* ABI on one MI355X, next to a CPU baseline.  No libumfpack exists in
this pipeline, so the CPU baseline is scipy's SuperLU (`splu`), labelled as a stand-in
(BASELINE.md §4).  b = A x* for the synthetic x*; checks |x - x*| and the scaled residual."""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="16,24,32")
    ap.add_argument("--cpu-max", type=int, default=32, help="largest m for the SuperLU stand-in")
    ap.add_argument("--nrhs", type=int, default=0, help="also time a batched solve of this many right-hand sides")
    ap.add_argument("--dim", type=int, default=3, choices=[2, 3], help="3: m^3 grid, 7-point; 2: m^2 grid, 5-point")
    ap.add_argument("--shift", type=complex, default=None,
                    help="factor the complex symmetric z I - A (a FEAST contour point) through umfpack_zi_* instead of A")
    args = ap.parse_args()
    import numpy as np
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(0)
    U = pkg.umfpack
    # warm-up on an 8^3 grid: the first launch of each kernel pays the one-time code-object load
    def poisson(m, dim):  # generated in HBM by the product's own generator (include/spl_synth.h), exported once
        H = pkg.DeviceMatrix.synthetic("poisson3d" if dim == 3 else "poisson2d", m)
        rp, ci, v = H.export_csr()  # symmetric: CSR arrays == CSC arrays
        H.free()
        return rp, ci, v

    rp, ci, v = poisson(8, 3)
    W = pkg.Matrix(512, 512, rp, ci, v)
    U.linearSolve_(U.factor(W, U.analyze(W)), U.UmfpackNormal, W, np.ones(512))
    for m in [int(t) for t in args.grid.split(",")]:
        n = m ** args.dim
        rp, ci, v = poisson(m, args.dim)
        S = sp.csc_matrix((v, ci, rp), shape=(n, n))  # symmetric: CSR arrays == CSC arrays
        xs = np.random.default_rng(0xBEEF).uniform(0.5, 1.5, n)  # manufactured solution, no cancellation
        if args.shift is not None:
            S = sp.csc_matrix(args.shift * sp.identity(n) - S)
            S.sort_indices()
            rp, ci, v = S.indptr, S.indices, S.data
            xs = xs + 1j * np.random.default_rng(0xFEED).uniform(0.5, 1.5, n)
        A = pkg.Matrix(n, n, rp, ci, v)
        b = S @ xs
        # Each size starts from an idle device, as in a fresh process: the blocks the previous size left
        # in the library's pool go back to the driver, and the driver's background wipe of released
        # memory (~40 GB/s when it runs, tools/probe/malloc_probe.hip) is over before the clock starts.
        released = pkg._ffi.release_cached_memory()
        time.sleep(0.5 + released / 10e9)
        t0 = time.perf_counter(); an = U.analyze(A); t1 = time.perf_counter()
        fa = U.factor(A, an); torch.cuda.synchronize(); t2 = time.perf_counter()
        x = U.linearSolve_(fa, U.UmfpackNormal, A, b); torch.cuda.synchronize(); t3 = time.perf_counter()
        err = float(np.max(np.abs(x - xs) / np.abs(xs)))
        res = float(np.max(np.abs(S @ x - b)) / (np.max(np.abs(b)) + 6 * np.max(np.abs(x))))
        out = {"metric": "sparse LU factor+solve seconds", "dim": args.dim, "m": m, "n": n, "nnz": int(rp[-1]),
               "shift": None if args.shift is None else str(args.shift),
               "gpu": {"analyze_s": round(t1 - t0, 3), "factor_s": round(t2 - t1, 3), "solve_s": round(t3 - t2, 3),
                       "total_s": round(t3 - t0, 3)},
               "max_rel_err_vs_manufactured": err, "scaled_residual": res, "within_1e-10": bool(err < 1e-10)}
        st = fa.stats
        rep = fa.solve_report
        out["solve"] = {"walks": rep["walks"], "refinement_steps": rep["ir_attempted"], "backward_error": rep["backward_error"],
                        "GB_per_walk": round(rep["walk_bytes"] * 1e-9, 3),
                        "TB_per_s": round(rep["walks"] * rep["walk_bytes"] / max(t3 - t2, 1e-9) * 1e-12, 3)}
        out["factorisation"] = {"path": st["path"], "kl": st["kl"], "ku": st["ku"], "fronts": st["fronts"],
                                "device_GB": round(st["device_bytes"] * 1e-9, 2), "flops": st["flops"],
                                "TFLOP_per_s": round(st["flops"] / max(t2 - t1, 1e-9) * 1e-12, 2)}
        bt = b if args.shift is None else np.asarray(S.conj().T @ xs).ravel()
        t = time.perf_counter(); xt = U.linearSolve_(fa, U.UmfpackTrans, A, bt); tt = time.perf_counter() - t
        out["gpu"]["solve_transposed_s"] = round(tt, 3)  # symmetric matrix: same system, U^T / L^T kernels
        out["transposed_max_rel_err"] = float(np.max(np.abs(xt - xs) / np.abs(xs)))
        if args.nrhs > 1:
            bs = [b * (1.0 + 0.01 * c) for c in range(args.nrhs)]
            t = time.perf_counter(); xm = U.linearSolveMany_(fa, U.UmfpackNormal, A, bs); tm = time.perf_counter() - t
            t = time.perf_counter()
            xo = [U.linearSolve_(fa, U.UmfpackNormal, A, bb) for bb in bs]
            to = time.perf_counter() - t
            out["batched_solve"] = {"nrhs": args.nrhs, "together_s": round(tm, 3), "one_at_a_time_s": round(to, 3),
                                    "max_rel_diff": float(max(np.max(np.abs(p - q)) / np.max(np.abs(q))
                                                              for p, q in zip(xm, xo)))}
        if m <= args.cpu_max:
            t = time.perf_counter(); lu = spla.splu(S); tf = time.perf_counter() - t
            t = time.perf_counter(); xc = lu.solve(b); ts = time.perf_counter() - t
            out["cpu_baseline"] = {"kind": "stand-in: scipy SuperLU splu (no libumfpack in this pipeline)", "cores": 1,
                                   "factor_s": round(tf, 3), "solve_s": round(ts, 3),
                                   "fill_nnz": int(lu.L.nnz + lu.U.nnz),
                                   "max_rel_err": float(np.max(np.abs(xc - xs) / np.abs(xs)))}
        # the same matrix factored again with the same analysis (what FEAST does per contour point): its
        # panels and fronts come back from the pool
        del fa
        gc.collect()
        t = time.perf_counter(); fa = U.factor(A, an); torch.cuda.synchronize(); tr = time.perf_counter() - t
        out["gpu"]["refactor_s"] = round(tr, 3)
        print(json.dumps(out), flush=True)
        del fa, an
        gc.collect()


if __name__ == "__main__":
    main()
