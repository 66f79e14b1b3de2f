#!/usr/bin/env python3
"""The static-pivoting stage of the LU at scale: a random unsymmetric matrix with a mesh pattern and a
useless diagonal (1e-12) — factors without interchanges fail their check at the first solve, which then
runs the maximum-product transversal (host), orders and factors B = Dr P A Dc on the tree and solves again.
Prints per size: factor (speculation), first solve (including that refactorisation), second solve, the path
the object ends on and the componentwise backward errors of both systems."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import scipy.sparse as sp
    from __graft_entry__ import load_package
    pkg = load_package()
    U = pkg.umfpack
    cases = [(int(t.split("^")[0]), int(t.split("^")[1])) for t in (sys.argv[1:] or ["58^3", "100^3", "1000^2"])]
    for m, dim in cases:
        rng = np.random.default_rng(m)
        T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
        I = sp.identity(m)
        P = (sp.kron(I, T) + sp.kron(T, I)) if dim == 2 else (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I))
        S = sp.csc_matrix(P)
        S.data = rng.uniform(-1.0, 1.0, S.nnz)
        S.setdiag(1e-12 * rng.uniform(0.5, 1.0, S.shape[0]))
        S = sp.csc_matrix(S)
        S.sort_indices()
        n = S.shape[0]
        M = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
        xs = rng.uniform(0.5, 1.5, n)
        b = np.asarray(S @ xs).ravel()
        t0 = time.perf_counter(); an = U.analyze(M); t1 = time.perf_counter()
        fact = U.factor(M, an); t2 = time.perf_counter()
        p0 = fact.path
        x = U.linearSolve_(fact, U.UmfpackNormal, M, b); t3 = time.perf_counter()
        x2 = U.linearSolve_(fact, U.UmfpackNormal, M, b); t4 = time.perf_counter()
        St = sp.csc_matrix(S.T)
        bt = np.asarray(St @ xs).ravel()
        xt = U.linearSolve_(fact, U.UmfpackTrans, M, bt); t5 = time.perf_counter()

        def berr(op, x, b):
            den = abs(op) @ np.abs(x) + np.abs(b)
            return float(np.max(np.abs(op @ x - b) / np.where(den > 0, den, 1.0)))
        print(json.dumps({"grid": "%d^%d" % (m, dim), "n": n, "nnz": int(S.nnz), "analyze_s": round(t1 - t0, 3),
                          "factor_speculative_s": round(t2 - t1, 3), "path_after_factor": p0,
                          "first_solve_incl_static_pivot_refactor_s": round(t3 - t2, 3), "second_solve_s": round(t4 - t3, 3),
                          "transposed_solve_s": round(t5 - t4, 3), "path": fact.path, "stats": fact.stats,
                          "backward_error": berr(S, x, b), "backward_error_transposed": berr(St, xt, bt),
                          "max_rel_err": float(np.max(np.abs(x - xs) / np.abs(xs)))}), flush=True)
        del fact, an


if __name__ == "__main__":
    main()
