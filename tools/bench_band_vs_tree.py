#!/usr/bin/env python3
"""Band LU vs multifrontal LU on the matrices where the choice between them is made (small meshes,
narrow bands): factor + solve seconds of both paths (SPL_LU_METHOD) and what umfpack_di_symbolic
picks by itself.  One JSON line per matrix."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import scipy.sparse as sp
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(0)
    U = pkg.umfpack
    rng = np.random.default_rng(7)

    def grid(m, dim):
        T = sp.diags([-np.ones(m - 1), 2.0 * dim * np.ones(m) / dim, -np.ones(m - 1)], (-1, 0, 1))
        I = sp.identity(m)
        if dim == 2:
            return sp.kron(I, T) + sp.kron(T, I)
        return sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)

    def banded(n, half, ndiag):
        offs = sorted(set(int(o) for o in rng.integers(1, half + 1, ndiag)))
        diags = [np.full(n - o, -1.0) for o in offs]
        A = sp.diags(diags, offs, shape=(n, n))
        A = A + A.T
        return A + sp.diags(np.asarray(abs(A).sum(axis=0)).ravel() + 1.0)

    cases = [("poisson3d 12^3", grid(12, 3)), ("poisson3d 16^3", grid(16, 3)), ("poisson3d 20^3", grid(20, 3)),
             ("poisson2d 64^2", grid(64, 2)), ("poisson2d 80^2", grid(80, 2)), ("poisson2d 100^2", grid(100, 2)),
             ("tridiagonal 1e5", banded(100_000, 1, 1)), ("tridiagonal 1e6", banded(1_000_000, 1, 1)),
             ("10 diagonals within 100, 1e5", banded(100_000, 100, 5)),
             ("20 diagonals within 1000, 1e5", banded(100_000, 1000, 10)),
             ("20 diagonals within 1000, 1e6", banded(1_000_000, 1000, 10))]
    for name, S in cases:
        S = sp.csc_matrix(S)
        S.sort_indices()
        n = S.shape[0]
        A = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
        xs = rng.uniform(0.5, 1.5, n)
        b = S @ xs
        out = {"matrix": name, "n": n, "nnz": int(S.nnz)}
        for method in ("band", "mf", "auto"):
            if method == "auto":
                os.environ.pop("SPL_LU_METHOD", None)
            else:
                os.environ["SPL_LU_METHOD"] = method
            an = U.analyze(A)
            fa = U.factor(A, an)  # first factorisation: kernels loaded, memory mapped
            del fa
            t0 = time.perf_counter(); fa = U.factor(A, an); torch.cuda.synchronize(); t1 = time.perf_counter()
            x = U.linearSolve_(fa, U.UmfpackNormal, A, b); torch.cuda.synchronize(); t2 = time.perf_counter()
            st = fa.stats
            out[method] = {"path": st["path"], "factor_s": round(t1 - t0, 4), "solve_s": round(t2 - t1, 4),
                           "flops": st["flops"], "kl": st["kl"], "fronts": st["fronts"],
                           "max_rel_err": float(np.max(np.abs(x - xs) / np.abs(xs)))}
            del fa, an
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
