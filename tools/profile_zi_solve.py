#!/usr/bin/env python3
"""one native complex factorisation of z I - A (3-D 7-point Laplacian, m^3) and a few batched solves of 16 right-hand
sides: the workload of a FEAST contour point, for rocprofv3 --kernel-trace --stats.  python tools/profile_zi_solve.py [m] [nrhs] [reps] [trans]   (trans: the conjugate-transposed system)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.sparse as sp
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    trans = len(sys.argv) > 4 and sys.argv[4] == "trans"
    n = m ** 3
    T = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    I = sp.identity(m)
    K = sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)
    S = sp.csc_matrix((0.01 + 0.007j) * sp.identity(n) - K)
    if trans:  # not symmetric any more: plain LU, and the transposed kernels (symmetric factors would take the untransposed ones)
        S = sp.csc_matrix(S + 1e-3 * sp.triu(S, 1))
    S.sort_indices()
    A = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    U = pkg.umfpack
    f = U.factor(A, U.analyze(A))
    rng = np.random.default_rng(1)
    B = torch.from_numpy(rng.normal(size=(k, n)) + 1j * rng.normal(size=(k, n))).cuda()
    mode = U.UmfpackTrans if trans else U.UmfpackNormal
    U.linearSolveManyDevice_(f, mode, A, B)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        X = U.linearSolveManyDevice_(f, mode, A, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("batched %ssolve of %d right-hand sides at %d^3: %.1f ms (stats %s)" % ("conjugate-transposed " if trans else "", k, m, dt * 1e3, f.stats))


if __name__ == "__main__":
    main()
