#!/usr/bin/env python3
"""Diagnostic: which LU path the complex (zi) solves of FEAST-like shifted matrices z*I - A take
(2 = no interchanges kept, 0 = replaced by partial pivoting) and how long they take."""
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
    U = pkg.umfpack
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n = m * m
    T = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
    rng = np.random.default_rng(1)
    xs = rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(0.5, 1.5, n)
    for z in (2 + 0.5j, 4 + 0.01j, 0.1 + 3j, 3.9 + 2j, 8.5 + 0j):
        S = (z * sp.identity(n) - A).tocsc()
        S.sort_indices()
        M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
        b = S @ xs
        t0 = time.perf_counter()
        fact = U.factor(M, U.analyze(M))
        p0 = fact.path
        t1 = time.perf_counter()
        x = U.linearSolve_(fact, U.UmfpackNormal, M, b)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("z=%s path %d -> %d  factor %.3f s solve %.3f s  err %.2e" %
              (z, p0, fact.path, t1 - t0, t2 - t1, np.max(np.abs(x - xs)) / np.max(np.abs(xs))), flush=True)


if __name__ == "__main__":
    main()
