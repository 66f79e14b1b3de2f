#!/usr/bin/env python3
"""Indefinite systems through the LU paths: 3-D Laplacian minus sigma I for shifts inside the spectrum.
The factors without interchanges are a speculation there; prints which path each solve ends on (0 =
replaced by band partial pivoting), the time and the accuracy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0); U = pkg.umfpack
def grid(m):
    T = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1)); I = sp.identity(m)
    return sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)
rng = np.random.default_rng(3)
for m in ([int(t) for t in sys.argv[1].split(',')] if len(sys.argv) > 1 else (30, 48)):
    for sigma in (0.0, 0.37, 1.03, 3.1, 6.2):
        S = sp.csc_matrix(grid(m) - sigma * sp.identity(m ** 3)); S.sort_indices(); n = m ** 3
        A = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
        xs = rng.uniform(0.5, 1.5, n); b = S @ xs
        t = time.perf_counter(); fa = U.factor(A, U.analyze(A)); p0 = fa.path
        x = U.linearSolve_(fa, U.UmfpackNormal, A, b); dt = time.perf_counter() - t
        res = np.max(np.abs(S @ x - b)) / (np.max(np.abs(b)) + 12 * np.max(np.abs(x)))
        print("m=%d sigma=%.2f path before %d after %d  %.3f s  rel err %.1e  scaled residual %.1e" % (m, sigma, p0, fa.path, dt, np.max(np.abs(x - xs) / np.abs(xs)), res), flush=True)
