#!/usr/bin/env python3
"""CPU stand-in for UMFPACK (none exists in this pipeline): scipy's SuperLU on the m^3 7-point Poisson matrix, one core.
usage: superlu_baseline.py m  -> one JSON line (kept under profiles/ for the sizes the bench cannot afford)"""
import json, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
m = int(sys.argv[1])
n = m ** 3
T = sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
I = sp.identity(m)
S = sp.csc_matrix(sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I))
xs = np.random.default_rng(0xBEEF).uniform(0.5, 1.5, n)
b = S @ xs
t = time.perf_counter(); lu = spla.splu(S); tf = time.perf_counter() - t
t = time.perf_counter(); x = lu.solve(b); ts = time.perf_counter() - t
print(json.dumps({"kind": "stand-in: scipy SuperLU splu (COLAMD), 1 core, this container's CPU", "m": m, "n": n, "nnz": int(S.nnz),
                  "factor_s": round(tf, 2), "solve_s": round(ts, 3), "fill_nnz": int(lu.L.nnz + lu.U.nnz),
                  "max_rel_err": float(np.max(np.abs(x - xs) / np.abs(xs)))}))
