#!/usr/bin/env python3
"""target of tools/pmc_solve.sh: factor the 3-D Poisson matrix m^3 once (`z` as second argument: the complex shifted matrix
z I - A of bench.py's f3 entry, through umfpack_zi_*), solve once (prints the solve report)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
U = pkg.umfpack
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H = pkg.DeviceMatrix.synthetic("poisson3d", m)
rp, ci, v = H.export_csr(); H.free()
n = m ** 3
S = sp.csc_matrix((v, ci, rp), shape=(n, n))
rng = np.random.default_rng(0xBEEF)
if len(sys.argv) > 2 and sys.argv[2] == "z":
    S = sp.csc_matrix((3.0 + 0.5j) * sp.identity(n) - S); S.sort_indices()
    xs = rng.uniform(0.5, 1.5, n) + 1j * rng.uniform(0.5, 1.5, n)
else:
    xs = rng.uniform(0.5, 1.5, n)
A = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
b = np.asarray(S @ xs).ravel()
fa = U.factor(A, U.analyze(A))
x = U.linearSolve_(fa, U.UmfpackNormal, A, b)
print("SOLVE_REPORT", fa.solve_report, float(np.max(np.abs(x - xs) / np.abs(xs))), flush=True)
