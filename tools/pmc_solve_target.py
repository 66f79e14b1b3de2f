#!/usr/bin/env python3
"""target of tools/pmc_solve.sh: factor the 3-D Poisson matrix m^3 once, solve once (prints the solve report)"""
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
A = pkg.Matrix(n, n, rp, ci, v)
S = sp.csr_matrix((v, ci, rp), shape=(n, n))
xs = np.random.default_rng(0xBEEF).uniform(0.5, 1.5, n)
b = S @ xs
fa = U.factor(A, U.analyze(A))
x = U.linearSolve_(fa, U.UmfpackNormal, A, b)
print("SOLVE_REPORT", fa.solve_report, float(np.max(np.abs(x - xs) / xs)), flush=True)
