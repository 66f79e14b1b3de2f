#!/usr/bin/env python3
"""Per-level times of one tree solve (SPL_MF_TIMING=1 around the solve only): 3-D Poisson on an m^3 grid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
xs = np.random.default_rng(0xBEEF).uniform(0.5, 1.5, n)
import scipy.sparse as sp
S = sp.csr_matrix((v, ci, rp), shape=(n, n))
b = S @ xs
t0 = time.perf_counter(); an = U.analyze(A); t1 = time.perf_counter()
fa = U.factor(A, an); torch.cuda.synchronize(); t2 = time.perf_counter()
x = U.linearSolve_(fa, U.UmfpackNormal, A, b); t3 = time.perf_counter()
x = U.linearSolve_(fa, U.UmfpackNormal, A, b); t4 = time.perf_counter()
xt = U.linearSolve_(fa, U.UmfpackTrans, A, b); t5 = time.perf_counter()
print("analyze %.3f factor %.3f solve(first) %.3f solve %.3f solveT %.3f err %.2e errT %.2e" % (
    t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, np.max(np.abs(x - xs) / xs), np.max(np.abs(xt - xs) / xs)), flush=True)
print(fa.stats, fa.solve_report, flush=True)
os.environ["SPL_MF_TIMING"] = "1"
x = U.linearSolve_(fa, U.UmfpackNormal, A, b)
if len(sys.argv) > 2:
    xt = U.linearSolve_(fa, U.UmfpackTrans, A, b)
