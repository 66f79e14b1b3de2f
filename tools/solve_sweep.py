#!/usr/bin/env python3
"""solve time of one factorisation under several values of an environment knob that is read per factorisation or per
solve: usage solve_sweep.py m VAR v1,v2,...   (3-D Poisson m^3; the factorisation is redone per value)"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
U = pkg.umfpack
m = int(sys.argv[1]); var = sys.argv[2]; vals = sys.argv[3].split(",")
H = pkg.DeviceMatrix.synthetic("poisson3d", m)
rp, ci, v = H.export_csr(); H.free()
n = m ** 3
A = pkg.Matrix(n, n, rp, ci, v)
S = sp.csr_matrix((v, ci, rp), shape=(n, n))
xs = np.random.default_rng(0xBEEF).uniform(0.5, 1.5, n)
b = S @ xs
an = U.analyze(A)
for val in vals:
    if val == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = val
    fa = U.factor(A, an); torch.cuda.synchronize()
    ts = []
    for rep in range(4):
        t = time.perf_counter(); x = U.linearSolve_(fa, U.UmfpackNormal, A, b); ts.append(time.perf_counter() - t)
    tt = []
    for rep in range(2):
        t = time.perf_counter(); xt = U.linearSolve_(fa, U.UmfpackTrans, A, b); tt.append(time.perf_counter() - t)
    r = fa.solve_report
    print("%s=%s: solve %.1f ms (best of 4), transposed call %.1f ms, walks %d, err %.2e" % (
        var, val, min(ts) * 1e3, min(tt) * 1e3, r["walks"], np.max(np.abs(x - xs) / xs)), flush=True)
    del fa; gc.collect()
