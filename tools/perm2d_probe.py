#!/usr/bin/env python3
"""Phases of one hard case of tools/fuzz_lu_scale.py (family perm2d: 5-point mesh, values 10^U(-3,3), rows in random
order) through analyze / factor / solve with SPL_MF_TIMING=1, next to scipy's SuperLU.  usage: perm2d_probe.py [m] [family]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
U = pkg.umfpack
m = int(sys.argv[1]) if len(sys.argv) > 1 else 632
fam = sys.argv[2] if len(sys.argv) > 2 else "perm2d"
rng = np.random.default_rng(5)
if fam == "perm2d":
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
    P = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)) + sp.identity(m * m)).tocoo()
    v = 10.0 ** rng.uniform(-3, 3, P.nnz) * rng.choice([-1.0, 1.0], P.nnz)
    perm = rng.permutation(m * m)
    S = sp.csc_matrix((v, (perm[P.row], P.col)), shape=(m * m, m * m))
else:  # mesh3d: random unsymmetric values on the 7-point pattern, a useless diagonal
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
    I = sp.identity(m)
    P = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I) + sp.identity(m ** 3)).tocoo()
    v = rng.uniform(-1.0, 1.0, P.nnz)
    v[P.row == P.col] = 1e-12
    S = sp.csc_matrix((v, (P.row, P.col)), shape=(m ** 3, m ** 3))
S.sort_indices()
n = S.shape[0]
# warm-up (code objects)
W = pkg.Matrix(4, 4, [0, 1, 2, 3, 4], [0, 1, 2, 3], [1.0, 1.0, 1.0, 1.0])
U.linearSolve_(U.factor(W, U.analyze(W)), U.UmfpackNormal, W, np.ones(4))
M = pkg.Matrix(n, n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
xs = rng.uniform(0.5, 1.5, n)
b = np.asarray(S @ xs).ravel()
os.environ["SPL_MF_TIMING"] = "1"
t0 = time.perf_counter(); an = U.analyze(M); t1 = time.perf_counter()
fa = U.factor(M, an); torch.cuda.synchronize(); t2 = time.perf_counter()
x = U.linearSolve_(fa, U.UmfpackNormal, M, b); t3 = time.perf_counter()
x2 = U.linearSolve_(fa, U.UmfpackNormal, M, b); t4 = time.perf_counter()
def be(x):
    r = np.abs(S @ x - b); d = abs(S) @ np.abs(x) + np.abs(b)
    return float(np.max(r / d))
print("== n %d: analyze %.2f factor %.2f first solve %.2f second solve %.3f total %.2f path %d backward error %.1e" % (
    n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t3 - t0, fa.path, be(x)), flush=True)
if os.environ.get("PROBE_SUPERLU", "1") == "1":
    t = time.perf_counter(); lu = spla.splu(S); xc = lu.solve(b); ts = time.perf_counter() - t
    print("== SuperLU %.2f s backward error %.1e" % (ts, be(xc)), flush=True)
