import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_package
pkg = load_package(); U = pkg.umfpack
m = 1000; n = m*m
T = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
S = ((8.5e-5 + 4e-5j) * sp.identity(n) - A).tocsc(); S.sort_indices()
M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
f = U.factor(M, U.analyze(M))
rng = np.random.default_rng(0)
B = torch.from_numpy(rng.normal(size=(16, n)) + 0j).cuda()
X = U.linearSolveManyDevice_(f, U.UmfpackNormal, M, B)
os.environ["SPL_MF_TIMING"] = "1"
t = time.perf_counter(); X = U.linearSolveManyDevice_(f, U.UmfpackNormal, M, B); torch.cuda.synchronize()
print("solve wall %.1f ms" % ((time.perf_counter()-t)*1e3), file=sys.stderr)
