#!/usr/bin/env python3
"""An UNSYMMETRIC complex system at scale through umfpack_zi_*: z I - K for a 3-D upwind convection-diffusion operator K
on an m^3 grid (not Hermitian, not symmetric, not dominant in the embedding's sense): native complex LU fronts with and
without threshold pivoting inside the diagonal blocks, both systems, componentwise backward error.
python tools/probe/zi_convection_probe.py [m]   (round 3, 80^3: path 4 either way, backward error 3e-16)"""
import sys, time, os
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package(); U = pkg.umfpack
m = int(sys.argv[1]) if len(sys.argv) > 1 else 80
n = m ** 3
rng = np.random.default_rng(4)
I = sp.identity(m)
D2 = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
C = sp.diags([-np.ones(m - 1), np.ones(m)], (-1, 0))  # upwind difference
K = sp.kron(sp.kron(I, I), D2 + 3.0 * C) + sp.kron(sp.kron(I, D2 - 2.0 * C.T), I) + sp.kron(sp.kron(D2 + 1.5 * C, I), I)
S = sp.csc_matrix((0.4 + 0.3j) * sp.identity(n) - K)
S.sort_indices()
A = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
xs = rng.normal(size=n) + 1j * rng.normal(size=n)
for env in ({"SPL_LU_BLOCK_PIVOT": "1"}, {"SPL_LU_BLOCK_PIVOT": "0"}):
    os.environ.update(env)
    t = time.perf_counter(); an = U.analyze(A); f = U.factor(A, an); t1 = time.perf_counter() - t
    st = f.stats
    for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, sp.csc_matrix(S.conj().T))):
        b = np.asarray(op @ xs).ravel()
        t = time.perf_counter(); x = U.linearSolve_(f, mode, A, b); ts = time.perf_counter() - t
        r = np.abs(op @ x - b); den = abs(op) @ np.abs(x) + np.abs(b)
        print(env, "analyze+factor %.3f s, solve %.3f s, path %d -> %d, flags %s, backward error %.2e" % (t1, ts, st["path"], f.path, (st["complex_fronts"], st["block_pivoting"]), float(np.max(r / den))))
