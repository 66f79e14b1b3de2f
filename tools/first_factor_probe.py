#!/usr/bin/env python3
"""First against second factorisation of the same matrix in a fresh process (phase times, SPL_MF_TIMING=1):
where a one-shot `linearSolve` pays more than the steady state.  usage: first_factor_probe.py [m [warm|cold [z [reps]]]]
(z: the complex shifted matrix z I - A of bench.py's f3 entry, through umfpack_zi_*; FFP_REPS: factorisations, default 3)"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_package
pkg = load_package()
torch.cuda.set_device(0)
U = pkg.umfpack
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 2 and sys.argv[2] == "warm":  # warm-up on a small grid first: code objects loaded, streams made
    H = pkg.DeviceMatrix.synthetic("poisson3d", 16)
    rp, ci, v = H.export_csr(); H.free()
    W = pkg.Matrix(4096, 4096, rp, ci, v)
    U.linearSolve_(U.factor(W, U.analyze(W)), U.UmfpackNormal, W, np.ones(4096))
H = pkg.DeviceMatrix.synthetic("poisson3d", m)
rp, ci, v = H.export_csr(); H.free()
n = m ** 3
if len(sys.argv) > 3 and sys.argv[3] == "z":
    import scipy.sparse as sp
    S = sp.csc_matrix((3.0 + 0.5j) * sp.identity(n) - sp.csc_matrix((v, ci, rp), shape=(n, n))); S.sort_indices()
    A = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
else:
    A = pkg.Matrix(n, n, rp, ci, v)
if os.environ.get("FFP_QUIET") != "1":
    os.environ["SPL_MF_TIMING"] = "1"
t0 = time.perf_counter(); an = U.analyze(A); t1 = time.perf_counter()
print("== analyze %.3f s" % (t1 - t0), file=sys.stderr, flush=True)
for rep in range(int(os.environ.get("FFP_REPS", "3"))):
    t = time.perf_counter(); fa = U.factor(A, an); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("== factor #%d %.3f s" % (rep, dt), file=sys.stderr, flush=True)
    del fa; gc.collect()
