#!/usr/bin/env python3
"""phases of umfpack_di_symbolic (SPL_MF_TIMING=1) on a 2-D / 3-D Poisson matrix: analyze_probe.py dim m"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0)
U = pkg.umfpack
dim, m = int(sys.argv[1]), int(sys.argv[2])
H = pkg.DeviceMatrix.synthetic("poisson2d" if dim == 2 else "poisson3d", m)
rp, ci, v = H.export_csr(); H.free()
n = m ** dim
A = pkg.Matrix(n, n, rp, ci, v)
W = pkg.Matrix(4, 4, [0, 1, 2, 3, 4], [0, 1, 2, 3], [1.0, 1.0, 1.0, 1.0])
U.analyze(W)
A._tuple32()
os.environ["SPL_MF_TIMING"] = "1"
for rep in range(2):
    t = time.perf_counter(); an = U.analyze(A); dt = time.perf_counter() - t
    print("== analyze #%d %.3f s" % (rep, dt), file=sys.stderr, flush=True)
