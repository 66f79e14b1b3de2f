import sys, os, time, gc
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package(); torch.cuda.set_device(0); U = pkg.umfpack
for m in [int(a) for a in sys.argv[1:]]:
    H = pkg.DeviceMatrix.synthetic("poisson3d", m); rp, ci, v = H.export_csr(); H.free()
    n = m ** 3
    A = pkg.Matrix(n, n, rp, ci, v)
    an = U.analyze(A)
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); fa = U.factor(A, an); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        fl = fa.stats["flops"]; del fa; gc.collect()
    print("m=%d factor best %.4f s (%s) = %.2f TFLOP/s" % (m, min(ts), " ".join("%.4f" % x for x in ts), fl / min(ts) * 1e-12), flush=True)
