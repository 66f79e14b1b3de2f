#!/usr/bin/env python3
"""The FEAST-style driver (sparse-linear_amd/feast.py) on a 2-D Laplacian whose eigenvalues are known
in closed form: eigenvalues inside a window, their errors, and the time.  Every contour point is one
complex factorisation (umfpack_zi_numeric, same analysis) and one batched solve of the subspace.
python tools/bench_feast.py --grid 200 --m0 24"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--m0", type=int, default=24)
    ap.add_argument("--lo", type=float, default=0.02)
    ap.add_argument("--hi", type=float, default=0.035)
    ap.add_argument("--dim", type=int, default=2, choices=[2, 3], help="2: m^2 grid (5-point), 3: m^3 grid (7-point)")
    ap.add_argument("--hermitian", action="store_true",
                    help="D A D^H with random unit-modulus D: complex Hermitian with the same eigenvalues, so the lower half "
                         "circle goes through UmfpackTrans solves (ijob 21, Feast.hs:227) on unsymmetric complex factors")
    args = ap.parse_args()
    import numpy as np
    import scipy.sparse as sp
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(0)
    m = args.grid
    n = m ** args.dim
    T = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    I = sp.identity(m)
    if args.dim == 2:
        S = (sp.kron(I, T) + sp.kron(T, I)).tocsc()
    else:
        S = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsc()
    S.sort_indices()
    S = S.astype(np.complex128)
    if args.hermitian:
        d = np.exp(2j * np.pi * np.random.default_rng(7).uniform(size=n))
        H = sp.diags(d) @ S @ sp.diags(d.conj())
        up = sp.triu(H, 1)  # (the lower triangle as the exact conjugate of the upper one: the driver tests A == A^H bitwise)
        S = sp.csc_matrix(up + up.conj().T + sp.diags(H.diagonal().real))
        S.sort_indices()
    A = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
    k = np.arange(1, m + 1)
    ev1 = 2.0 - 2.0 * np.cos(k * np.pi / (m + 1))
    if args.dim == 2:
        exact = np.sort((ev1[:, None] + ev1[None, :]).ravel())
    else:
        exact = np.sort((ev1[:, None, None] + ev1[None, :, None] + ev1[None, None, :]).ravel())
    inside = exact[(exact > args.lo) & (exact < args.hi)]
    t = time.perf_counter()
    lam, X = pkg.feast.eigSH(args.m0, (args.lo, args.hi), A)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    lam = np.sort(np.asarray(lam))
    ok = len(lam) == len(inside)
    err = float(np.max(np.abs(lam - inside) / inside)) if ok and len(lam) else None
    print(json.dumps({"matrix": "%d-D Laplacian %d^%d%s" % (args.dim, m, args.dim, ", complex Hermitian (D A D^H)" if args.hermitian else ""), "n": n, "window": [args.lo, args.hi], "m0": args.m0,
                      "eigenvalues_exact_in_window": len(inside), "found": len(lam), "max_rel_error": err,
                      "seconds": round(dt, 3),
                      "stage_seconds": {k: round(v, 3) for k, v in pkg.feast.geigSH_.last_clock.items()}}), flush=True)


if __name__ == "__main__":
    main()
