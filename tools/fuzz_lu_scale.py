#!/usr/bin/env python3
"""Scale fuzz of the solve step: unsymmetric systems of 1e5 .. 1e6 unknowns on which factoring
without interchanges is NOT safe — weak, tiny or structurally zero diagonals, values over many orders of magnitude —
through umfpack_di_symbolic / numeric / solve (Umfpack.hs:71-102) against scipy's SuperLU (threshold partial
pivoting, the stand-in for UMFPACK in this pipeline).  Families (the patterns are meshes or mesh-like, so that a
direct solver of either kind finishes; the VALUES and the row order are what make them hard):
  perm2d    5-point pattern on an m x m grid, values 10^U(-3, 3) with random signs, rows randomly permuted (the
            diagonal of the permuted matrix is structurally almost empty)
  tiny2d    the same pattern in its natural order with a diagonal of 1e-12 (a useless pivot everywhere)
  kkt2d     saddle point [[K, G^T], [G, 0]]: K a 2-D Laplacian with random positive weights, G one constraint per
            2 x 2 cell patch — a structurally ZERO diagonal block
  conv2d    upwind convection-diffusion at cell Peclet numbers 10^U(0, 4) in random directions: unsymmetric, weakly
            dominant by rows, not by columns
  scaled3d  7-point pattern on an m^3 grid, random unsymmetric values, rows and columns scaled by 10^U(-6, 6)
Per case: the path the library ended on (1 band without interchanges, 2 band with partial pivoting, 3 multifrontal
speculation, 5 static pivoting ...), statuses, componentwise backward error of x and of SuperLU's x, seconds.
Summary: how many cases ended on each path, how many returned a negative status, how many of those SuperLU solved.
python tools/fuzz_lu_scale.py [seed] [cases] [max_unknowns]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid2d(sp, m):
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
    return (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)) + sp.identity(m * m)).tocsr()


def make(sp, rng, family, nmax):
    if family == "perm2d":
        m = int(rng.integers(300, max(301, int(np.sqrt(nmax)))))
        P = grid2d(sp, m).tocoo()
        v = 10.0 ** rng.uniform(-3, 3, P.nnz) * rng.choice([-1.0, 1.0], P.nnz)
        perm = rng.permutation(m * m)
        return sp.csc_matrix((v, (perm[P.row], P.col)), shape=(m * m, m * m))
    if family == "tiny2d":
        m = int(rng.integers(300, max(301, int(np.sqrt(nmax)))))
        P = grid2d(sp, m).tocoo()
        v = rng.uniform(-1.0, 1.0, P.nnz)
        v[P.row == P.col] = 1e-12
        return sp.csc_matrix((v, (P.row, P.col)), shape=(m * m, m * m))
    if family == "kkt2d":
        m = 2 * int(rng.integers(120, max(121, int(np.sqrt(nmax / 1.25)) // 2)))
        n1 = m * m
        P = grid2d(sp, m).tocoo()
        off = P.row != P.col
        w = rng.uniform(0.5, 2.0, P.nnz)
        K = sp.csr_matrix((-w[off], (P.row[off], P.col[off])), shape=(n1, n1))
        K = K + K.T
        K = K + sp.diags(-np.asarray(K.sum(axis=1)).ravel() + 1e-3)
        cells = (m // 2) ** 2
        ci, cj = np.divmod(np.arange(cells), m // 2)
        rows = np.repeat(np.arange(cells), 4)
        cols = np.stack([(2 * ci) * m + 2 * cj, (2 * ci) * m + 2 * cj + 1, (2 * ci + 1) * m + 2 * cj,
                         (2 * ci + 1) * m + 2 * cj + 1], axis=1).ravel()
        G = sp.csr_matrix((rng.uniform(0.5, 1.5, 4 * cells), (rows, cols)), shape=(cells, n1))
        return sp.bmat([[K, G.T], [G, None]], format="csc")
    if family == "conv2d":
        m = int(rng.integers(300, max(301, int(np.sqrt(nmax)))))
        n = m * m
        idx = np.arange(n).reshape(m, m)
        pe = 10.0 ** rng.uniform(0, 4)
        th = rng.uniform(0, 2 * np.pi)
        bx, by = pe * np.cos(th), pe * np.sin(th)
        r, c, v = [], [], []

        def link(a, b, coef):
            r.append(a.ravel()); c.append(b.ravel()); v.append(np.full(a.size, coef))
        link(idx[:, 1:], idx[:, :-1], -1.0 - max(bx, 0.0))   # west
        link(idx[:, :-1], idx[:, 1:], -1.0 - max(-bx, 0.0))  # east
        link(idx[1:, :], idx[:-1, :], -1.0 - max(by, 0.0))
        link(idx[:-1, :], idx[1:, :], -1.0 - max(-by, 0.0))
        r, c, v = np.concatenate(r), np.concatenate(c), np.concatenate(v)
        A = sp.csr_matrix((v, (r, c)), shape=(n, n))
        A = A + sp.diags(-np.asarray(A.sum(axis=1)).ravel() * (1.0 + 1e-6))
        return sp.csc_matrix(A)
    if family == "scaled3d":
        m = int(rng.integers(30, max(31, int(round(nmax ** (1.0 / 3.0))))))
        T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
        I = sp.identity(m)
        P = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I) + sp.identity(m ** 3)).tocoo()
        v = rng.uniform(-1.0, 1.0, P.nnz)
        dr, dc = 10.0 ** rng.uniform(-6, 6, m ** 3), 10.0 ** rng.uniform(-6, 6, m ** 3)
        return sp.csc_matrix((v * dr[P.row] * dc[P.col], (P.row, P.col)), shape=(m ** 3, m ** 3))
    raise ValueError(family)


def backward_error(S, x, b):
    r = np.abs(S @ x - b)
    den = np.abs(S) @ np.abs(x) + np.abs(b)
    ok = den > 0
    return float(np.max(r[ok] / den[ok])) if ok.any() else 0.0


def main():
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from __graft_entry__ import load_package
    pkg = load_package()
    U = pkg.umfpack
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 300_000
    rng = np.random.default_rng(seed)
    families = ["perm2d", "tiny2d", "kkt2d", "conv2d", "scaled3d"]
    paths, negative, negative_but_superlu_ok, inaccurate = {}, 0, 0, 0
    for case in range(ncase):
        fam = families[case % len(families)]
        S = make(sp, rng, fam, nmax)
        S.sort_indices()
        n = S.shape[0]
        xs = rng.uniform(0.5, 1.5, n)
        b = np.asarray(S @ xs).ravel()
        M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
        out = {"case": case, "family": fam, "n": n, "nnz": int(S.nnz)}
        t = time.perf_counter()
        status, x, path, fact = 0, None, None, None
        try:
            fact = U.factor(M, U.analyze(M))
            x = U.linearSolve_(fact, U.UmfpackNormal, M, b)
            path = fact.path
        except Exception as e:  # a negative status surfaces as an exception of the Python mirror (Umfpack.hs:67,81,101)
            status = getattr(e, "status", -1)
            out["error"] = str(e)[:120]
        out["gpu_s"] = round(time.perf_counter() - t, 2)
        out["status"], out["path"] = status, path
        if x is not None:
            out["backward_error"] = backward_error(S, x, b)
        paths[path] = paths.get(path, 0) + 1
        # SuperLU beside it (2-D sizes only: its fill on 3-D meshes of this size takes minutes)
        if fam != "scaled3d" or n <= 40 ** 3:
            t = time.perf_counter()
            try:
                xr = spla.splu(S).solve(b)
                out["superlu_backward_error"] = backward_error(S, xr, b)
                out["superlu_s"] = round(time.perf_counter() - t, 2)
            except Exception as e:
                out["superlu_error"] = str(e)[:80]
        if status < 0 or x is None:
            negative += 1
            if out.get("superlu_backward_error", 1.0) < 1e-10:
                negative_but_superlu_ok += 1
        elif out["backward_error"] > 1e-10:
            inaccurate += 1
        print(json.dumps(out), flush=True)
        del fact
    print(json.dumps({"summary": True, "cases": ncase, "paths": {str(k): v for k, v in paths.items()},
                      "negative_status": negative, "negative_but_superlu_solved": negative_but_superlu_ok,
                      "backward_error_above_1e-10": inaccurate}))


if __name__ == "__main__":
    main()
