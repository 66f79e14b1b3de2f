#!/usr/bin/env python3
"""Fuzz of the umfpack_zi_* path: random complex sparse systems whose diagonals are dominant in the
real part, in the imaginary part, or mixed row by row (the wrapper's static pivoting inside the 2 x 2
blocks of the embedding), both systems (A x = b, A^H x = b), one and several right-hand sides,
against scipy's complex SuperLU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from __graft_entry__ import load_package
    pkg = load_package()
    U = pkg.umfpack
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    for case in range(ncase):
        n = int(rng.choice([1, 2, 5, 33, 64, 65, 200, 700, 2500]))
        dens = min(1.0, float(rng.choice([1.0, 3.0, 6.0])) / n)
        R = sp.random(n, n, density=dens, random_state=int(rng.integers(1 << 31)), format="csc")
        I = sp.random(n, n, density=dens, random_state=int(rng.integers(1 << 31)), format="csc")
        S = (R + 1j * I).tocsc()
        weight = np.asarray(abs(S).sum(axis=0)).ravel() + 1.0
        kind = case % 4
        if kind == 0:
            d = weight.astype(complex)                      # real parts dominate
        elif kind == 1:
            d = 1j * weight * rng.choice([-1.0, 1.0], n)    # imaginary parts dominate, real part exactly 0
        elif kind == 2:
            d = np.where(rng.integers(0, 2, n) == 1, weight, 1j * weight)  # row by row
        else:
            d = weight * np.exp(1j * rng.uniform(0, 2 * np.pi, n))        # any direction
        S = sp.csc_matrix(S + sp.diags(d))
        S.sort_indices()
        M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
        lu = spla.splu(S)
        fact = U.factor(M, U.analyze(M))
        xs = rng.normal(size=n) + 1j * rng.normal(size=n)
        nrhs = int(rng.choice([1, 3, 9]))
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, S.conj().T.tocsc())):
            b = np.asarray(op @ xs).ravel()
            if nrhs == 1:
                x = U.linearSolve_(fact, mode, M, b)
            else:
                x = U.linearSolveMany_(fact, mode, M, [b * (1.0 + 0.5j * c) for c in range(nrhs)])[nrhs - 1]
                x = x / (1.0 + 0.5j * (nrhs - 1))
            res = np.max(np.abs(op @ x - b)) / (np.max(np.abs(b)) + np.max(np.abs(x)) + 1e-300)
            ref = lu.solve(b, trans="N" if mode == U.UmfpackNormal else "H")
            res_ref = np.max(np.abs(op @ ref - b)) / (np.max(np.abs(b)) + np.max(np.abs(ref)) + 1e-300)
            if not (res <= max(1e-12, 100 * res_ref)):
                bad += 1
                print("FAIL case %d n=%d kind=%d nrhs=%d mode=%d path=%d residual %.2e (SuperLU %.2e)"
                      % (case, n, kind, nrhs, mode, fact.path, res, res_ref), flush=True)
    print("fuzz_complex: %d cases, %d failures" % (ncase, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
