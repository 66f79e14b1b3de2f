#!/usr/bin/env python3
"""Fuzz of the umfpack_di_* paths: random sparse systems (dominant and not, symmetric pattern and not,
several components, empty rows filled on the diagonal) through every LU path — band / multifrontal,
forced cut depths — against scipy's SuperLU.  Prints one line per failure and a summary."""
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
        kind = case % 3
        if case % 5 == 4:
            # mesh-like pattern with random unsymmetric values: deep frontal trees, many fronts
            m = int(rng.integers(20, 140))
            n = m * m
            T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
            P = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
            P.data = rng.uniform(-1.0, 1.0, P.nnz)
            S = P
            dens = 4.0 / n
        else:
            n = int(rng.choice([1, 2, 3, 7, 40, 63, 64, 65, 130, 257, 400, 900]))
            dens = rng.choice([0.5, 2.0, 5.0]) / max(n, 1)
            S = sp.random(n, n, density=min(1.0, dens), random_state=int(rng.integers(1 << 31)), format="csc")
        if kind == 0:      # diagonally dominant by columns
            S = S + sp.diags(np.asarray(abs(S).sum(axis=0)).ravel() + 1.0)
        elif kind == 1:    # symmetric positive definite-ish, not dominant
            S = S + S.T + sp.diags(np.full(n, 0.3 + 2.0 * dens * n))
        else:              # general: diagonal of mixed size, interchanges may be needed
            S = S + sp.diags(rng.choice([1e-9, 0.5, 3.0], n))
        S = sp.csc_matrix(S)
        S.sort_indices()
        xs = rng.uniform(0.5, 1.5, n)
        method = ["band", "mf"][case % 2]
        os.environ["SPL_LU_METHOD"] = method
        os.environ["SPL_MF_CUT"] = str(int(rng.integers(0, 4)))
        # size limits of the front classes (one workgroup / lockstep / per-front pipeline; one-workgroup
        # solves): small values send these small trees through the code of the large ones
        limits = {}
        if rng.integers(0, 2):
            limits = {"SPL_MF_SMALL": str(int(rng.choice([64, 128, 256]))),
                      "SPL_MF_MIDMAX": str(int(rng.choice([128, 512, 4096]))),
                      "SPL_MF_BIGSOLVE": str(int(rng.choice([64, 256, 1024])))}
        for key in ("SPL_MF_SMALL", "SPL_MF_MIDMAX", "SPL_MF_BIGSOLVE"):
            os.environ.pop(key, None)
        os.environ.update(limits)
        nrhs = int(rng.choice([1, 1, 3, 8, 9]))
        M = pkg.Matrix(n, n, S.indptr, S.indices, S.data)
        try:
            lu = spla.splu(S)
            cond_ok = True
        except RuntimeError:
            cond_ok = False  # exactly singular for SuperLU: skip
        if not cond_ok:
            continue
        du = np.abs(lu.U.diagonal())
        if du.min() < 1e-8 * du.max():
            continue  # numerically singular: every LU returns rubbish, nothing to compare
        fact = U.factor(M, U.analyze(M))
        for mode, op in ((U.UmfpackNormal, S), (U.UmfpackTrans, S.T.tocsc())):
            b = np.asarray(op @ xs).ravel()
            if nrhs == 1:
                x = U.linearSolve_(fact, mode, M, b)
            else:  # several right-hand sides through the factors together; the first one is checked
                x = U.linearSolveMany_(fact, mode, M, [b * (1.0 + 0.25 * c) for c in range(nrhs)])[0]
            ref = lu.solve(b, trans="N" if mode == U.UmfpackNormal else "T")
            res = np.max(np.abs(op @ x - b)) / (np.max(np.abs(b)) + np.max(np.abs(x)) + 1e-300)
            res_ref = np.max(np.abs(op @ ref - b)) / (np.max(np.abs(b)) + np.max(np.abs(ref)) + 1e-300)
            if not (res <= max(1e-12, 100 * res_ref)):
                bad += 1
                print("FAIL case %d n=%d kind=%d method=%s cut=%s limits=%s nrhs=%d mode=%d path=%d residual %.2e "
                      "(SuperLU %.2e)" % (case, n, kind, method, os.environ["SPL_MF_CUT"], limits, nrhs, mode, fact.path,
                                          res, res_ref), flush=True)
    print("fuzz: %d cases, %d failures" % (ncase, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
