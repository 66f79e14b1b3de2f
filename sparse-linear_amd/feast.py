"""A FEAST-style driver over the GPU hot path — the in-repo CALLER of `analyze` / `factor` /
`linearSolve_` / `lin` / `mulV` (feast/src/Numeric/LinearAlgebra/Feast.hs:115-240).

In the reference the contour-integration algorithm itself lives in the un-vendored Fortran
library libfeast (reverse-communication interface); what the Haskell code contributes, and what
this module mirrors, is the usage pattern of the hot path (Feast.hs:197-233):
  * ONE symbolic analysis of the pattern of ze*B - A             (Feast.hs:212)
  * one numeric factorisation per contour point (same pattern)   (Feast.hs:214-218, ijob 10)
  * many solves per factorisation, one right-hand side at a time (Feast.hs:197-201, ijob 11/21)
  * SpMV with A and B on the subspace columns                     (Feast.hs:203-208, ijob 30/40)
The contour quadrature and the Rayleigh-Ritz step (FEAST's published algorithm, Polizzi 2009) are
restated here in numpy on small dense matrices; every sparse operation goes through the C ABI.
Interface: `eigSH m0 (emin, emax) A`, `geigSH m0 (emin, emax) A B` (Feast.hs:53-72).
"""
import numpy as np

from . import umfpack as U
from .sparse import cmap, diag, hermitian, lin, mulV


class FeastParams(object):
    """FeastParams{feastDebug, feastContourPoints = 8, feastTolerance = 12} (Feast.hs:76-89)"""

    def __init__(self, feastDebug=False, feastContourPoints=8, feastTolerance=12):
        self.feastDebug, self.feastContourPoints, self.feastTolerance = feastDebug, feastContourPoints, feastTolerance


defaultFeastParams = FeastParams()


def _as_complex(m):
    return m if m.is_complex else cmap(lambda v: v.astype(np.complex128), m)


def geigSH_(params, m0, interval, matA, matB=None, guess=None):
    """(eigenvalues, eigenvectors, residuals) of A x = lambda B x inside (emin, emax); A (and B)
    Hermitian.  m0 = subspace size, must be >= the number of eigenvalues in the interval."""
    n = matA.ncols
    if matA.nrows != n:
        raise ValueError("geigSH_: matrix not square")
    if not hermitian(matA):                       # Feast.hs:129
        raise ValueError("geigSH_: matrix A not hermitian")
    if matB is not None and not hermitian(matB):  # Feast.hs:130
        raise ValueError("geigSH_: matrix B not hermitian")
    emin, emax = interval
    A = _as_complex(matA)
    B = _as_complex(matB) if matB is not None else diag(np.ones(n, dtype=np.complex128))
    c, r = 0.5 * (emin + emax), 0.5 * (emax - emin)
    ne = 2 * params.feastContourPoints            # full circle, trapezoidal rule
    thetas = 2.0 * np.pi * (np.arange(ne) + 0.5) / ne
    rng = np.random.default_rng(0)
    Y = guess if guess is not None else rng.normal(size=(n, m0)) + 0j
    # ONE symbolic analysis for every contour point: the pattern of ze*B - A never changes (Feast.hs:212)
    analysis = U.analyze(lin(-1.0, A, 1.0, B))
    tol = 10.0 ** (-params.feastTolerance)
    lam, X, res = np.zeros(0), np.zeros((n, 0), dtype=complex), np.zeros(0)
    for it in range(20):
        BY = np.stack([mulV(B, np.ascontiguousarray(Y[:, j])) for j in range(m0)], axis=1)  # ijob 40
        Q = np.zeros((n, m0), dtype=complex)
        for th in thetas:
            ze = c + r * np.exp(1j * th)
            mat = lin(-1.0, A, 1.0, cmap(lambda v: v * ze, B))    # ijob 10: ze*B - A, sparse add on the GPU
            fact = U.factor(mat, analysis)                          #          numeric LU, same analysis
            # ijob 11 (Feast.hs:197-201 solves one subspace column at a time): all m0 columns in
            # one pass through the factors
            qs = U.linearSolveMany_(fact, U.UmfpackNormal, mat, [BY[:, j] for j in range(m0)])
            for j in range(m0):
                Q[:, j] += (r * np.exp(1j * th) / ne) * qs[j]
        # Rayleigh-Ritz on the filtered subspace (dense, m0 x m0)
        AQ = np.stack([mulV(A, np.ascontiguousarray(Q[:, j])) for j in range(m0)], axis=1)        # ijob 30
        BQ = np.stack([mulV(B, np.ascontiguousarray(Q[:, j])) for j in range(m0)], axis=1)        # ijob 40
        Aq, Bq = Q.conj().T @ AQ, Q.conj().T @ BQ
        # drop numerically dependent directions of the subspace
        w, V = np.linalg.eigh(0.5 * (Bq + Bq.conj().T))
        keep = w > 1e-12 * w.max()
        T = V[:, keep] / np.sqrt(w[keep])
        ev, Z = np.linalg.eigh(0.5 * ((T.conj().T @ Aq @ T) + (T.conj().T @ Aq @ T).conj().T))
        Xs = Q @ (T @ Z)
        inside = (ev > emin) & (ev < emax)
        lam, X = ev[inside], Xs[:, inside]
        if X.shape[1]:
            AX = np.stack([mulV(A, np.ascontiguousarray(X[:, j])) for j in range(X.shape[1])], axis=1)
            BX = np.stack([mulV(B, np.ascontiguousarray(X[:, j])) for j in range(X.shape[1])], axis=1)
            res = np.linalg.norm(AX - BX * lam, axis=0) / (np.linalg.norm(BX, axis=0) * max(abs(emin), abs(emax)))
            if params.feastDebug:
                print("feast iteration %d: %d eigenvalues, max residual %.3e" % (it, len(lam), res.max()))
            if res.max() < tol:
                break
        Y = np.zeros((n, m0), dtype=complex)
        Y[:, :Xs.shape[1]] = Xs
        if Xs.shape[1] < m0:
            Y[:, Xs.shape[1]:] = rng.normal(size=(n, m0 - Xs.shape[1]))
    return lam, X, res


def geigSHParams(params, m0, interval, matA, matB):
    lam, X, _ = geigSH_(params, m0, interval, matA, matB)
    return lam, X


def eigSHParams(params, m0, interval, matA):
    lam, X, _ = geigSH_(params, m0, interval, matA, None)
    return lam, X


def eigSH(m0, interval, matA):
    return eigSHParams(defaultFeastParams, m0, interval, matA)


def geigSH(m0, interval, matA, matB):
    return geigSHParams(defaultFeastParams, m0, interval, matA, matB)
