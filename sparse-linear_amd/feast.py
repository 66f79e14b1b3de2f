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
import os
import threading

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


def _positions(union, part):
    """index in `union`'s entry arrays of every entry of `part` (same shape, pattern a subset, both with
    ascending rows inside ascending columns)"""
    if len(part.indices) == len(union.indices) and np.array_equal(part.pointers, union.pointers) and \
            np.array_equal(part.indices, union.indices):
        return np.arange(len(union.indices))  # the same pattern (A's, when B's lies inside it): 60 ms of searches at 80^3

    def keys(m):
        cols = np.repeat(np.arange(m.ncols, dtype=np.int64), np.diff(m.pointers))
        return cols * m.nrows + m.indices
    ku, kp = keys(union), keys(part)
    pos = np.searchsorted(ku, kp)
    if len(kp) and (pos.max() >= len(ku) or not np.array_equal(ku[pos], kp)):
        raise ValueError("geigSH_: pattern of an operand is not inside the pattern of ze*B - A")
    return pos


def _apply(mat, V):
    """rows of the result = mat * rows of V (ijob 30 / 40, Feast.hs:203-208): one SpMV of the hot path per
    subspace vector, device pointers in and out (spl_matrix_spmv_dev), nothing leaves HBM"""
    import torch
    out = torch.empty_like(V)
    h = mat.device_handle()
    stream = torch.cuda.current_stream(V.device).cuda_stream
    for j in range(V.shape[0]):
        h.spmv_dev(V[j].data_ptr(), out[j].data_ptr(), False, stream)
    return out


_pool = None
_pool_workers = 0
_tls = __import__("threading").local()


def _contour_pool():
    """worker threads for the contour points of an iteration (SPL_FEAST_THREADS, default 4 — 80^3, 16 columns, eight
    points per iteration: 8.0 s with two points in flight, 7.2 with three, 7.0 with four; how many actually run side by
    side is capped by what a factorisation holds, see geigSH_ —; 1 = in the calling thread): created once — the library
    keeps a set of streams per host thread that ever factored"""
    global _pool
    k = int(os.environ.get("SPL_FEAST_THREADS", "4"))
    if k <= 1:
        return None
    global _pool_workers
    if _pool is None or _pool_workers != k:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=k, thread_name_prefix="feast-contour")
        _pool_workers = k
    return _pool


def geigSH_(params, m0, interval, matA, matB=None, guess=None):
    """(eigenvalues, eigenvectors, residuals) of A x = lambda B x inside (emin, emax); A (and B)
    Hermitian.  m0 = subspace size, must be >= the number of eigenvalues in the interval.

    Contour: `feastContourPoints` points on the UPPER half circle, as in libfeast (fpm(2) counts the
    half contour): the lower half is served by the same factors — (conj(ze) B - A)^-1 = ((ze B - A)^-1)^H
    for Hermitian A, B — through `UmfpackTrans` solves (ijob 21, Feast.hs:228), or, when A, B and the
    subspace are real, by complex conjugation of the upper half's solutions (the real-symmetric
    driver of libfeast never issues ijob 21).

    The subspace lives in HBM for the whole iteration (torch tensors of shape (m0, n), one vector per row):
    the SpMVs, the batched solves (spl_umfpack_*_solve_many_dev) and the dense products Q^H A Q, Q^H B Q
    read and write device memory; only the m0 x m0 reduced problem and the values of ze*B - A for the
    next factorisation pass through the host."""
    import time
    import torch
    from . import _ffi
    _ffi.require_gpu()
    n = matA.ncols
    if matA.nrows != n:
        raise ValueError("geigSH_: matrix not square")
    if not hermitian(matA):                       # Feast.hs:129
        raise ValueError("geigSH_: matrix A not hermitian")
    if matB is not None and not hermitian(matB):  # Feast.hs:130
        raise ValueError("geigSH_: matrix B not hermitian")
    emin, emax = interval
    dev = torch.device("cuda", torch.cuda.current_device())
    real_problem = (not matA.is_complex or not np.any(matA.values.imag)) and \
                   (matB is None or not matB.is_complex or not np.any(matB.values.imag)) and \
                   (guess is None or not np.any(np.imag(guess)))
    A = _as_complex(matA)
    B = _as_complex(matB) if matB is not None else diag(np.ones(n, dtype=np.complex128))
    # SpMV operands: a real problem keeps a real subspace and real SpMVs (half the bytes)
    opA = cmap(lambda v: np.ascontiguousarray(v.real), A) if real_problem else A
    opB = cmap(lambda v: np.ascontiguousarray(v.real), B) if real_problem else B
    sub_t = torch.float64 if real_problem else torch.complex128
    c, r = 0.5 * (emin + emax), 0.5 * (emax - emin)
    nh = params.feastContourPoints                # points on the half contour
    ne = 2 * nh                                   # full circle, trapezoidal rule
    thetas = np.pi * (np.arange(nh) + 0.5) / nh   # upper half; the lower half is its mirror image
    # the random start of the subspace is drawn ON the device (a seeded Philox stream: the same numbers every run): m0
    # vectors of n normals drawn on the host and copied over were 0.13 s of a 2.8 s solve at 80^3
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)

    def random_rows(k):
        return torch.randn((k, n), generator=gen, device=dev, dtype=torch.float64).to(sub_t)

    if guess is not None:
        g = np.ascontiguousarray(np.asarray(guess).T)   # (n, m0) columns -> rows
        Y = torch.from_numpy(g.real.copy() if real_problem else g.astype(np.complex128)).to(dev).to(sub_t)
    else:
        Y = random_rows(m0)
    # ONE symbolic analysis for every contour point: the pattern of ze*B - A never changes (Feast.hs:212).
    # The pattern (union of the two, Sparse.hs:385-399) and the places of A's and B's entries in it are
    # found once; a contour point only recomputes the values  ze * b - a.
    shifted = lin(-1.0, A, 1.0, B)
    analysis = U.analyze(shifted)
    a_u = np.zeros(len(shifted.values), dtype=np.complex128)
    b_u = np.zeros(len(shifted.values), dtype=np.complex128)
    a_u[_positions(shifted, A)] = A.values
    b_u[_positions(shifted, B)] = B.values
    tol = 10.0 ** (-params.feastTolerance)
    lam, X, res = np.zeros(0), np.zeros((n, 0), dtype=complex), np.zeros(0)
    last_trace = None
    clock = {"values": 0.0, "factor": 0.0, "solve": 0.0, "contour": 0.0, "spmv": 0.0, "dense": 0.0}

    def tick(key, t0):
        torch.cuda.synchronize(dev)
        clock[key] += time.perf_counter() - t0
        return time.perf_counter()

    pool = _contour_pool()
    held = [0.0]  # device bytes of the largest factorisation seen (one list cell: written from the worker threads)
    gate = [None]  # how many contour points may be in flight (set after the first one has shown what it holds)
    # The contour points do not move between iterations, so neither do the factors of ze*B - A.  The reference keeps one
    # factorisation at a time (Feast.hs:214-218: `put (Just (mat, fact))`, a refactorisation per point and iteration —
    # what a CPU's memory allows); 288 GB of HBM hold all eight of a mid-sized problem (80^3: 8 x 10 GB), so the
    # factors of a point stay resident for the later iterations as long as room remains for the work in flight
    # (libfeast's fpm(10) "store factorisations").  Same factors, same bits.  SPL_FEAST_KEEP_FACTORS=0: the reference's way.
    keep_factors = os.environ.get("SPL_FEAST_KEEP_FACTORS", "1") != "0"
    kept = {}          # contour point -> (mat, fact)
    kept_lock = threading.Lock()
    counts = {"factorisations": 0, "factors_reused": 0}
    alloc_s0 = _ffi.device_alloc_seconds()
    for it in range(20):
        t0 = time.perf_counter()
        BY = _apply(opB, Y)                                                    # ijob 40
        rhs = BY if BY.dtype == torch.complex128 else BY.to(torch.complex128)
        t0 = tick("spmv", t0)
        def contour_point(i):
            # One contour point: the factorisation of ze*B - A and the solves with it.  Runs on a worker thread: the
            # points of an iteration are independent, and one mid-sized factorisation is a chain of short launches
            # that leaves most of the device idle — the library's LU entry points work on the calling thread's own
            # streams, so two points in flight overlap (80^3: 1.4x the factorisations, 1.85x the solves per second).
            torch.cuda.set_device(dev)
            if pool is not None and gate[0] is not None:
                with gate[0]:
                    return contour_point_in_flight(i)
            return contour_point_in_flight(i)

        def contour_point_in_flight(i):
            if pool is not None:
                # torch's own work of this point (allocations, the weighted sums) on a stream of this thread: on the
                # legacy default stream every such operation is a barrier for the LU streams of ALL threads.  The
                # library's calls return when their results are complete, and this function ends with a synchronise.
                # (one stream per worker thread AND device: a stream belongs to the device it was made on, and entering it
                # makes that device current — a later call on another device must not inherit it)
                streams = getattr(_tls, "streams", None)
                if streams is None:
                    streams = _tls.streams = {}
                key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
                if key not in streams:
                    streams[key] = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(streams[key]):
                    torch.cuda.set_device(dev)
                    return contour_point_on_this_stream(i)
            return contour_point_on_this_stream(i)

        def contour_point_on_this_stream(i):
            t = time.perf_counter()
            th = thetas[i]
            ze = c + r * np.exp(1j * th)
            w = complex(r * np.exp(1j * th) / ne)
            with kept_lock:
                resident = kept.get(i)
            if resident is not None:
                mat, fact = resident
                t1 = t2 = time.perf_counter()
                with kept_lock:
                    counts["factors_reused"] += 1
            else:
                mat = type(shifted)(n, n, shifted.pointers, shifted.indices, ze * b_u - a_u)   # ijob 10: ze*B - A
                narrowed = getattr(shifted, "_narrowed", None)  # the same pattern arrays: their int32 copies serve every point
                if narrowed is not None and narrowed[0] is mat.pointers and narrowed[1] is mat.indices:
                    mat._narrowed = narrowed
                t1 = time.perf_counter()
                fact = U.factor(mat, analysis)                                 #          numeric LU, same analysis
                held[0] = max(held[0], float(fact.stats["device_bytes"]))
                t2 = time.perf_counter()
                with kept_lock:
                    counts["factorisations"] += 1
                    if keep_factors:
                        # room for the points in flight (panels + transient fronts + the work of their solves: taken
                        # as twice the resident bytes each) must remain after this one stays
                        free_bytes, _ = torch.cuda.mem_get_info(dev)
                        in_flight = _pool_workers if pool is not None else 1
                        if free_bytes >= 2.0 * held[0] * (in_flight + 1):
                            kept[i] = (mat, fact)
            # ijob 11 (Feast.hs:197-201 solves one subspace column at a time): all m0 vectors in one pass
            # through the factors
            qs = U.linearSolveManyDevice_(fact, U.UmfpackNormal, mat, rhs)
            if real_problem:
                part = 2.0 * (qs * w).real
            else:
                qh = U.linearSolveManyDevice_(fact, U.UmfpackTrans, mat, rhs)  # ijob 21: the mirrored point
                part = qs * w + qh * w.conjugate()
                del qh
            del fact, qs, resident
            # (this thread's torch stream only: a device-wide synchronise would wait for the other points in flight)
            torch.cuda.current_stream(dev).synchronize()
            return part, (t1 - t, t2 - t1, time.perf_counter() - t2)

        if pool is not None and held[0] == 0.0:
            # the first point of the run alone: what a factorisation holds decides how many fit side by side (panels +
            # transient fronts + the work of the solves: taken as twice the resident bytes, against 80 % of the device)
            first = contour_point_on_this_stream(0)
            # (mem_get_info: one runtime call; get_device_properties initialises the SMI library first, 27 ms)
            fit = int(0.8 * torch.cuda.mem_get_info(dev)[1] // max(2.0 * held[0], 1.0))
            if fit < 2:
                pool = None
            else:
                gate[0] = threading.BoundedSemaphore(min(fit, _pool_workers))
            rest = [contour_point(i) for i in range(1, nh)] if pool is None else list(pool.map(contour_point, range(1, nh)))
            parts = [first] + rest
        else:
            parts = [contour_point(i) for i in range(nh)] if pool is None else list(pool.map(contour_point, range(nh)))
        Q = torch.zeros((m0, n), dtype=sub_t, device=dev)
        for part, (tv, tf, ts) in parts:   # summed in contour order whatever the threads did: same bits every run
            Q += part
            clock["values"] += tv          # (thread seconds: with several points in flight they add up to more than
            clock["factor"] += tf          # the wall time of the stage, which is "contour")
            clock["solve"] += ts
        del parts
        t0 = tick("contour", t0)
        # Rayleigh-Ritz on the filtered subspace (dense, m0 x m0)
        AQ = _apply(opA, Q)                                                    # ijob 30
        BQ = _apply(opB, Q)                                                    # ijob 40
        t0 = tick("spmv", t0)
        Aq = (Q.conj() @ AQ.T).cpu().numpy()
        Bq = (Q.conj() @ BQ.T).cpu().numpy()
        # drop numerically dependent directions of the subspace
        w_, V = np.linalg.eigh(0.5 * (Bq + Bq.conj().T))
        keep = w_ > 1e-12 * w_.max()
        T = V[:, keep] / np.sqrt(w_[keep])
        Ar_ = T.conj().T @ Aq @ T
        ev, Z = np.linalg.eigh(0.5 * (Ar_ + Ar_.conj().T))
        TZ = T @ Z                                                              # m0 x (kept directions)
        TZd = torch.from_numpy(np.ascontiguousarray(TZ.T)).to(dev).to(sub_t)   # rows = Ritz vectors' coefficients
        Xs = TZd @ Q
        inside = (ev > emin) & (ev < emax)
        lam = ev[inside]
        sel = torch.from_numpy(np.nonzero(inside)[0]).to(dev)
        t0 = tick("dense", t0)
        if len(lam):
            # A X and B X are linear images of what was already multiplied: (T Z)^T (A Q), (T Z)^T (B Q)
            AX, BX = (TZd @ AQ)[sel], (TZd @ BQ)[sel]
            lam_d = torch.from_numpy(lam).to(dev).to(sub_t)[:, None]
            res = (torch.linalg.vector_norm(AX - BX * lam_d, dim=1) /
                   (torch.linalg.vector_norm(BX, dim=1) * max(abs(emin), abs(emax)))).cpu().numpy()
            t0 = tick("dense", t0)
            if params.feastDebug:
                print("feast iteration %d: %d eigenvalues, max residual %.3e" % (it, len(lam), res.max()))
            # libfeast's default stopping test (fpm(6) = 0 in its documentation): the relative change of the
            # trace — the sum of the eigenvalues inside — between two iterations; the residual test is the
            # alternative (fpm(6) = 1) and ends the loop here as well
            trace = float(np.sum(lam))
            if res.max() < tol or (last_trace is not None and abs(trace - last_trace) / max(abs(emin), abs(emax)) < tol):
                break
            last_trace = trace
        else:
            last_trace = None
        Y = Xs if Xs.shape[0] == m0 else torch.cat([Xs, random_rows(m0 - Xs.shape[0])], dim=0)
        Y = Y.contiguous()
    X = Xs[sel].cpu().numpy().T.astype(np.complex128) if len(lam) else np.zeros((n, 0), dtype=complex)
    if params.feastDebug:
        print("feast seconds: " + ", ".join("%s %.3f" % kv for kv in sorted(clock.items())))
    kept.clear()
    # (hipmalloc: seconds inside the driver's allocator — fresh memory another process released a moment ago is wiped
    # first, and the kernels of every thread stand still meanwhile: DESIGN.md 5.5)
    geigSH_.last_clock = dict(clock, iterations=it + 1, hipmalloc=_ffi.device_alloc_seconds() - alloc_s0, **counts)
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
