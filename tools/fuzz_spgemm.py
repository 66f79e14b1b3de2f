#!/usr/bin/env python3
"""Fuzz of mm (Sparse.hs:691-702) on structures the R-MAT generator never makes: hub columns in A, heavy and empty
columns in B, rows crowded into narrow ranges, rectangular shapes, real and complex values — through every form of
the SpGEMM (automatic choice, ordered single pass in both column shapes, compacting single pass, symbolic + numeric two-pass, split sort
keys), each compared with the oracle bit for bit (structure and values).
python tools/fuzz_spgemm.py [seed] [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FORMS = ({}, {"SPL_SPGEMM_ORDERED": "1", "SPL_SPGEMM_ORDERED_SHAPE": "small"},
         {"SPL_SPGEMM_ORDERED": "1", "SPL_SPGEMM_ORDERED_SHAPE": "large"}, {"SPL_SPGEMM_ORDERED": "0"},
         {"SPL_SPGEMM_TWO_PASS": "1"}, {"SPL_SPGEMM_SPLIT_KEYS": "1"})
KEYS = ("SPL_SPGEMM_ORDERED", "SPL_SPGEMM_ORDERED_SHAPE", "SPL_SPGEMM_TWO_PASS", "SPL_SPGEMM_SPLIT_KEYS")


def pattern(rng, kind, nr, nc, k):
    rows, cols = rng.integers(0, nr, k), rng.integers(0, nc, k)
    if kind == 1:    # hub columns: a few columns hold hundreds to thousands of entries
        for _ in range(int(rng.integers(1, 4))):
            c = int(rng.integers(0, nc))
            m = int(min(nr, rng.integers(100, 3000)))
            rows = np.concatenate([rows, rng.choice(nr, m, replace=False)])
            cols = np.concatenate([cols, np.full(m, c)])
    elif kind == 2:  # rows crowded into a narrow range (bucket overflow in the counting sort)
        rows = np.minimum(nr - 1, rng.integers(0, max(1, nr // 64), k))
    elif kind == 3:  # half of the columns empty
        cols = 2 * (cols // 2)
        cols = np.minimum(cols, nc - 1)
    return rows, cols


def main():
    from __graft_entry__ import load_package
    from oracle import oracle as O
    pkg = load_package()
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(seed)
    bad = calls = 0
    for case in range(ncase):
        m = int(rng.choice([1, 5, 64, 300, 2049, 6000, 20000]))
        n = int(rng.choice([1, 7, 128, 700, 4000, 12000]))
        p = int(rng.choice([1, 3, 100, 900, 5000]))
        ka, kb = int(rng.integers(0, 12 * max(m, n) + 1)), int(rng.integers(0, 8 * max(n, p) + 1))
        cplx = case % 4 == 3
        ra, ca = pattern(rng, case % 4 if not cplx else 1, m, n, ka)
        rb, cb = pattern(rng, (case // 4) % 4, n, p, kb)
        va = rng.normal(size=len(ra))
        vb = rng.normal(size=len(rb))
        A, B = O.compress(m, n, ra, ca, va), O.compress(n, p, rb, cb, vb)
        if cplx:
            A = (A[0], A[1], A[2], A[3], A[4] + 1j * rng.normal(size=len(A[4])))
            B = (B[0], B[1], B[2], B[3], B[4] + 1j * rng.normal(size=len(B[4])))
            ref = O.mm_z(A, B)
        else:
            ref = O.mm(A, B)
        Am, Bm = pkg.Matrix(n, m, A[2], A[3], A[4]), pkg.Matrix(p, n, B[2], B[3], B[4])
        for form in FORMS:
            for k_ in KEYS:
                os.environ.pop(k_, None)
            os.environ.update(form)
            if os.environ.get("SPL_FUZZ_VERBOSE"):
                print("case %d form %s: %dx%d * %dx%d nnz %d %d complex=%d" % (case, form, m, n, n, p, len(A[3]), len(B[3]), cplx),
                      flush=True)
            t_call = time.perf_counter()
            C = pkg.mm(Am, Bm)
            if time.perf_counter() - t_call > 2.0:
                print("slow: case %d form %s: %dx%d * %dx%d nnz %d %d: %.1f s" %
                      (case, form, m, n, n, p, len(A[3]), len(B[3]), time.perf_counter() - t_call), flush=True)
            calls += 1
            ok = (C.nrows, C.ncols) == (ref[0], ref[1]) and np.array_equal(C.pointers, ref[2]) and \
                np.array_equal(C.indices, ref[3]) and np.array_equal(C.values, ref[4])
            if not ok:
                bad += 1
                print("case %d form %s: %dx%d * %dx%d nnz %d %d complex=%d: mismatch (nnz %d vs %d)" %
                      (case, form, m, n, n, p, len(A[3]), len(B[3]), cplx, int(C.pointers[-1]), int(ref[2][-1])), flush=True)
        if case % 20 == 19:
            print("... %d cases, %d products, %d failures" % (case + 1, calls, bad), flush=True)
    for k_ in KEYS:
        os.environ.pop(k_, None)
    print("fuzz_spgemm: %d cases, %d products, %d failures" % (ncase, calls, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
