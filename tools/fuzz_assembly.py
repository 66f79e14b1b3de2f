#!/usr/bin/env python3
"""Fuzz of the assembly operations of Data.Matrix.Sparse on the device — compress (duplicates, any order, empty
rows / columns), transpose, lin / + / - (real and complex scalars), mulM, mulVT, takeDiag, kronecker, hcat / vcat /
fromBlocks / fromBlocksDiag (real and complex blocks, missing blocks) — on irregular random inputs, every result
compared with the oracle bit for bit (structure and values).  python tools/fuzz_assembly.py [seed] [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from __graft_entry__ import load_package
    from oracle import oracle as O
    pkg = load_package()
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(seed)
    bad = checks = 0

    def eq(M, t):
        return (M.nrows, M.ncols) == (t[0], t[1]) and np.array_equal(M.pointers, t[2]) and np.array_equal(M.indices, t[3]) \
            and np.array_equal(M.values, t[4])

    def tup(M):
        return (M.nrows, M.ncols, M.pointers, M.indices, M.values)

    def coo(nr, nc, cplx=False, ints=False):
        kind = int(rng.integers(0, 4))
        k = int(rng.integers(0, 6 * max(nr, nc) + 1))
        rows, cols = rng.integers(0, nr, k), rng.integers(0, nc, k)
        if kind == 1 and k:   # heavy duplication
            rows, cols = rows % max(1, nr // 7 + 1), cols % max(1, nc // 5 + 1)
        elif kind == 2 and k:  # one crowded column, many empty ones
            cols = np.where(rng.random(k) < 0.7, int(rng.integers(0, nc)), cols)
        vals = rng.integers(-4, 5, k).astype(float) if ints else rng.normal(size=k)
        if cplx:
            vals = vals + 1j * (rng.integers(-4, 5, k) if ints else rng.normal(size=k))
        return rows, cols, vals

    def check(name, ok, info=""):
        nonlocal bad, checks
        checks += 1
        if not ok:
            bad += 1
            print("MISMATCH %s %s" % (name, info), flush=True)

    for case in range(ncase):
        nr = int(rng.choice([1, 2, 7, 64, 65, 300, 2049, 9000]))
        nc = int(rng.choice([1, 3, 64, 129, 700, 5000]))
        cplx = case % 3 == 2
        r, c, v = coo(nr, nc, cplx)
        info = "case %d %dx%d nnz_in %d complex=%d" % (case, nr, nc, len(r), cplx)
        if cplx:  # the oracle's compress is real: both components, the pattern is the same
            re, im = O.compress(nr, nc, r, c, v.real), O.compress(nr, nc, r, c, v.imag)
            A = (nr, nc, re[2], re[3], re[4] + 1j * im[4])
        else:
            A = O.compress(nr, nc, r, c, v)
        M = pkg.compress(nr, nc, r, c, v)
        check("compress", eq(M, A), info)
        if not cplx:
            check("transpose", eq(pkg.transpose(M), O.transpose(A)), info)
            check("takeDiag", np.array_equal(pkg.takeDiag(M), O.take_diag(A)), info)
            x = rng.normal(size=nr)
            got, want = pkg.mulVT(M, x), O.mulV(O.transpose(A), x)
            if np.diff(A[2]).max(initial=0) <= 512:
                check("mulVT", np.array_equal(got, want), info)
            else:  # a column longer than one LDS chunk is summed by a wavefront tree: rounding level
                scale = O.mulV(O.transpose((A[0], A[1], A[2], A[3], np.abs(A[4]))), np.abs(x))
                check("mulVT", bool(np.all(np.abs(got - want) <= 1e-13 * np.maximum(scale, 1e-300) * 64)), info)
            k = int(rng.integers(1, 5))
            Bd = rng.normal(size=(nc, k))
            check("mulM", np.array_equal(pkg.mulM(M, Bd), O.mulM(A, Bd)), info)
        r2, c2, v2 = coo(nr, nc, cplx)
        if cplx:
            re, im = O.compress(nr, nc, r2, c2, v2.real), O.compress(nr, nc, r2, c2, v2.imag)
            A2 = (nr, nc, re[2], re[3], re[4] + 1j * im[4])
        else:
            A2 = O.compress(nr, nc, r2, c2, v2)
        M2 = pkg.Matrix(nc, nr, A2[2], A2[3], A2[4])
        al, be = (complex(rng.normal(), rng.normal()), complex(rng.normal(), rng.normal())) if cplx else (float(rng.normal()), float(rng.normal()))
        ref = O.lin_z(al, A, be, A2) if cplx else O.lin(al, A, be, A2)
        check("lin", eq(pkg.lin(al, M, be, M2), ref), info)
        check("add", eq(M + M2, O.lin_z(1.0, A, 1.0, A2) if cplx else O.add(A, A2)), info)
        check("sub", eq(M - M2, O.lin_z(1.0, A, -1.0, A2) if cplx else O.sub(A, A2)), info)
        # combinators on small blocks
        hs = [int(rng.integers(1, 40)) for _ in range(int(rng.integers(1, 4)))]
        ws = [int(rng.integers(1, 40)) for _ in range(int(rng.integers(1, 4)))]
        blocks_t, blocks_m = [], []
        for i, h in enumerate(hs):
            rt, rm = [], []
            for j, w in enumerate(ws):
                if rng.random() < 0.25 and len(hs) > 1 and len(ws) > 1 and i != j:
                    rt.append(None); rm.append(None)
                    continue
                rr, cc, vv = coo(h, w, cplx)
                if cplx:
                    re, im = O.compress(h, w, rr, cc, vv.real), O.compress(h, w, rr, cc, vv.imag)
                    t = (h, w, re[2], re[3], re[4] + 1j * im[4])
                else:
                    t = O.compress(h, w, rr, cc, vv)
                rt.append(t); rm.append(pkg.Matrix(w, h, t[2], t[3], t[4]))
            blocks_t.append(rt); blocks_m.append(rm)
        try:
            ref = O.fromBlocks(blocks_t)
        except Exception:
            ref = None  # underspecified (a block row / column of missing blocks only): both sides must refuse
        if ref is None:
            try:
                pkg.fromBlocks(blocks_m)
                check("fromBlocks refuses", False, info)
            except Exception:
                check("fromBlocks refuses", True)
        else:
            check("fromBlocks", eq(pkg.fromBlocks(blocks_m), ref), info)
        full_row = [b for b in blocks_m[0] if b is not None]
        check("hcat", eq(pkg.hcat(full_row), O.hcat([tup(b) for b in full_row])), info)
        full_col = [rw[0] for rw in blocks_m if rw[0] is not None]
        if full_col:
            check("vcat", eq(pkg.vcat(full_col), O.vcat([tup(b) for b in full_col])), info)
        if not cplx:
            ka, kb_ = blocks_t[0][0], blocks_t[-1][-1]
            if ka is not None and kb_ is not None:
                check("kronecker", eq(pkg.kronecker(blocks_m[0][0], blocks_m[-1][-1]), O.kronecker(ka, kb_)), info)
        if case % 25 == 24:
            print("... %d cases, %d checks, %d failures" % (case + 1, checks, bad), flush=True)
    print("fuzz_assembly: %d cases, %d checks, %d failures" % (ncase, checks, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
