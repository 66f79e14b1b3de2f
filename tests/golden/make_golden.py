#!/usr/bin/env python3
"""Generates tests/golden/*.json: small input/expected-output vectors for the hot path.

The reference is Haskell and cannot be run in this pipeline (no GHC), so these vectors are
produced by the CPU oracle (oracle/, pinned by the reference's own test properties, see
tests/test_oracle_properties.py) and cross-checked here against closed forms / scipy before
being written.  They freeze the oracle (a later edit that changes any output bit fails
tests/test_golden.py) and give the HIP path fixed known answers.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def mat(m):
    return {"nrows": int(m[0]), "ncols": int(m[1]), "pointers": [int(v) for v in m[2]],
            "indices": [int(v) for v in m[3]], "values": [float(v) for v in m[4]]}


def vec(v):
    return [float(t) for t in v]


def scipy_of(m):
    import scipy.sparse as sp
    return sp.csc_matrix((m[4], m[3], m[2]), shape=(m[0], m[1]))


def main():
    rng = np.random.default_rng(20261003)
    cases = []
    for name, nr, nc, k in (("tiny", 3, 4, 6), ("square", 12, 12, 60), ("tall", 40, 7, 90), ("wide", 6, 50, 120)):
        rows, cols = rng.integers(0, nr, k), rng.integers(0, nc, k)
        vals = np.round(rng.normal(size=k), 3)
        A = O.compress(nr, nc, rows, cols, vals)
        x = np.round(rng.normal(size=nc), 3)
        y0 = np.round(rng.normal(size=nr), 3)
        B = O.compress(nc, 9, rng.integers(0, nc, k), rng.integers(0, 9, k), np.round(rng.normal(size=k), 3))
        A2 = O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), np.round(rng.normal(size=k), 3))
        mulv, axpy, mm, lin, tr = O.mulV(A, x), O.axpy(A, x, y0), O.mm(A, B, literal=True), O.lin(2.0, A, -0.5, A2), O.transpose(A)
        # independent cross-checks before freezing
        assert np.allclose(mulv, scipy_of(A) @ x, rtol=1e-13, atol=1e-13)
        assert np.allclose(scipy_of(mm).toarray(), (scipy_of(A) @ scipy_of(B)).toarray(), rtol=1e-12, atol=1e-12)
        assert np.allclose(scipy_of(lin).toarray(), 2.0 * scipy_of(A).toarray() - 0.5 * scipy_of(A2).toarray())
        assert np.array_equal(scipy_of(tr).toarray(), scipy_of(A).toarray().T)
        cases.append({"name": name, "coo": {"rows": [int(v) for v in rows], "cols": [int(v) for v in cols], "vals": vec(vals)},
                      "A": mat(A), "x": vec(x), "y0": vec(y0), "B": mat(B), "A2": mat(A2),
                      "mulV": vec(mulv), "axpy": vec(axpy), "mm": mat(mm), "lin_2_m05": mat(lin), "transpose": mat(tr)})
    json.dump({"generator": "tests/golden/make_golden.py (CPU oracle; reference cannot be run here)", "cases": cases},
              open(os.path.join(OUT, "hot_path.json"), "w"), indent=0)

    # solve: Poisson 2-D (closed-form eigenpair) and a pivoting case
    m = 6
    rp, ci, v = O.gen_poisson2d_csr(m)
    A = (m * m, m * m, rp, ci.astype(np.int64), v)
    xs = np.round(rng.uniform(0.5, 1.5, m * m), 3)
    b = O.mulV(A, xs)
    x, st = O.linear_solve(A, b)
    assert st == 0 and np.allclose(x, xs, rtol=1e-12)
    P = O.fromTriples(3, 3, [(0, 1, 2.0), (1, 0, 4.0), (2, 2, 5.0), (0, 2, 1.0)])
    bp = np.array([4.0, 8.0, 10.0])
    xp, _ = O.linear_solve(P, bp)
    assert np.allclose(scipy_of(P) @ xp, bp)
    json.dump({"poisson2d_m6": {"A": mat(A), "x_true": vec(xs), "b": vec(b)},
               "pivot3": {"A": mat(P), "b": vec(bp), "x": vec(xp)},
               "fixtures_from_reference_tests": {
                   "test-feast.hs:25": {"triples": [[0, 0, 2.0], [0, 1, -1.0], [1, 0, -1.0], [1, 1, 2.0]], "eigenvalues": [1.0, 3.0]},
                   "Sparse.hs:63-64 hermitian": {"triples": [[0, 0, 2.0], [0, 1, -1.0], [1, 0, -1.0], [1, 1, 2.0]]},
                   "Sparse.hs:67 sigma_x": {"triples": [[0, 1, 1.0], [1, 0, 1.0]]}}},
              open(os.path.join(OUT, "solve.json"), "w"), indent=0)

    # synthetic generators: first rows of C2 / banded / x (what every rank regenerates)
    rp, ci, v = O.gen_random_csr(10_000_000, 20, row0=0, row1=3)
    rb, cb, vb = O.gen_banded_csr(10_000_000, row0=5_000_000, row1=5_000_001)
    json.dump({"random_1e7_k20_rows0_3": {"rowptr": [int(t) for t in rp], "colidx": [int(t) for t in ci], "val": vec(v)},
               "banded_1e7_row5000000": {"colidx": [int(t) for t in cb], "val": vec(vb)},
               "x_1e7_first4": vec(O.gen_vector(10_000_000, j0=0, j1=4))},
              open(os.path.join(OUT, "synthetic.json"), "w"), indent=0)
    # combinators (Sparse.hs:504-597) and Complex Double axpy_ (Sparse.hs:433-457): cross-checked against
    # scipy's bmat / complex product before being frozen
    import scipy.sparse as sp

    def cmat(m):
        d = mat((m[0], m[1], m[2], m[3], np.real(m[4])))
        d["values_im"] = [float(t) for t in np.imag(m[4])]
        return d

    def rnd(nr, nc, k, cplx=False):
        v = np.round(rng.normal(size=k), 3)
        if cplx:
            v = v + 1j * np.round(rng.normal(size=k), 3)
        return O.compress(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), v) if not cplx else \
            _compress_z(nr, nc, rng.integers(0, nr, k), rng.integers(0, nc, k), v)

    def _compress_z(nr, nc, r, c, v):
        re = O.compress(nr, nc, r, c, np.real(v))
        im = O.compress(nr, nc, r, c, np.imag(v))
        assert np.array_equal(re[2], im[2]) and np.array_equal(re[3], im[3])
        return (nr, nc, re[2], re[3], re[4] + 1j * im[4])

    hs, ws = (3, 5, 2), (4, 1, 6)
    blocks = [[rnd(h, w, 2 * max(h, w)) for w in ws] for h in hs]
    blocks[0][2] = None
    blocks[1][0] = None
    blocks[2][1] = None
    fb = O.fromBlocks(blocks)
    dense = sp.bmat([[None if b is None else scipy_of(b) for b in row] for row in blocks]).toarray()
    assert np.array_equal(scipy_of(fb).toarray(), dense)
    hc = O.hcat(blocks[1][1:])
    vc = O.vcat([blocks[0][0], blocks[2][0]])
    assert np.array_equal(scipy_of(hc).toarray(), np.hstack([scipy_of(b).toarray() for b in blocks[1][1:]]))
    assert np.array_equal(scipy_of(vc).toarray(), np.vstack([scipy_of(blocks[0][0]).toarray(), scipy_of(blocks[2][0]).toarray()]))
    fd = O.fromBlocksDiag([[blocks[0][0], blocks[1][1], blocks[2][2]], [blocks[0][1], blocks[1][2], None], [None, None, None]])
    dd = sp.bmat([[scipy_of(blocks[0][0]), scipy_of(blocks[0][1]), None], [None, scipy_of(blocks[1][1]), scipy_of(blocks[1][2])],
                  [None, None, scipy_of(blocks[2][2])]]).toarray()
    assert np.array_equal(scipy_of(fd).toarray(), dd)
    Z = rnd(9, 7, 30, cplx=True)
    xz = np.round(rng.normal(size=7), 3) + 1j * np.round(rng.normal(size=7), 3)
    yz = np.round(rng.normal(size=9), 3) + 1j * np.round(rng.normal(size=9), 3)
    out = yz.copy()
    O.axpy_z(Z, xz, out)
    assert np.allclose(out, sp.csc_matrix((Z[4], Z[3], Z[2]), shape=(9, 7)) @ xz + yz, rtol=1e-13, atol=1e-13)
    json.dump({"blocks": [[None if b is None else mat(b) for b in row] for row in blocks],
               "fromBlocks": mat(fb), "hcat_row1_cols12": mat(hc), "vcat_col0_rows02": mat(vc), "fromBlocksDiag": mat(fd),
               "complex": {"A": cmat(Z), "x": [[float(t.real), float(t.imag)] for t in xz],
                           "y0": [[float(t.real), float(t.imag)] for t in yz],
                           "axpy": [[float(t.real), float(t.imag)] for t in out]}},
              open(os.path.join(OUT, "combinators_complex.json"), "w"), indent=0)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".json")))


if __name__ == "__main__":
    main()
