"""Committed golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them bit for bit; GPU: the HIP path reproduces them."""
import json
import os

import numpy as np
import pytest

from helpers import tuples_equal

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(G, name)))


def tup(d):
    return (d["nrows"], d["ncols"], np.array(d["pointers"], dtype=np.int64), np.array(d["indices"], dtype=np.int64),
            np.array(d["values"], dtype=np.float64))


def test_oracle_reproduces_golden(O):
    for c in load("hot_path.json")["cases"]:
        A, B, A2 = tup(c["A"]), tup(c["B"]), tup(c["A2"])
        assert tuples_equal(O.compress(A[0], A[1], c["coo"]["rows"], c["coo"]["cols"], c["coo"]["vals"]), A)
        assert np.array_equal(O.mulV(A, c["x"]), np.array(c["mulV"]))
        assert np.array_equal(O.axpy(A, c["x"], c["y0"]), np.array(c["axpy"]))
        assert tuples_equal(O.mm(A, B), tup(c["mm"])) and tuples_equal(O.mm(A, B, literal=True), tup(c["mm"]))
        assert tuples_equal(O.lin(2.0, A, -0.5, A2), tup(c["lin_2_m05"]))
        assert tuples_equal(O.transpose(A), tup(c["transpose"]))
    s = load("solve.json")
    x, st = O.linear_solve(tup(s["poisson2d_m6"]["A"]), s["poisson2d_m6"]["b"])
    assert st == 0 and O.count_not_close(x, np.array(s["poisson2d_m6"]["x_true"]), 1e-10) == 0
    x, _ = O.linear_solve(tup(s["pivot3"]["A"]), s["pivot3"]["b"])
    assert O.count_not_close(x, np.array(s["pivot3"]["x"]), 1e-10) == 0
    g = load("synthetic.json")
    rp, ci, v = O.gen_random_csr(10_000_000, 20, row0=0, row1=3)
    assert rp.tolist() == g["random_1e7_k20_rows0_3"]["rowptr"] and ci.tolist() == g["random_1e7_k20_rows0_3"]["colidx"]
    assert v.tolist() == g["random_1e7_k20_rows0_3"]["val"]
    _, cb, vb = O.gen_banded_csr(10_000_000, row0=5_000_000, row1=5_000_001)
    assert cb.tolist() == g["banded_1e7_row5000000"]["colidx"] and vb.tolist() == g["banded_1e7_row5000000"]["val"]
    assert O.gen_vector(10_000_000, j0=0, j1=4).tolist() == g["x_1e7_first4"]


@pytest.mark.gpu
def test_hip_path_reproduces_golden(gpu, pkg, O):
    def M(d):
        return pkg.Matrix(d["ncols"], d["nrows"], d["pointers"], d["indices"], d["values"])
    for c in load("hot_path.json")["cases"]:
        A, B, A2 = M(c["A"]), M(c["B"]), M(c["A2"])
        assert pkg.compress(A.nrows, A.ncols, c["coo"]["rows"], c["coo"]["cols"], c["coo"]["vals"]) == A
        assert np.array_equal(pkg.mulV(A, c["x"]), np.array(c["mulV"]))
        assert np.array_equal(pkg.axpy(A, c["x"], c["y0"]), np.array(c["axpy"]))
        assert A * B == M(c["mm"])
        assert pkg.lin(2.0, A, -0.5, A2) == M(c["lin_2_m05"])
        assert pkg.transpose(A) == M(c["transpose"])
    s = load("solve.json")
    x = pkg.umfpack.solve(M(s["poisson2d_m6"]["A"]), np.array(s["poisson2d_m6"]["b"]))
    assert O.count_not_close(x, np.array(s["poisson2d_m6"]["x_true"]), 1e-10) == 0
    x = pkg.umfpack.solve(M(s["pivot3"]["A"]), np.array(s["pivot3"]["b"]))
    assert O.count_not_close(x, np.array(s["pivot3"]["x"]), 1e-10) == 0
    f = s["fixtures_from_reference_tests"]
    Fm = pkg.fromTriples(2, 2, [tuple(t) for t in f["test-feast.hs:25"]["triples"]])
    for lam, v in zip(f["test-feast.hs:25"]["eigenvalues"], ([1.0, 1.0], [1.0, -1.0])):
        assert np.array_equal(pkg.mulV(Fm, np.array(v)), lam * np.array(v))
    g = load("synthetic.json")
    H = pkg.DeviceMatrix.synthetic("random", 10_000_000, 20, row0=0, row1=3)
    rp, ci, v = H.export_csr()
    assert rp.tolist() == g["random_1e7_k20_rows0_3"]["rowptr"] and ci.tolist() == g["random_1e7_k20_rows0_3"]["colidx"]
    assert v.tolist() == g["random_1e7_k20_rows0_3"]["val"]


def _cplx(pairs):
    return np.array([complex(a, b) for a, b in pairs], dtype=np.complex128)


def _ctup(d):
    t = tup(d)
    return (t[0], t[1], t[2], t[3], t[4] + 1j * np.array(d["values_im"], dtype=np.float64))


def test_oracle_reproduces_combinators_and_complex(O):
    g = load("combinators_complex.json")
    blocks = [[None if b is None else tup(b) for b in row] for row in g["blocks"]]
    assert tuples_equal(O.fromBlocks(blocks), tup(g["fromBlocks"]))
    assert tuples_equal(O.hcat(blocks[1][1:]), tup(g["hcat_row1_cols12"]))
    assert tuples_equal(O.vcat([blocks[0][0], blocks[2][0]]), tup(g["vcat_col0_rows02"]))
    assert tuples_equal(O.fromBlocksDiag([[blocks[0][0], blocks[1][1], blocks[2][2]], [blocks[0][1], blocks[1][2], None],
                                          [None, None, None]]), tup(g["fromBlocksDiag"]))
    c = g["complex"]
    y = _cplx(c["y0"])
    O.axpy_z(_ctup(c["A"]), _cplx(c["x"]), y)
    assert np.array_equal(y, _cplx(c["axpy"]))


@pytest.mark.gpu
def test_hip_path_reproduces_combinators_and_complex(gpu, pkg):
    def M(d, cplx=False):
        v = np.array(d["values"]) + (1j * np.array(d["values_im"]) if cplx else 0.0)
        return pkg.Matrix(d["ncols"], d["nrows"], d["pointers"], d["indices"], v)
    g = load("combinators_complex.json")
    blocks = [[None if b is None else M(b) for b in row] for row in g["blocks"]]
    assert pkg.fromBlocks(blocks) == M(g["fromBlocks"])
    assert pkg.hcat(blocks[1][1:]) == M(g["hcat_row1_cols12"])
    assert pkg.vcat([blocks[0][0], blocks[2][0]]) == M(g["vcat_col0_rows02"])
    assert pkg.fromBlocksDiag([[blocks[0][0], blocks[1][1], blocks[2][2]], [blocks[0][1], blocks[1][2], None],
                               [None, None, None]]) == M(g["fromBlocksDiag"])
    c = g["complex"]
    A = M(c["A"], cplx=True)
    assert A.is_complex
    assert np.array_equal(pkg.axpy(A, _cplx(c["x"]), _cplx(c["y0"])), _cplx(c["axpy"]))
