"""feast/tests/test-feast.hs:23-32 — the reference's end-to-end known-answer test through
FEAST -> UMFPACK -> axpy_: eigenvalues of [[2,-1],[-1,2]] on (0,4) are [1,3], closeness
`x == y || |x-y|/|x+y| < 1e-10` (test-feast.hs:14-19).  Here the contour driver is
sparse-linear_amd/feast.py and every sparse operation (complex sparse add, numeric LU per contour
point with ONE symbolic analysis, solves, SpMV) runs on the GPU through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def close_enough(a, b, tol=1e-10):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return a.shape == b.shape and bool(np.all((a == b) | (np.abs(a - b) / np.abs(a + b) < tol)))


def test_feast_reference_fixture(gpu, pkg):
    m = pkg.fromTriples(2, 2, [(0, 0, 2 + 0j), (0, 1, -1 + 0j), (1, 0, -1 + 0j), (1, 1, 2 + 0j)])
    eigenvalues, vectors = pkg.feast.eigSHParams(pkg.feast.FeastParams(feastDebug=False), 2, (0.0, 4.0), m)
    assert len(eigenvalues) == 2                       # "gives the correct number of eigenvalues"
    assert close_enough(eigenvalues, [1.0, 3.0])       # "gives the correct eigenvalues"
    D = pkg.pack(m)
    for lam, v in zip(eigenvalues, vectors.T):
        assert np.max(np.abs(D @ v - lam * v)) < 1e-9


def test_feast_poisson_interior_window(gpu, pkg, O):
    """1-D Laplacian, n = 60: eigenvalues 2 - 2cos(k pi/(n+1)) known in closed form; pick a window with 4"""
    n = 60
    tri = []
    for c in range(n):
        for r_, x in ((c - 1, -1.0), (c, 2.0), (c + 1, -1.0)):
            if 0 <= r_ < n:
                tri.append((r_, c, x))
    A = pkg.fromTriples(n, n, tri)
    exact = 2 - 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
    lo, hi = 0.5 * (exact[9] + exact[10]), 0.5 * (exact[13] + exact[14])
    lam, X = pkg.feast.eigSH(8, (lo, hi), A)
    assert close_enough(np.sort(lam), exact[10:14], 1e-9)


def test_feast_generalized(gpu, pkg):
    rng = np.random.default_rng(4)
    n = 40
    d = np.arange(1.0, n + 1)
    A = pkg.diag(d) + pkg.fromTriples(n, n, [(i, i + 1, 0.1) for i in range(n - 1)] + [(i + 1, i, 0.1) for i in range(n - 1)])
    B = pkg.diag(np.full(n, 2.0))
    lam, X = pkg.feast.geigSH(6, (2.2, 4.3), A, B)
    ref = np.linalg.eigvalsh(pkg.pack(A)) / 2.0
    inside = ref[(ref > 2.2) & (ref < 4.3)]
    assert close_enough(np.sort(lam), inside, 1e-9)


def test_feast_complex_hermitian_uses_transposed_solves(gpu, pkg):
    """A genuinely complex Hermitian matrix: the lower half of the contour comes from UmfpackTrans solves with
    the upper half's factors (ijob 21, Feast.hs:228); eigenvalues against numpy's dense eigvalsh"""
    rng = np.random.default_rng(12)
    n = 50
    tri = [(i, i, float(i + 1)) for i in range(n)]
    for i in range(n - 1):
        z = complex(0.3 * rng.normal(), 0.3 * rng.normal())
        tri += [(i, i + 1, z), (i + 1, i, z.conjugate())]
    A = pkg.fromTriples(n, n, [(r, c, complex(v)) for r, c, v in tri])
    assert A.is_complex and pkg.hermitian(A)
    ref = np.linalg.eigvalsh(pkg.pack(A))
    lo, hi = 0.5 * (ref[11] + ref[12]), 0.5 * (ref[16] + ref[17])
    lam, X = pkg.feast.eigSH(8, (lo, hi), A)
    assert close_enough(np.sort(lam), ref[12:17], 1e-9)
    assert pkg.feast.geigSH_.last_clock["iterations"] <= 12
    D = pkg.pack(A)
    for l, v in zip(lam, X.T):
        assert np.linalg.norm(D @ v - l * v) < 1e-8 * np.linalg.norm(v)


def test_feast_real_and_complex_paths_agree(gpu, pkg):
    """a real symmetric matrix given as Double and as Complex Double with a complex start: conjugation of the
    upper half contour and the transposed solves give the same eigenvalues"""
    n = 30
    tri = [(i, i, 2.0) for i in range(n)] + [(i, i + 1, -1.0) for i in range(n - 1)] + [(i + 1, i, -1.0) for i in range(n - 1)]
    A = pkg.fromTriples(n, n, tri)
    exact = 2 - 2 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
    lo, hi = 0.5 * (exact[4] + exact[5]), 0.5 * (exact[8] + exact[9])
    lam_r, _, _ = pkg.feast.geigSH_(pkg.feast.defaultFeastParams, 8, (lo, hi), A)
    rng = np.random.default_rng(5)
    guess = rng.normal(size=(n, 8)) + 1j * rng.normal(size=(n, 8))
    lam_c, _, _ = pkg.feast.geigSH_(pkg.feast.defaultFeastParams, 8, (lo, hi), A, guess=guess)
    assert close_enough(np.sort(lam_r), exact[5:9], 1e-9) and close_enough(np.sort(lam_c), exact[5:9], 1e-9)


@pytest.mark.parametrize("threads", ["1", "3"])
def test_feast_resident_factors_give_the_same_bits_as_refactoring(gpu, pkg, threads, monkeypatch):
    """round 4: the factors of ze*B - A of every contour point stay in HBM for the later iterations (the points do not
    move); the reference refactors each point in each iteration (Feast.hs:214-218), which SPL_FEAST_KEEP_FACTORS=0
    still does.  The same factors either way: eigenvalues and vectors agree bit for bit, with the contour points
    on worker threads or not, and the later iterations factor nothing."""
    monkeypatch.setenv("SPL_FEAST_THREADS", threads)
    m = 12
    H = pkg.DeviceMatrix.synthetic("poisson3d", m)
    rp, ci, v = H.export_csr()
    H.free()
    n = m ** 3
    A = pkg.Matrix(n, n, rp, ci, v)  # symmetric: its CSR arrays are its CSC arrays
    ref = np.linalg.eigvalsh(pkg.pack(A))
    uniq = np.unique(np.round(ref, 9))  # (a cube: the eigenvalues come in groups of 1, 3, 3, 3, 1, 6 ...)
    lo, hi = 0.5 * (uniq[0] + uniq[1]), 0.5 * (uniq[2] + uniq[3])
    inside = ref[(ref > lo) & (ref < hi)]
    assert len(inside) == 6
    params = pkg.feast.FeastParams(feastContourPoints=4)
    lam1, X1 = pkg.feast.eigSHParams(params, 12, (lo, hi), A)
    c1 = dict(pkg.feast.geigSH_.last_clock)
    monkeypatch.setenv("SPL_FEAST_KEEP_FACTORS", "0")
    lam0, X0 = pkg.feast.eigSHParams(params, 12, (lo, hi), A)
    c0 = dict(pkg.feast.geigSH_.last_clock)
    assert close_enough(np.sort(lam1), inside, 1e-9)
    assert np.array_equal(lam1, lam0) and np.array_equal(X1, X0)
    assert c1["iterations"] == c0["iterations"] >= 2
    assert c0["factorisations"] == 4 * c0["iterations"] and c0["factors_reused"] == 0
    assert c1["factorisations"] == 4 and c1["factors_reused"] == 4 * (c1["iterations"] - 1)


def test_feast_keeps_only_the_factors_that_fit(gpu, pkg, monkeypatch):
    """the factors of a contour point stay resident only while room remains for the points in flight (twice the resident
    bytes of a factorisation each): with the device reporting less free memory than that, every point is refactored in
    every iteration — the reference's way — and the eigenvalues are the same bits."""
    import torch
    monkeypatch.setenv("SPL_FEAST_THREADS", "2")
    m = 12
    H = pkg.DeviceMatrix.synthetic("poisson3d", m)
    rp, ci, v = H.export_csr()
    H.free()
    n = m ** 3
    A = pkg.Matrix(n, n, rp, ci, v)
    ref = np.linalg.eigvalsh(pkg.pack(A))
    uniq = np.unique(np.round(ref, 9))
    lo, hi = 0.5 * (uniq[0] + uniq[1]), 0.5 * (uniq[2] + uniq[3])
    params = pkg.feast.FeastParams(feastContourPoints=4)
    lam1, X1 = pkg.feast.eigSHParams(params, 12, (lo, hi), A)
    kept = dict(pkg.feast.geigSH_.last_clock)
    real_info = torch.cuda.mem_get_info
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (1024, real_info(*a, **k)[1]))
    lam0, X0 = pkg.feast.eigSHParams(params, 12, (lo, hi), A)
    none = dict(pkg.feast.geigSH_.last_clock)
    assert np.array_equal(lam1, lam0) and np.array_equal(X1, X0)
    assert kept["factors_reused"] > 0 and none["factors_reused"] == 0
    assert none["factorisations"] == 4 * none["iterations"]
