"""Host side of the static-pivoting stage (csrc/static_pivot.hpp): the maximum-product transversal and
its scalings, checked on the CPU with a small native driver (tests/native/sp_check.cpp) against
scipy's minimum-weight bipartite matching: the product of the matched entries is maximal, and
B = Dr P A Dc has a unit diagonal and no entry above 1."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import min_weight_full_bipartite_matching

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sp") / "sp_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "sparse-linear_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "sp_check.cpp"), "-o", exe], check=True)
    return exe


def run(checker, S, threads=None):
    S = sp.csc_matrix(S)
    S.sort_indices()
    text = "%d %d\n%s\n%s\n%s\n" % (S.shape[0], S.nnz, " ".join(map(str, S.indptr)), " ".join(map(str, S.indices)),
                                  " ".join(repr(float(v)) for v in S.data))
    env = dict(os.environ)
    if threads is not None:
        env["SPL_SP_THREADS"] = str(threads)
    r = subprocess.run([checker], input=text, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {}
    for line in r.stdout.splitlines():
        k, _, rest = line.partition(" ") if not line.startswith("ok=") else ("ok", "", line[3:])
        out[k] = rest
    return S, out


def check(S, out):
    n = S.shape[0]
    assert out["ok"] == "1"
    rows = np.array(out["rows"].split(), dtype=int)
    assert sorted(rows) == list(range(n))  # a permutation
    dr, dc = np.array(out["dr"].split(), dtype=float), np.array(out["dc"].split(), dtype=float)
    bi, bx = np.array(out["bi"].split(), dtype=int), np.array(out["bx"].split(), dtype=float)
    B = sp.csc_matrix((bx, bi, S.indptr), shape=(n, n))
    assert np.allclose(np.abs(B.diagonal()), 1.0, rtol=1e-10)
    assert np.max(np.abs(bx)) <= 1.0 + 1e-10
    # B really is Dr P A Dc
    P = sp.csc_matrix((np.ones(n), (np.arange(n), rows)), shape=(n, n))  # row j of P A = row rows[j] of A
    ref = sp.diags(dr[rows]) @ (P @ S) @ sp.diags(dc)
    assert abs(ref - B).max() < 1e-12
    # the matching maximises the product of |a| (minimises the sum of -log |a|) over all perfect matchings
    W = S.copy().tocsr()
    W.data = -np.log(np.abs(W.data)) + 50.0  # positive weights; the constant shifts every perfect matching alike
    r_ind, c_ind = min_weight_full_bipartite_matching(W)
    best = np.sum(np.log(np.abs(np.asarray(S.tocsr()[r_ind, c_ind]).ravel())))
    mine = np.sum(np.log(np.abs(np.asarray(S.tocsr()[rows, np.arange(n)]).ravel())))
    assert mine >= best - 1e-9 * max(1.0, abs(best))


@pytest.mark.parametrize("seed,n,dens", [(0, 1, 1.0), (1, 7, 0.5), (2, 60, 0.1), (3, 300, 0.02), (4, 1500, 0.004)])
def test_random_matrices(checker, seed, n, dens):
    rng = np.random.default_rng(seed)
    S = sp.random(n, n, density=dens, random_state=seed, format="csc", data_rvs=lambda k: rng.normal(size=k) * 10.0 ** rng.integers(-6, 6, k))
    S = S + sp.csc_matrix((rng.normal(size=n) * 1e-3, (rng.permutation(n), np.arange(n))), shape=(n, n))  # a hidden transversal
    S = sp.csc_matrix(S)
    S.eliminate_zeros()
    check(*run(checker, S))


def test_mesh_pattern_and_two_by_two_blocks(checker):
    rng = np.random.default_rng(5)
    m = 40
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    S = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
    S.data = rng.uniform(-1.0, 1.0, S.nnz)
    check(*run(checker, S))
    n = 200  # [[1e-14, 3], [3, 1e-14]] blocks: the transversal must take the off-diagonal 3s
    off = np.zeros(n - 1)
    off[0::2] = 3.0
    B = sp.diags([off, np.full(n, 1e-14), off], (-1, 0, 1), format="csc")
    S, out = run(checker, B)
    check(S, out)
    rows = np.array(out["rows"].split(), dtype=int)
    assert np.array_equal(rows, np.arange(n) ^ 1)


def test_searches_side_by_side_give_the_same_optimal_matching(checker):
    """round 4: the shortest-augmenting-path searches of a batch of unmatched columns run on a team of threads against
    the same state, and their results are applied in column order when nothing they read has changed.  What is applied
    depends on the results only: the matching and the scalings are the same, digit for digit, for every team size —
    and optimal (scipy's matching).  A 2-D mesh with a useless diagonal leaves a sixth of the columns to the searches."""
    rng = np.random.default_rng(11)
    m = 110
    T = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], (-1, 0, 1))
    S = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))).tocsc()
    S.data = rng.uniform(-1.0, 1.0, S.nnz)
    S.setdiag(1e-12)
    S, one = run(checker, S, threads=1)
    check(S, one)
    for team in (2, 5):
        _, out = run(checker, S, threads=team)
        assert out == one


def test_structurally_singular_is_refused(checker):
    S = sp.csc_matrix(np.array([[1.0, 2.0, 0.0], [0.0, 0.0, 0.0], [3.0, 4.0, 5.0]]))
    assert run(checker, S)[1]["ok"] == "0"
    S = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 0.0]]) + np.diag([0, 0, 1.0]))
    S[2, 2] = 0.0
    S.eliminate_zeros()
    assert run(checker, sp.csc_matrix(S))[1]["ok"] == "0"
    # two columns whose only entries share one row
    S = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 1.0]]))
    assert run(checker, S)[1]["ok"] == "0"
