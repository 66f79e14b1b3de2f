"""Host side of the multifrontal LU (csrc/mf_symbolic.hpp): the nested-dissection ordering and the
frontal tree are plain C++, so their invariants are checked on the CPU with a small native driver
(tests/native/mf_check.cpp): permutation, post-order, boundary lists ascending / owned by ancestors /
passed on to the parent, every entry of A + A^T inside the front of its earlier-eliminated index."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("mf") / "mf_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "sparse-linear_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "mf_check.cpp"), "-o", exe], check=True)
    return exe


def run(checker, S, leaf, team=None, mult=1):
    S = sp.csc_matrix(S)
    S.sort_indices()
    text = "%d %d\n%s\n%s\n" % (S.shape[0], S.nnz, " ".join(map(str, S.indptr)), " ".join(map(str, S.indices)))
    env = dict(os.environ) if team is None else dict(os.environ, SPL_ND_TEAM=str(team))
    r = subprocess.run([checker, str(leaf), str(mult)], input=text, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return dict(kv.split("=") for kv in r.stdout.split())


def poisson(m, dim):
    T = sp.diags([-np.ones(m - 1), 2 * np.ones(m), -np.ones(m - 1)], (-1, 0, 1))
    I = sp.identity(m)
    if dim == 2:
        return sp.kron(I, T) + sp.kron(T, I)
    return sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)


@pytest.mark.parametrize("m,dim,leaf", [(30, 2, 64), (60, 2, 256), (12, 3, 64), (24, 3, 256)])
def test_tree_invariants_on_grids(checker, m, dim, leaf):
    out = run(checker, poisson(m, dim), leaf)
    assert int(out["bad"]) == 0 and int(out["fronts"]) > 1
    # nested dissection beats the band: the largest front is far smaller than n
    assert int(out["maxfront"]) < int(out["n"]) // 2


def test_tree_invariants_on_irregular_patterns(checker):
    rng = np.random.default_rng(4)
    n = 3000
    # unsymmetric random pattern (the tree is built on A + A^T), with isolated vertices and two components
    rows, cols = rng.integers(0, n // 2, 6000), rng.integers(0, n // 2, 6000)
    rows2, cols2 = rng.integers(n // 2, n - 50, 4000), rng.integers(n // 2, n - 50, 4000)
    S = sp.coo_matrix((np.ones(10000), (np.concatenate([rows, rows2]), np.concatenate([cols, cols2]))), shape=(n, n))
    out = run(checker, S + sp.identity(n), 64)
    assert int(out["bad"]) == 0
    # a dense matrix cannot be dissected: one front
    out = run(checker, np.ones((40, 40)), 8)
    assert int(out["bad"]) == 0 and int(out["fronts"]) == 1
    # 1 x 1 and diagonal matrices
    assert int(run(checker, sp.identity(1), 64)["bad"]) == 0
    assert int(run(checker, sp.identity(500), 64)["bad"]) == 0


def test_team_traversals_reproduce_the_sequential_tree(checker):
    """regions of 200 000 vertices and more build their level structures with a team of threads
    (team_bfs): the ordering and the tree must be those of one thread, whatever the team"""
    for A in (poisson(62, 3),    # 238 328 vertices: wide levels, expanded by the whole team
              poisson(460, 2)):  # 211 600 vertices: narrow levels only, expanded by the first thread
        outs = [run(checker, A, 256, team=t) for t in (1, 2, 5, 16)]
        assert all(int(o["bad"]) == 0 for o in outs)
        assert len({o["hash"] for o in outs}) == 1 and len({o["flops"] for o in outs}) == 1


def test_dense_row_and_column_do_not_make_a_dense_front(checker):
    """arrow matrix (a hub joined to every vertex) and two hubs on a chain: the hubs become separators
    of their own and the rest is dissected; without that the whole matrix would be one leaf"""
    n = 6000
    hub = sp.lil_matrix((n, n))
    hub[0, :] = 1
    hub[:, 0] = 1
    out = run(checker, sp.csc_matrix(hub) + sp.identity(n), 64)
    assert int(out["bad"]) == 0 and int(out["fronts"]) > 10 and int(out["maxfront"]) <= 4 * 64 + 1
    chain = sp.diags([np.ones(n - 1), np.ones(n - 1)], (-1, 1)).tolil()
    for h in (17, 4000):
        chain[h, :] = 1
        chain[:, h] = 1
    out = run(checker, sp.csc_matrix(chain) + sp.identity(n), 64)
    assert int(out["bad"]) == 0 and int(out["maxfront"]) <= 4 * 64 + 2
    # a dense block stays one leaf
    out = run(checker, np.ones((300, 300)), 64)
    assert int(out["bad"]) == 0 and int(out["fronts"]) == 1


@pytest.mark.parametrize("m,dim,leaf", [(40, 2, 32), (14, 3, 128)])
def test_expanded_tree_of_block_matrices(checker, m, dim, leaf):
    """mult = 2 (the real embedding of a complex matrix, umfpack_zi.hip): ordered on the small pattern, every
    vertex replaced by its two unknowns — all invariants hold against the expanded pattern, the fronts are
    exactly twice as large and there are as many of them"""
    S = poisson(m, dim)
    small = run(checker, S, leaf)
    big = run(checker, S, leaf, mult=2)
    assert int(big["bad"]) == 0 and int(big["n"]) == 2 * int(small["n"])
    assert int(big["fronts"]) == int(small["fronts"]) and int(big["maxfront"]) == 2 * int(small["maxfront"])
    # an unsymmetric pattern without a stored diagonal
    rng = np.random.default_rng(9)
    n = 800
    R = sp.coo_matrix((np.ones(4000), (rng.integers(0, n, 4000), rng.integers(0, n, 4000))), shape=(n, n))
    assert int(run(checker, R, 16, mult=2)["bad"]) == 0

