"""matrices for tools/transversal_bench.cpp: transversal_gen.py perm2d|mesh3d m out.bin (the families of tools/fuzz_lu_scale.py)"""
import sys, numpy as np, scipy.sparse as sp
fam, m, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(5)
if fam == "perm2d":
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
    P = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)) + sp.identity(m * m)).tocoo()
    v = 10.0 ** rng.uniform(-3, 3, P.nnz) * rng.choice([-1.0, 1.0], P.nnz)
    perm = rng.permutation(m * m)
    S = sp.csc_matrix((v, (perm[P.row], P.col)), shape=(m * m, m * m))
else:  # mesh3d: random unsymmetric 7-point values, diagonal 1e-12
    T = sp.diags([np.ones(m - 1), np.ones(m - 1)], (-1, 1))
    I = sp.identity(m)
    P = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I) + sp.identity(m ** 3)).tocoo()
    v = rng.uniform(-1.0, 1.0, P.nnz)
    v[P.row == P.col] = 1e-12
    S = sp.csc_matrix((v, (P.row, P.col)), shape=(m ** 3, m ** 3))
S.sort_indices()
n = S.shape[0]
with open(out, "wb") as f:
    np.array([n, S.nnz], dtype=np.int64).tofile(f)
    S.indptr.astype(np.int32).tofile(f); S.indices.astype(np.int32).tofile(f); S.data.astype(np.float64).tofile(f)
print(n, S.nnz)
