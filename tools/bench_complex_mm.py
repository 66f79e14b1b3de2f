#!/usr/bin/env python3
"""mm on Complex Double (csrc/spgemm_z.hip) next to the real mm on the same pattern: one-shot calls through the
C ABI (upload, product, download), seconds and products; a sample of columns is checked against the oracle."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from __graft_entry__ import load_package
    from oracle import oracle as O
    pkg = load_package()
    n, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 200_000), (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
    rp, ci, v = O.gen_random_csr(n, K)
    A = O.csr_to_csc_tuple(n, n, rp, ci, v)
    rng = np.random.default_rng(2)
    Mr = pkg.Matrix(n, n, A[2], A[3], A[4])
    Mz = pkg.Matrix(n, n, A[2], A[3], (A[4] + 1j * rng.uniform(0.5, 1.5, len(A[4]))).astype(np.complex128))
    out = {"n": n, "nnz": int(A[2][-1])}
    for name, M in (("real", Mr), ("complex", Mz)):
        pkg.mm(M, M)
        t = time.perf_counter()
        C = pkg.mm(M, M)
        out[name + "_s"] = round(time.perf_counter() - t, 4)
        out[name + "_nnzC"] = int(C.pointers[-1])
    cols = rng.choice(n, 64, replace=False)
    sel = (n, len(cols), np.concatenate([[0], np.cumsum(np.diff(Mz.pointers)[cols])]),
           np.concatenate([Mz.indices[Mz.pointers[c]:Mz.pointers[c + 1]] for c in cols]),
           np.concatenate([Mz.values[Mz.pointers[c]:Mz.pointers[c + 1]] for c in cols]))
    ref = O.mm_z((n, n, Mz.pointers, Mz.indices, Mz.values), sel)
    ok = True
    for t_, c in enumerate(cols):
        a, b = C.pointers[c], C.pointers[c + 1]
        ok &= np.array_equal(C.indices[a:b], ref[3][ref[2][t_]:ref[2][t_ + 1]]) and \
            np.array_equal(C.values[a:b], ref[4][ref[2][t_]:ref[2][t_ + 1]])
    out["sampled_columns_bit_identical"] = bool(ok)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
