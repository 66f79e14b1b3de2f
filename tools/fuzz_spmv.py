#!/usr/bin/env python3
"""Fuzz of the SpMV kernels on irregular structures the synthetic generators never produce: empty rows and
columns, a few very long rows, skewed column distributions, rectangular shapes, duplicated-looking neighbours,
sizes around the panel / block boundaries.  Every kernel family is forced in turn on the same handle (CSR-stream,
sub-wavefront, column-blocked image, sliced ELL where its shape test admits the matrix, column-sorted panels in
every storage form with random shapes) and compared with the oracle: bit for bit in the reference-order kernels,
to 1e-10 (and exactly on integer data) in the order-free panels; y <- A x + y as well as y = A x.
python tools/fuzz_spmv.py [seed] [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def structure(rng, kind, nr, nc):
    """(rows, cols) of a random pattern"""
    if kind == 0:    # uniform
        k = int(rng.integers(0, 8 * nr + 1))
        return rng.integers(0, nr, k), rng.integers(0, nc, k)
    if kind == 1:    # a few very long rows + sparse rest, many empty rows
        k = int(rng.integers(1, 3 * nr + 2))
        rows = rng.integers(0, max(1, nr // 3), k)
        cols = rng.integers(0, nc, k)
        for _ in range(int(rng.integers(1, 4))):
            r = int(rng.integers(0, nr))
            m = int(min(nc, rng.integers(1, 5000)))
            rows = np.concatenate([rows, np.full(m, r)])
            cols = np.concatenate([cols, rng.choice(nc, m, replace=False)])
        return rows, cols
    if kind == 2:    # columns crowded into a narrow range (one index block gets almost everything)
        k = int(rng.integers(1, 6 * nr + 2))
        lo = int(rng.integers(0, nc))
        wdt = int(rng.integers(1, max(2, nc // 50)))
        return rng.integers(0, nr, k), np.minimum(nc - 1, lo + rng.integers(0, wdt, k))
    if kind == 3:    # banded with holes
        k = int(rng.integers(1, 5 * nr + 2))
        rows = rng.integers(0, nr, k)
        cols = np.clip((rows.astype(np.int64) * nc) // max(nr, 1) + rng.integers(-40, 41, k), 0, nc - 1)
        return rows, cols
    # power-law rows
    k = int(rng.integers(1, 6 * nr + 2))
    rows = np.minimum(nr - 1, (nr * rng.power(0.3, k)).astype(np.int64))
    return rows, rng.integers(0, nc, k)


def main():
    import torch
    from __graft_entry__ import load_package
    from oracle import oracle as O
    pkg = load_package()
    torch.cuda.set_device(0)
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    rng = np.random.default_rng(seed)
    bad = launches = 0
    s = torch.cuda.current_stream()
    for case in range(ncase):
        nr = int(rng.choice([1, 2, 63, 64, 65, 257, 1000, 4097, 20479, 20480, 50_000, 131_073]))
        nc = int(rng.choice([1, 3, 64, 1000, 4096, 131_072, 131_073, 400_000])) if case % 3 else nr
        ints = case % 4 == 0
        rows, cols = structure(rng, case % 5, nr, nc)
        vals = rng.integers(-8, 9, len(rows)).astype(float) if ints else rng.normal(size=len(rows))
        A = O.compress(nr, nc, rows, cols, vals)          # CSC tuple, duplicates summed
        M = pkg.Matrix(nc, nr, A[2], A[3], A[4])
        xh = rng.integers(-5, 6, nc).astype(float) if ints else rng.normal(size=nc)
        y0 = rng.integers(-5, 6, nr).astype(float) if ints else rng.normal(size=nr)
        ref, refa = O.mulV(A, xh), O.axpy(A, xh, y0)
        scale = O.mulV((A[0], A[1], A[2], A[3], np.abs(A[4])), np.abs(xh)) + np.abs(y0)
        x = torch.from_numpy(xh).cuda()
        H = pkg.DeviceMatrix.from_csc(M)
        # CSR-stream: rows longer than one 512-entry LDS chunk are summed by a wavefront tree (documented exception
        # to the reference order): exact comparison only when no row is that long
        lens = np.diff(O.transpose(A)[2])
        short = bool(lens.max(initial=0) <= 512)
        plans = [("stream", lambda: H.set_variant(1)), ("stream4", lambda: H.set_variant(4)),
                 ("default", lambda: (H.set_variant(0), H.optimize()))]

        def blocked():
            H.build_blocked(int(rng.choice([0, 64, 500, 1221])), int(rng.choice([0, 8, 12, 16])), 0)
            H.set_variant(8)
        plans.append(("blocked", blocked))

        def sell():
            H.optimize()
            H.set_variant(15)
        plans.append(("sell", sell))
        for form in (1, 2, 4, 5):
            def panel(form=form):
                P = int(rng.choice([0, 64, 777, 4096, 20479]))
                w = int(rng.choice([0, 4, 9, 13, 17]))
                H.build_panel(P, w if P else 0, 0, form)
                H.set_variant(16)
            plans.append(("panel%d" % form, panel))
        for name, plan in plans:
            try:
                plan()
            except Exception as e:  # a shape the variant does not admit (reported as an error, never a crash)
                if "argument" in str(e) or "invalid" in str(e):
                    continue
                raise
            exact = ints or (name in ("blocked", "sell")) or (name in ("stream", "stream4", "default") and short)
            for acc in (False, True):
                y = torch.from_numpy(y0.copy()).cuda() if acc else torch.full((nr,), 7.0, dtype=torch.float64, device="cuda")
                H.spmv_dev(x.data_ptr(), y.data_ptr(), accumulate=acc, stream=s.cuda_stream)
                torch.cuda.synchronize()
                launches += 1
                got, want = y.cpu().numpy(), (refa if acc else ref)
                ok = np.array_equal(got, want) if exact else bool(np.all(np.abs(got - want) <= 1e-13 * np.maximum(scale, 1e-300) * max(1, len(rows))))
                if not ok:
                    bad += 1
                    print("case %d %s acc=%d: nr=%d nc=%d nnz=%d kind=%d ints=%d: max diff %.3e" %
                          (case, name, acc, nr, nc, len(A[3]), case % 5, ints, float(np.max(np.abs(got - want)))), flush=True)
        H.free()
        if case % 25 == 24:
            print("... %d cases, %d launches, %d failures" % (case + 1, launches, bad), flush=True)
    print("fuzz_spmv: %d cases, %d launches, %d failures" % (ncase, launches, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
