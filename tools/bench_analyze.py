#!/usr/bin/env python3
"""Time of umfpack_di_symbolic alone (host work: orderings and the frontal tree) on mesh matrices;
sorted times of --reps calls.  python tools/bench_analyze.py --grid 100 --dim 3"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="64,100")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(0)
    for m in [int(t) for t in args.grid.split(",")]:
        H = pkg.DeviceMatrix.synthetic("poisson3d" if args.dim == 3 else "poisson2d", m)
        rp, ci, v = H.export_csr()
        H.free()
        n = m ** args.dim
        A = pkg.Matrix(n, n, rp, ci, v)
        ts = []
        for _ in range(args.reps):
            t = time.perf_counter()
            an = pkg.umfpack.analyze(A)
            ts.append(time.perf_counter() - t)
            del an
        print(json.dumps({"m": m, "dim": args.dim, "n": n, "analyze_s": [round(t, 4) for t in sorted(ts)]}), flush=True)


if __name__ == "__main__":
    main()
