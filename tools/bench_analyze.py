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


def _throttled():
    try:
        kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except Exception:
        return None


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
        ts, cpu = [], []
        thr0 = _throttled()
        for _ in range(args.reps):
            t, c = time.perf_counter(), time.process_time()
            an = pkg.umfpack.analyze(A)
            ts.append(time.perf_counter() - t)
            cpu.append(time.process_time() - c)
            del an
        thr1 = _throttled()
        out = {"m": m, "dim": args.dim, "n": n, "analyze_s": [round(t, 4) for t in sorted(ts)],
               "cpu_s": [round(c, 3) for c in sorted(cpu)]}
        if thr0 and thr1:  # the box's CPU share is a CFS quota: periods in which the process group was stopped
            out["throttled"] = {"periods": thr1[0] - thr0[0], "ms": round((thr1[1] - thr0[1]) / 1e3, 1)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
