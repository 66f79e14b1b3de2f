#!/usr/bin/env python3
"""Sequence of kernels of the last walk in a rocprofv3 --kernel-trace csv: start offset, duration, grid, short name.
usage: trace_seq.py <kernel_trace.csv> [max rows]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")) for r in rows))
inits = [i for i, e in enumerate(ev) if "solve_init_kernel" in e[2]]
win = ev[inits[-1]:]
t0 = win[0][0]
prev_end = t0
for s, e, k, g, w in win[: int(sys.argv[2]) if len(sys.argv) > 2 else 400]:
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
    try: wgs = int(g) // max(1, int(w))
    except Exception: wgs = -1
    print("%9.1f us  dur %7.1f  gap %5.1f  wgs %6d  %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, (s - prev_end) * 1e-3, wgs, m.group(0)[:50] if m else k[:50]))
    prev_end = max(prev_end, e)
