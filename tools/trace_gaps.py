#!/usr/bin/env python3
"""GPU busy time inside the last factorisation of a rocprofv3 --kernel-trace run of tools/first_factor_probe.py:
union of the kernel intervals (busy), sum of the kernel durations (work), per-kernel totals.  usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the factorisations are delimited by assemble_kernel bursts: take the window from the last 'rel_kernel'/first assemble of the last run
starts = [i for i, e in enumerate(ev) if "assemble_kernel" in e[2]]
# split into runs: a gap of > 20 ms between assemble kernels starts a new factorisation
runs, cur = [], [starts[0]]
for a, b in zip(starts, starts[1:]):
    if ev[b][0] - ev[a][0] > 50_000_000: runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
first = runs[-1][0]
# end: last compact_kernel after it
last = max(i for i, e in enumerate(ev) if "compact_kernel" in e[2])
win = ev[first:last + 1]
t0, t1 = win[0][0], max(e[1] for e in win)
busy, cur_end = 0, t0
for s, e, _ in win:
    if e <= cur_end: continue
    busy += e - max(s, cur_end); cur_end = e
work = sum(e - s for s, e, _ in win)
print("window %.2f ms, GPU busy %.2f ms (%.0f %%), sum of kernel durations %.2f ms (concurrency %.2f), %d kernels" % (
    (t1 - t0) * 1e-6, busy * 1e-6, 100.0 * busy / (t1 - t0), work * 1e-6, work / max(busy, 1), len(win)))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    import re
    m = re.search(r"(\w+_kernel\w*|__amd_rocclr_\w+)", k)
    n = m.group(1) if m else k[:40]
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-34s calls %6d  total %8.2f ms  avg %7.1f us" % (n, c, d * 1e-6, d / c * 1e-3))
