#!/usr/bin/env python3
"""Where a tree solve's time goes, from a rocprofv3 --kernel-trace run of tools/pmc_solve_target.py: the window of the
LAST walk-and-residual sequence (from the last solve_init_kernel of the first solve call... to the end), GPU busy time,
sum of the kernel durations, gaps between consecutive kernels, per-kernel totals.  usage: trace_solve.py <kernel_trace.csv>"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
inits = [i for i, e in enumerate(ev) if "solve_init_kernel" in e[2]]
first = inits[0]
win = ev[first:]
t0, t1 = win[0][0], max(e[1] for e in win)
busy, cur_end, gaps = 0, t0, []
for s, e, _ in win:
    if s > cur_end: gaps.append(s - cur_end)
    if e <= cur_end: continue
    busy += e - max(s, cur_end); cur_end = e
work = sum(e - s for s, e, _ in win)
print("solve window %.3f ms (%d walks), GPU busy %.3f ms (%.0f %%), sum of kernel durations %.3f ms, %d kernels, gaps: %d, mean %.2f us, total %.3f ms" % (
    (t1 - t0) * 1e-6, len(inits), busy * 1e-6, 100.0 * busy / (t1 - t0), work * 1e-6, len(win), len(gaps), sum(gaps) / max(1, len(gaps)) * 1e-3, sum(gaps) * 1e-6))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in win:
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
    agg[m.group(0)[:60] if m else k[:60]][0] += 1
    agg[m.group(0)[:60] if m else k[:60]][1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print("  %-60s calls %6d  total %8.3f ms  avg %7.1f us" % (n, c, d * 1e-6, d / c * 1e-3))
