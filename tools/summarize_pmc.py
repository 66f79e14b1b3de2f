#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of tools/profile_spmv.sh: per-kernel average duration
(kernel-trace stats) and per-dispatch PMC counter values for the SpMV kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for row in csv.DictReader(open(f)):
        if float(row["Percentage"]) > 0.5:
            print("  %-60s calls=%s avg=%.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "spmv" not in name:
                continue
            acc[name[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print("== %s :: %s" % (os.path.basename(d), k))
        for c, vals in sorted(cs.items()):
            print("  %-32s n=%d mean=%.6g" % (c, len(vals), sum(vals) / len(vals)))
