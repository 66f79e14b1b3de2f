#!/bin/bash
# HBM-side traffic of the kernels of the numeric factorisation from the PMC counters (separate passes, as the guide
# prescribes), per kernel, next to their time from a kernel trace of the same command:
#   bash tools/pmc_factor.sh [m [z]]   -> gpurun_out/pmc_factor/summary.txt    (3-D Poisson m^3 — z: the complex shifted matrix —,
#   tools/first_factor_probe.py: FFP_REPS factorisations, default 3)
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_factor
m=${1:-100}; zarg=""; [ "${2:-}" = "z" ] && { zarg="cold z"; export PMC_FACTOR_Z=1; }
rm -rf "$out"; mkdir -p "$out"
export FFP_QUIET=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$repo/tools/first_factor_probe.py" "$m" $zarg > "$out/trace.log" 2>&1
echo "[pmc_factor] trace rc=$?"
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$out/rd" -- python3 "$repo/tools/first_factor_probe.py" "$m" $zarg > "$out/rd.log" 2>&1
echo "[pmc_factor] read pass rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/wr" -- python3 "$repo/tools/first_factor_probe.py" "$m" $zarg > "$out/wr.log" 2>&1
echo "[pmc_factor] write pass rc=$?"
python3 - "$out" "$m" <<'PY' | tee "$out/summary.txt"
import csv, glob, re, sys, collections
out, m = sys.argv[1], sys.argv[2]
def short(k):
    mm = re.search(r"(\w+_kernel\w*|__amd_rocclr_\w+)(<[^>]*>)?", k)
    return mm.group(0)[:40] if mm else k[:40]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/rd/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/wr/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur[short(r["Kernel_Name"])]
        d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
import os
print("3-D Poisson %s^3%s, tools/first_factor_probe.py (analysis + %s factorisation(s)): per kernel, summed over the run" % (m, " (complex shift z I - A)" if os.environ.get("PMC_FACTOR_Z") else "", os.environ.get("FFP_REPS", "3")))
rows = []
for name, c in acc.items():
    rd = c.get("TCC_EA0_RDREQ_sum", 0) * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
    wr = c.get("WRITE_SIZE", 0) * 1024
    rows.append((dur[name][1], name, dur[name][0], rd, wr))
for t, name, calls, rd, wr in sorted(rows, reverse=True)[:16]:
    print("  %-40s launches %6d  %8.2f ms  read %8.2f GB  written %8.2f GB  -> %5.2f TB/s while it runs" % (
        name, calls, t * 1e3, rd * 1e-9, wr * 1e-9, (rd + wr) / t * 1e-12 if t > 0 else 0))
PY
