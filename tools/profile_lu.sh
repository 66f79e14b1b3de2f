#!/bin/bash
# kernel statistics of config C5 at full size: rocprofv3 --kernel-trace --stats around tools/bench_solve.py
# usage: tools/profile_lu.sh [grid] [extra bench_solve arguments, e.g. --shift 3.0+0.5j]   (writes gpurun_out/lu_prof/)
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/lu_prof
grid=${1:-200}
shift 2>/dev/null
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$repo/tools/bench_solve.py" --grid "$grid" --cpu-max 0 "$@" > "$out/bench.log" 2>&1
echo "[profile_lu] rc=$?"
tail -2 "$out/bench.log"
f=$(ls "$out"/kt/*/*kernel_stats.csv | head -1)
python3 - "$f" > "$out/kernel_stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-78s calls=%-6s total=%10.2f ms avg=%10.1f us %6.1f%%" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) * 1e-6,
                                                                     float(r["AverageNs"]) * 1e-3, float(r["Percentage"])))
PY
cat "$out/kernel_stats.txt"
