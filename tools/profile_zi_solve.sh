#!/bin/bash
# kernel statistics of the FEAST-like complex batched solve: tools/profile_zi_solve.py under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/zsolve_prof
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$repo/tools/profile_zi_solve.py" "${1:-80}" "${2:-16}" "${3:-3}" ${4:-} > "$out.log" 2>&1
grep "batched" "$out.log" | cut -c1-90
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-66s calls=%-6s total=%9.2f ms avg=%9.1f us %5.1f%%" % (r["Name"].replace("spl::(anonymous namespace)::", "").replace("void ", "")[:66], r["Calls"], float(r["TotalDurationNs"]) * 1e-6, float(r["AverageNs"]) * 1e-3, float(r["Percentage"])))
PY
