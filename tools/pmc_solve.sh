#!/bin/bash
# HBM-side traffic of the triangular solves from the PMC counters (separate passes, as the guide prescribes):
#   bash tools/pmc_solve.sh [m [z]]   -> gpurun_out/pmc_solve/summary.txt   (z: the complex shifted matrix, umfpack_zi_*)
# Sums TCC_EA0_RDREQ (x 128 B, less 96 B per 32-B request) and WRITE_SIZE over the kernels of ONE linearSolve_ call and
# sets them against walks x bytes per walk of spl_umfpack_solve_report.
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_solve
m=${1:-128}; zflag=${2:-}
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$out/rd" -- python3 "$repo/tools/pmc_solve_target.py" "$m" $zflag > "$out/rd.log" 2>&1
echo "[pmc_solve] read pass rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/wr" -- python3 "$repo/tools/pmc_solve_target.py" "$m" $zflag > "$out/wr.log" 2>&1
echo "[pmc_solve] write pass rc=$?"
python3 - "$out" "$m" <<'PY' | tee "$out/summary.txt"
import csv, glob, re, sys, collections, ast
out, m = sys.argv[1], sys.argv[2]
solve = re.compile(r"big_gemv|big_super|big_chain|chain_build|big_gather|big_scatter|big_boundary|solve_forward|solve_backward|solve_gather|solve_init")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not solve.search(k): continue
        name = re.search(r"(big_\w+|chain_\w+|solve_\w+)(<[^>]*>)?", k).group(0)[:44]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_EA0_RDREQ_sum": calls[name] += 1
rep = None
for l in open(out + "/rd.log"):
    if l.startswith("SOLVE_REPORT"): rep = ast.literal_eval(l[len("SOLVE_REPORT "):l.index("}") + 1])
tot_r = tot_w = 0.0
print("3-D Poisson %s^3, one linearSolve_ call: %s" % (m, rep))
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0)):
    rd = c.get("TCC_EA0_RDREQ_sum", 0) * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
    wr = c.get("WRITE_SIZE", 0) * 1024
    if name.startswith("chain_build"):  # once per factorisation (the first solve builds the chain matrices): not a walk's traffic
        print("  %-42s launches %5d  read %9.3f GB  written %8.3f GB   (once per factorisation, not in the sums)" % (name, calls[name], rd * 1e-9, wr * 1e-9))
        continue
    tot_r += rd; tot_w += wr
    print("  %-42s launches %5d  read %9.3f GB  written %8.3f GB" % (name, calls[name], rd * 1e-9, wr * 1e-9))
alg = rep["walks"] * rep["walk_bytes"] if rep else 0
print("all solve kernels: read %.3f GB + written %.3f GB = %.3f GB; algorithmic %d walks x %.3f GB = %.3f GB; traffic / algorithmic = %.3f"
      % (tot_r * 1e-9, tot_w * 1e-9, (tot_r + tot_w) * 1e-9, rep["walks"] if rep else 0, (rep["walk_bytes"] if rep else 0) * 1e-9, alg * 1e-9, (tot_r + tot_w) / alg if alg else 0))
PY
