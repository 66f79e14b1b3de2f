#!/bin/bash
# HBM-side traffic per launch of the SpMV kernel on the matrices of the other configurations (bench.py's spmv_poisson3d_200, spmv_rmat_20):
#   bash tools/pmc_spmv_other.sh poisson3d:200 reference   -> gpurun_out/pmc_spmv_<which>_<order>/summary.txt
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:-/root/repo}
which=${1:-poisson3d:200}; order=${2:-reference}
out=$repo/gpurun_out/pmc_spmv_${which//:/_}_$order
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$out/rd" -- python3 "$repo/tools/pmc_spmv_target.py" "$which" "$order" > "$out/rd.log" 2>&1
echo "[pmc_spmv_other] read pass rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/wr" -- python3 "$repo/tools/pmc_spmv_target.py" "$which" "$order" > "$out/wr.log" 2>&1
echo "[pmc_spmv_other] write pass rc=$?"
python3 - "$out" "$which" "$order" <<'PY' | tee "$out/summary.txt"
import csv, glob, sys, collections, re
out, which, order = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmv_" not in k: continue
        m = re.search(r"spmv_\w+(<[^>]*>)?", k)
        acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
tgt = [l for l in open(out + "/rd.log") if l.startswith("SPMV_TARGET")]
print("SpMV on %s, sum order %s: %s" % (which, order, tgt[0].strip() if tgt else "?"))
for k, c in acc.items():
    if len(c.get("TCC_EA0_RDREQ_sum", [])) < 6: continue  # the six timed launches (optimize() tries shapes once or twice)
    rd = sum(c["TCC_EA0_RDREQ_sum"]) / len(c["TCC_EA0_RDREQ_sum"]) * 128 - sum(c["TCC_EA0_RDREQ_32B_sum"]) / len(c["TCC_EA0_RDREQ_32B_sum"]) * 96
    wr = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
    print("  %-40s launches %d  read %.4f GB  written %.4f GB  traffic per launch %d bytes" % (k, len(c["TCC_EA0_RDREQ_sum"]), rd * 1e-9, wr * 1e-9, int(rd + wr)))
PY
