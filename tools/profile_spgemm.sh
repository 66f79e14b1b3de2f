#!/bin/bash
# Profile tools/bench_spgemm.py (config C4 by default) on the GPU box: one kernel-trace/stats run, then
# separate PMC passes.  Usage: bash tools/profile_spgemm.sh <tag> [bench_spgemm args...]
set -u
export TMPDIR=/tmp
tag=${1:-spgemm}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
run() {
  local name=$1; shift
  echo "[profile] $name" | tee -a "$out/progress.log"
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 tools/bench_spgemm.py --reps 2 --cpu-cols 0 "${ARGS[@]}" > "$out/$name.log" 2>&1
  echo "[profile] $name rc=$?" | tee -a "$out/progress.log"
}
ARGS=("$@")
run trace --kernel-trace --stats
run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
run pmc_write --pmc WRITE_SIZE TCC_EA0_WRREQ_sum
run pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run pmc_sq2 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR
python3 - "$out" <<'PY' | tee "$out/summary.txt"
import csv, glob, collections, sys
root = sys.argv[1]
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats:", f)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    for r in rows[:12]:
        print("  %-70s calls=%s total=%.1f us avg=%.1f us (%s%%)" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
              float(r["AverageNs"]) / 1e3, r.get("Percentage", "")))
for d in sorted(glob.glob(root + "/pmc_*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "spgemm" not in k and "compact" not in k: continue
        print("== %s :: %s" % (d.rstrip("/").split("/")[-1], k))
        for c, v in sorted(cs.items()): print("  %-28s n=%d mean=%.5g sum=%.5g" % (c, len(v), sum(v) / len(v), sum(v)))
PY
