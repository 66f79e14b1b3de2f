#!/bin/bash
# Profile bench.py's SpMV on the GPU box: one kernel-trace/stats run, then separate
# PMC passes (TCC has 4 counter slots per pass: FETCH_SIZE costs 3, WRITE_SIZE 2).
# Usage (from the repo root, under gpurun):  bash tools/profile_spmv.sh <tag> [bench args...]
# Outputs under gpurun_out/<tag>/ ; copy the summaries you keep into profiles/.
set -u
export TMPDIR=/tmp
tag=${1:-prof}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
run() {  # name, rocprofv3 args...
  local name=$1; shift
  echo "[profile] $name" | tee -a "$out/progress.log"
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --no-cpu-baseline --no-secondary "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1
  echo "[profile] $name rc=$?" | tee -a "$out/progress.log"
}
BENCH_ARGS=(--steps 20 --warmup 3 "$@")
run trace --kernel-trace --stats
BENCH_ARGS=(--steps 2 --warmup 1 "$@")
run pmc_fetch --pmc FETCH_SIZE TCC_EA0_RDREQ_DRAM_sum
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run pmc_req --pmc TCC_REQ_sum TCC_READ_sum TCC_READ_SECTORS_sum TCP_TCC_READ_REQ_sum
python3 tools/summarize_pmc.py "$out" | tee "$out/summary.txt"
