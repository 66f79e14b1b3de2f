#!/bin/bash
# quick look at config C4's product on the GPU box: seconds (5 reps), per-phase stamps of the ordered kernel, host laps
#   bash tools/spgemm_quick.sh <tag>   -> gpurun_out/<tag>/
out=gpurun_out/${1:-spq}; mkdir -p "$out"
timeout -k 10 200 python3 tools/bench_spgemm.py --reps 5 --cpu-cols 2048 > "$out/spgemm.txt" 2>&1; echo "bench rc=$?"; tail -1 "$out/spgemm.txt" | cut -c1-120,400-700
SPL_SPGEMM_STAMPS=1 timeout -k 10 120 python3 tools/bench_spgemm.py --reps 2 --cpu-cols 0 > "$out/stamps.txt" 2>&1; grep "ordered\]" "$out/stamps.txt" | tail -1
SPL_SPGEMM_TIMING=1 timeout -k 10 120 python3 tools/bench_spgemm.py --reps 2 --cpu-cols 0 > "$out/timing.txt" 2>&1; grep "ordered kernel" "$out/timing.txt" | tail -1
