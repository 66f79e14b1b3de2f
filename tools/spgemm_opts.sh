#!/bin/bash
# A/B of the ordered SpGEMM kernel's ring on config C4: bash tools/spgemm_opts.sh <tag> <ring> [<ring> ...]   (ring: default | narrow)
out=gpurun_out/${1:-spo}; shift; mkdir -p "$out"
for o in "$@"; do
  SPL_SPGEMM_RING=$o SPL_SPGEMM_TIMING=1 timeout -k 10 60 python3 tools/bench_spgemm.py --reps 4 --cpu-cols 1024 > "$out/ring_$o.txt" 2>&1
  rc=$?
  echo "ring=$o rc=$rc kernel ms: $(grep 'ordered kernel' "$out/ring_$o.txt" | awk '{print $4}' | tr '\n' ' ') | $(tail -1 "$out/ring_$o.txt" | grep -o '"seconds": [0-9.]*') $(tail -1 "$out/ring_$o.txt" | grep -o '"structure_and_values_bit_identical": [a-z]*')"
  if [ $rc -ne 0 ]; then echo "stopping: a run failed or hung"; break; fi
done
