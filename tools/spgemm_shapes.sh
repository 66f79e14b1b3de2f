#!/bin/bash
# seconds of A*A over a ladder of shapes (tools/bench_spgemm.py --reps 3, 256 sampled rows against the oracle): bash tools/spgemm_shapes.sh <out file>
out=${1:-gpurun_out/spgemm_shapes.txt}
run() {
  local label=$1; shift
  r=$(timeout -k 10 120 python3 tools/bench_spgemm.py --reps 3 --cpu-cols 256 "$@" 2>/dev/null | tail -1)
  echo "$label  $(echo "$r" | grep -o '"seconds": [0-9.]*')  $(echo "$r" | grep -o '"products": [0-9]*')  $(echo "$r" | grep -o '"structure_and_values_bit_identical": [a-z]*')" | tee -a "$out"
}
: > "$out"
run "scale 20  ef  8  ER" --scale 20 --edge-factor 8
run "scale 20  ef 24  ER" --scale 20 --edge-factor 24
run "scale 20  ef 32  ER = C4" --scale 20 --edge-factor 32
run "scale 20  ef 36  ER" --scale 20 --edge-factor 36
run "scale 20  ef 40  ER" --scale 20 --edge-factor 40
run "scale 20  ef 48  ER" --scale 20 --edge-factor 48
run "scale 21  ef 16  ER" --scale 21 --edge-factor 16
run "scale 19  ef 32  (0.45,0.22,0.22)" --scale 19 --edge-factor 32 --abc 0.45,0.22,0.22
run "scale 18  ef 16  (0.57,0.19,0.19)" --scale 18 --edge-factor 16 --abc 0.57,0.19,0.19
