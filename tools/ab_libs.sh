#!/bin/bash
# is the A/B order-dependent?  legs: new old old new old new new old (phases of the analysis, SPL_MF_TIMING=1 for both)
cd "$(dirname "$0")/.." || exit 1
L=sparse-linear_amd/lib
for v in new old old new old new new old; do
  cp $L/ab_$v.so $L/libsparse_linear_hip.so
  echo -n "$v: "
  SPL_MF_TIMING=1 timeout -k 10 120 python tools/bench_analyze.py --grid 100 --reps 6 2>&1 | grep "build_tree\] dissection\|analyze_s" | tr '\n' ' ' | sed 's/\[build_tree\] //g; s/  */ /g'
  echo
done
