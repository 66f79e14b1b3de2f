#!/bin/bash
cd $GRAFT_REPO_ROOT
for k in 1 2 4; do
  echo "threads $k"
  SPL_FEAST_THREADS=$k timeout -k 10 400 python tools/bench_feast.py --dim 3 --grid 80 --m0 16 --lo 0.003 --hi 0.0175 2>/dev/null | tail -1 | cut -c150-500
done
SPL_FEAST_THREADS=2 timeout -k 10 400 python tools/bench_feast.py --grid 1000 --m0 16 --lo 3.94e-05 --hi 0.000131 2>/dev/null | tail -1 | cut -c150-500
