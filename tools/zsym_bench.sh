#!/bin/bash
# complex symmetric z I - A (FEAST contour point): symmetric embedding (L D L^T) against the general embedding
for sym in 1 0; do
  echo "SPL_ZI_SYMMETRIC=$sym"
  SPL_ZI_SYMMETRIC=$sym timeout -k 10 400 python tools/bench_solve.py --grid 60,100 --cpu-max 0 --shift "3.0+0.5j" 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['m'], d['gpu'], d['max_rel_err_vs_manufactured'], d['transposed_max_rel_err'], d['factorisation']['path'], d['factorisation']['flops'], d['factorisation']['device_GB'])"
done
SPL_ZI_SYMMETRIC=1 timeout -k 10 200 python tools/bench_solve.py --dim 2 --grid 1000 --cpu-max 0 --shift "0.001+0.0005j" 2>&1 | grep "^{" | cut -c1-600
SPL_ZI_SYMMETRIC=0 timeout -k 10 200 python tools/bench_solve.py --dim 2 --grid 1000 --cpu-max 0 --shift "0.001+0.0005j" 2>&1 | grep "^{" | cut -c1-600
