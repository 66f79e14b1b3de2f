#!/bin/bash
# complex symmetric z I - A (FEAST contour point) through umfpack_zi_*: native complex fronts (L D L^T), the symmetric
# real embedding of D A D (SPL_ZI_NATIVE=0 SPL_ZI_SYMMETRIC=1) and the general real embedding (=0 / =0)
run() {  # label, env..., then bench_solve args after --
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "== $label"
  env "${envs[@]}" timeout -k 10 400 python tools/bench_solve.py --cpu-max 0 "$@" 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['m'], d['gpu'], 'err %.1e %.1e' % (d['max_rel_err_vs_manufactured'], d['transposed_max_rel_err']), 'path', d['factorisation']['path'], 'flops %.3g' % d['factorisation']['flops'], 'GB', d['factorisation']['device_GB'])"
}
G3=${1:-60,100}
run "native complex fronts, 3-D" SPL_ZI_NATIVE=1 -- --grid $G3 --shift "3.0+0.5j"
run "symmetric embedding, 3-D" SPL_ZI_NATIVE=0 SPL_ZI_SYMMETRIC=1 -- --grid $G3 --shift "3.0+0.5j"
run "general embedding, 3-D" SPL_ZI_NATIVE=0 SPL_ZI_SYMMETRIC=0 -- --grid $G3 --shift "3.0+0.5j"
run "native complex fronts, 2-D 1000^2" SPL_ZI_NATIVE=1 -- --dim 2 --grid 1000 --shift "0.001+0.0005j"
run "general embedding, 2-D 1000^2" SPL_ZI_NATIVE=0 SPL_ZI_SYMMETRIC=0 -- --dim 2 --grid 1000 --shift "0.001+0.0005j"
