#!/bin/bash
# timing-only ablations of the panel kernel on C2 (wrong results by construction):
# bit 0 gathers read x[lane], bit 1 no value loads, bit 2 no LDS fold
export SPL_ALLOW_ABLATION=1
for a in 0 1 2 3 4 5 6 7; do
  echo "ablate=$a"
  SPL_PANEL_ABLATE=$a python tools/bench_spmv_variants.py panel:19532:17:12:2 2>/dev/null
done
