cd $GRAFT_REPO_ROOT
for v in 1000000 400000 150000; do
  echo "SPL_ND_GPU_MIN=$v"
  SPL_ND_GPU_MIN=$v timeout -k 10 400 python tools/bench_solve.py --grid 100,100,160 --cpu-max 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' 3d', d['m'], d['gpu']['analyze_s'], d['factorisation']['flops'])
"
  SPL_ND_GPU_MIN=$v timeout -k 10 400 python tools/bench_solve.py --dim 2 --grid 1400,3000 --cpu-max 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' 2d', d['m'], d['gpu']['analyze_s'], d['factorisation']['flops'])
"
done
