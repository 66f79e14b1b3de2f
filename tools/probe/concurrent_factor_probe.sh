#!/bin/bash
# one process, then two at once: refactor time of z I - A at 80^3 (native complex fronts)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python tools/bench_solve.py --grid 80,80,80 --cpu-max 0 --shift 3+0.5j 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('$1', d['gpu']['factor_s'], d['gpu']['refactor_s'], d['gpu']['solve_s'])
"; }
run single
run A & run B & wait
