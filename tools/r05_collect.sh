#!/bin/bash
# Round-5 evidence in one GPU call: kernel trace + PMC passes of the headline SpMV, of config C4's product, of the solves
# (100^3, 200^3) and of the factorisation (100^3); the full bench line; three FEAST runs.  Everything lands under
# gpurun_out/r05/ (the raw counter files of the LU passes are deleted once summarised: gpurun merges at most 64 MiB back);
# the summaries worth keeping are copied to profiles/ by hand afterwards.
#   bash tools/r05_collect.sh [steps...]     steps: spmv spgemm solve100 factor100 solve200 spmvother solvez100 factorz100 factor200 bench feast
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd "$repo" || exit 1
mkdir -p gpurun_out/r05
steps=${*:-spmv spgemm solve100 factor100 solve200 bench feast}
for st in $steps; do
  echo "[r05_collect] $st $(date +%T)"
  case $st in
    spmv)      bash tools/profile_spmv.sh r05/spmv_random > gpurun_out/r05/profile_spmv.log 2>&1 || { echo "spmv profile failed"; exit 1; } ;;
    spgemm)    bash tools/profile_spgemm.sh r05/spgemm_c4 > gpurun_out/r05/profile_spgemm.log 2>&1 || { echo "spgemm profile failed"; exit 1; } ;;
    solve100)  bash tools/pmc_solve.sh 100 > gpurun_out/r05/pmc_solve_100.log 2>&1 || { echo "pmc_solve 100 failed"; exit 1; }
               cp gpurun_out/pmc_solve/summary.txt gpurun_out/r05/pmc_solve_100.txt; rm -rf gpurun_out/pmc_solve ;;
    factor100) bash tools/pmc_factor.sh 100 > gpurun_out/r05/pmc_factor_100.log 2>&1 || { echo "pmc_factor 100 failed"; exit 1; }
               cp gpurun_out/pmc_factor/summary.txt gpurun_out/r05/pmc_factor_100.txt; rm -rf gpurun_out/pmc_factor ;;
    solve200)  bash tools/pmc_solve.sh 200 > gpurun_out/r05/pmc_solve_200.log 2>&1 || { echo "pmc_solve 200 failed"; exit 1; }
               cp gpurun_out/pmc_solve/summary.txt gpurun_out/r05/pmc_solve_200.txt; rm -rf gpurun_out/pmc_solve ;;
    spmvother) bash tools/pmc_spmv_other.sh poisson3d:200 reference > gpurun_out/r05/pmc_spmv_poisson.log 2>&1 && bash tools/pmc_spmv_other.sh poisson3d:200 free >> gpurun_out/r05/pmc_spmv_poisson.log 2>&1 &&
               bash tools/pmc_spmv_other.sh rmat:20 reference > gpurun_out/r05/pmc_spmv_rmat.log 2>&1 && bash tools/pmc_spmv_other.sh rmat:20 free >> gpurun_out/r05/pmc_spmv_rmat.log 2>&1 || { echo "pmc_spmv_other failed"; exit 1; } ;;
    solvez100) bash tools/pmc_solve.sh 100 z > gpurun_out/r05/pmc_solve_z100.log 2>&1 || { echo "pmc_solve z 100 failed"; exit 1; }
               cp gpurun_out/pmc_solve/summary.txt gpurun_out/r05/pmc_solve_z100.txt; rm -rf gpurun_out/pmc_solve ;;
    factorz100) bash tools/pmc_factor.sh 100 z > gpurun_out/r05/pmc_factor_z100.log 2>&1 || { echo "pmc_factor z 100 failed"; exit 1; }
               cp gpurun_out/pmc_factor/summary.txt gpurun_out/r05/pmc_factor_z100.txt; rm -rf gpurun_out/pmc_factor ;;
    factor200) FFP_REPS=1 bash tools/pmc_factor.sh 200 > gpurun_out/r05/pmc_factor_200.log 2>&1 || { echo "pmc_factor 200 failed"; exit 1; }
               cp gpurun_out/pmc_factor/summary.txt gpurun_out/r05/pmc_factor_200.txt; rm -rf gpurun_out/pmc_factor ;;
    bench)     timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err || { echo "bench failed"; exit 1; }
               wc -c gpurun_out/r05/bench.json ;;
    feast)     for i in 1 2 3; do timeout -k 10 200 python3 tools/bench_secondary.py --item feast:80 2>/dev/null | grep '^{' | tail -1 >> gpurun_out/r05/feast3.jsonl || { echo "feast failed"; exit 1; }; done ;;
  esac
done
echo "[r05_collect] done $(date +%T)"
