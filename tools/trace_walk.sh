#!/bin/bash
# kernel trace of one solve (tools/pmc_solve_target.py m [z]) -> gpurun_out/<tag>_summary.txt (tools/trace_solve.py) and
# gpurun_out/<tag>_seq.txt (tools/trace_seq.py: the kernels of the last walk in order).  usage: trace_walk.sh <tag> <m> [z]
export TMPDIR=/tmp
R=$PWD
tag=$1; shift
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/pmc_solve_target.py "$@" > gpurun_out/$tag.log 2>&1
f=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python3 tools/trace_solve.py $f > gpurun_out/${tag}_summary.txt
python3 tools/trace_seq.py $f 2000 > gpurun_out/${tag}_seq.txt
rm -rf gpurun_out/$tag
