#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 400 python tools/fuzz_complex.py 7 150 2>/dev/null | tail -1; env "$@" timeout -k 10 400 python tools/fuzz_lu.py 11 250 2>/dev/null | tail -1; }
run X=1
run SPL_LU_METHOD=mf SPL_ZI_NATIVE=1
run SPL_LU_METHOD=mf SPL_ND_GPU_MIN=300 SPL_MF_SMALL=64 SPL_MF_MIDMAX=256 SPL_MF_BIGSOLVE=64 SPL_ZI_NATIVE=1
run SPL_LU_METHOD=mf SPL_LU_BLOCK_PIVOT=0 SPL_ZI_NATIVE=1
