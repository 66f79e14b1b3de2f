#!/bin/bash
# detailed counter passes for the default SpMV kernel: bash tools/pmc_detail.sh <tag> [bench args]
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $BARGS > $out/$name.log 2>&1; echo "[pmc] $name rc=$?"; }
BARGS="$*"
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp3 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TD_TCP_STALL_CYCLES_sum
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
python3 - $out <<'PY' | tee $out/summary.txt
import csv,glob,collections,sys
for d in sorted(glob.glob(sys.argv[1]+"/*/")):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "spmv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print("%-44s n=%d mean=%.5g" % (k, len(v), sum(v)/len(v)))
PY
