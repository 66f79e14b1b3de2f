#!/bin/bash
# usage: bash tools/pmc_sweep.sh <tag> "<bench args 1>" "<bench args 2>" ...
# per configuration: kernel time from bench.py, then TCC hit/miss/RDREQ for the SpMV kernel
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
i=0
for cfg in "$@"; do
  i=$((i+1))
  t=$(timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/$tag/c$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 $cfg > gpurun_out/$tag/c$i.log 2>&1
  python3 - "$cfg" "$t" gpurun_out/$tag/c$i <<'PY' | tee -a gpurun_out/$tag/summary.txt
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[3]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k,v in acc.items()}
print("%-28s ms=%s rdreq=%.4g hit=%.4g miss=%.4g req=%.4g fabricGB=%.2f" % (sys.argv[1], sys.argv[2], m.get("TCC_EA0_RDREQ_sum",0), m.get("TCC_HIT_sum",0), m.get("TCC_MISS_sum",0), m.get("TCC_REQ_sum",0), m.get("TCC_EA0_RDREQ_sum",0)*128/1e9))
PY
done
