#!/bin/bash
# usage: bash tools/gpu_batch.sh <outdir> <timeout_s> "<cmd 1>" "<cmd 2>" ...
# Runs the commands one after the other, each under its own `timeout -k 10`, stdout+stderr to
# <outdir>/stepN.log.  A step that fails goes on to the next; a step that is killed at its limit
# (rc 124 / 137) ends the batch: no further GPU step after a hang.
out=$1; shift
lim=$1; shift
mkdir -p "$out"
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "[batch] step $i: $cmd" | tee -a "$out/progress.log"
  timeout -k 10 "$lim" bash -c "$cmd" > "$out/step$i.log" 2>&1
  rc=$?
  echo "[batch] step $i rc=$rc" | tee -a "$out/progress.log"
  tail -n 15 "$out/step$i.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "[batch] step $i hit its limit: stopping" | tee -a "$out/progress.log"
    exit 1
  fi
done
exit 0
