#!/bin/bash
# usage: repeat_parity.sh <label> <count> [pytest args]: runs the parity tests <count> times, prints failures
label=$1; count=$2; shift 2
mkdir -p gpurun_out/rep
for i in $(seq 1 $count); do
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x "$@" > gpurun_out/rep/${label}_$i.txt 2>&1
  echo "$label run $i: $(tail -1 gpurun_out/rep/${label}_$i.txt)"
  grep -E "^FAILED" gpurun_out/rep/${label}_$i.txt
done
