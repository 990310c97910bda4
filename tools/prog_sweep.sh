#!/bin/bash
# sweep of the progressive plan's knobs: one evaluation alone + fit rate, per n
for cfg in "0 -1 -1 1" "1 -1 -1 1" "1 -1 -1 0" "1 1 2 0" "1 2 2 0" "1 1 3 0" "1 2 4 0"; do
  set -- $cfg
  for n in ${NS:-4096 2048}; do
    echo "PROG=$1 UNEAR=$2 KNEAR=$3 SMALL=$4: $(HBEGP_DAG_PROG=$1 HBEGP_DAG_PROG_UNEAR=$2 HBEGP_DAG_PROG_KNEAR=$3 HBEGP_DAG_PROG_SMALL=$4 timeout -k 10 200 python3 tools/split_probe.py $n 2>&1 | grep -v amdgpu | cut -c1-120)"
  done
done
