#!/bin/bash
# workgroups per task-queue launch with three slots (HBEGP_DAG_OVERSUB 100 / 112 / 120 / 130 -> 88 / 96 / 104 / 112), final build
OUT=gpurun_out/r05_ah
mkdir -p $OUT; rm -f $OUT/log.txt
for round in 1 2 3; do
  for os in 112 100 120 130; do
    r=$(HBEGP_DAG_OVERSUB=$os timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s); echo "round $round oversub=$os: $r" | tee -a $OUT/log.txt
  done
done
