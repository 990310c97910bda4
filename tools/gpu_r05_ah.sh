#!/bin/bash
# workgroups per task-queue launch with three slots, final build: HBEGP_DAG_OVERSUB (multiples of 8) and forced counts in between (HBEGP_DAG_WG)
OUT=gpurun_out/r05_ah
mkdir -p $OUT; rm -f $OUT/log2.txt
for round in 1 2; do
  for wg in 0 92 94 98 100; do
    r=$(HBEGP_DAG_WG=$wg timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s); echo "round $round forced workgroups=$wg (0: default 96): $r" | tee -a $OUT/log2.txt
  done
done
