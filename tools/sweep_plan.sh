#!/bin/bash
# plan parameters re-swept with the round-3 task loop (fixed-work fits of config M, fits/s): each line one configuration
OUT=gpurun_out/sweep_plan
mkdir -p $OUT
for cfg in "BASE=1" "HBEGP_DAG_RL_GROUP=16" "HBEGP_DAG_RL_GROUP=8" "HBEGP_DAG_RL_NEAR=2" "HBEGP_DAG_SMALLH=2" "HBEGP_DAG_SMALLH=8" "HBEGP_DAG_CRIT=0" "HBEGP_DAG_CRIT=2" "HBEGP_DAG_OVERSUB=100" "HBEGP_DAG_OVERSUB=118" "HBEGP_MAX_CONCURRENT=2" "BASE=2"; do
  env $cfg timeout -k 10 120 python3 tools/fit_rate.py 5 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$cfg: /" | tee -a $OUT/progress.txt
done
