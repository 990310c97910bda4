#!/bin/bash
# the GPU suite under the library's alternative configurations (each must be green): plan validation on every problem, eager
# launches instead of graph replay, launch path everywhere, task queue everywhere
OUT=gpurun_out/matrix
mkdir -p $OUT
for cfg in "HBEGP_DAG_VALIDATE=1" "HBEGP_NO_GRAPH=1" "HBEGP_DAG=0" "HBEGP_DAG=1" "HBEGP_DAG_RL=0"; do
  name=$(echo $cfg | tr '=' '_')
  env $cfg timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/$name.txt 2>&1
  echo "$cfg: $(tail -1 $OUT/$name.txt)" | tee -a $OUT/progress.txt
  grep -n "^FAILED" $OUT/$name.txt | head -10 | tee -a $OUT/progress.txt
done
