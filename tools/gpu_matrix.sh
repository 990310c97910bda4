#!/bin/bash
# the GPU suite under the library's alternative configurations (each must be green): plan validation on every problem, eager
# launches instead of graph replay, launch path everywhere, task queue everywhere; round 4: copy nodes + stream sync instead of the
# pinned-memory protocol, divide-and-conquer / row-progressive inverse everywhere, 64x64 chain tiles, no LDS path, host-driven small
# fits; round 5: 16 hardware queues.  Usage: gpu_matrix.sh [first [last]]  (configurations first..last of the list, 0-based)
OUT=gpurun_out/matrix
mkdir -p $OUT
CFGS=("HBEGP_DAG_VALIDATE=1" "HBEGP_NO_GRAPH=1" "HBEGP_DAG=0" "HBEGP_DAG=1" "HBEGP_DAG_RL=0" "HBEGP_HOSTIO=0" "HBEGP_DAG_PROG=0" "HBEGP_DAG_PROG=1" "HBEGP_DAG_CHAIN32=0" "HBEGP_SMALL=0" "HBEGP_SMALL_FIT=0" "GPU_MAX_HW_QUEUES=16")
first=${1:-0}; last=${2:-$((${#CFGS[@]} - 1))}
for i in $(seq $first $last); do
  cfg=${CFGS[$i]}
  name=$(echo $cfg | tr '=' '_')
  env $cfg timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/$name.txt 2>&1
  echo "$cfg: $(tail -1 $OUT/$name.txt)" | tee -a $OUT/progress_$first.txt
  grep -n "^FAILED" $OUT/$name.txt | head -10 | tee -a $OUT/progress_$first.txt
done
