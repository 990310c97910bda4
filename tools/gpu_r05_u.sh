#!/bin/bash
# crowded device, mid n (launch path by default below 6 blocks): would the task queue with the recursion plan (launch-path bits) do better?
OUT=gpurun_out/r05_u
mkdir -p $OUT
for n in 512 256; do
  for cfg in "HBEGP_DAG_MIN_BLOCKS=6" "HBEGP_DAG=1 HBEGP_DAG_RL=0" "HBEGP_DAG=1"; do
    echo "== n=$n $cfg GPU_MAX_HW_QUEUES=16" | tee -a $OUT/log.txt
    env $cfg GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py $n 1 4 8 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
  done
done
echo "== n=512 default queues, default path" | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/concurrent_fits.py 512 1 4 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
