#!/bin/bash
OUT=gpurun_out/r05_q
mkdir -p $OUT
for q in 4 16; do
  for us in 1000 3000; do
    echo "== GPU_MAX_HW_QUEUES=$q window $us us" | tee -a $OUT/log.txt
    HBEGP_SMALL_BATCH_US=$us FIT_PHASES=1 GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 4 8 16 32 2>&1 | grep "fits/s\|per fit" | tee -a $OUT/log.txt
  done
done
