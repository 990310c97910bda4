#!/bin/bash
# crowded device, n = 1024: persistent task-queue launches vs the launch-per-product path
OUT=gpurun_out/r05_t
mkdir -p $OUT
for dag in 1 0; do
  echo "== HBEGP_DAG=$dag GPU_MAX_HW_QUEUES=16" | tee -a $OUT/log.txt
  HBEGP_DAG=$dag GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 4 8 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
done
echo "== HBEGP_DAG=0 GPU_MAX_HW_QUEUES=16 n=512" | tee -a $OUT/log.txt
HBEGP_DAG=0 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 512 1 4 8 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
