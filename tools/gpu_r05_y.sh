#!/bin/bash
OUT=gpurun_out/r05_y
mkdir -p $OUT; rm -f $OUT/log.txt
for i in 1 2 3; do
  timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 2>&1 | grep "fits/s" | sed "s/^/default run $i: /" | tee -a $OUT/log.txt
done
for i in 1 2; do
  HBEGP_FUSE_ALPHA=0 timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 2>&1 | grep "fits/s" | sed "s/^/unfused run $i: /" | tee -a $OUT/log.txt
done
for i in 1 2; do
  timeout -k 10 300 python3 tools/fit_rate.py 8 1024 2>&1 | grep "fits/s" | sed "s/^/fit_rate run $i: /" | tee -a $OUT/log.txt
done
AMD_LOG_LEVEL=0 GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 2>&1 | grep "fits/s" | sed "s/^/8 queues: /" | tee -a $OUT/log.txt
