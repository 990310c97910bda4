#!/bin/bash
# round 5, concurrent fits on one GPU: hardware-queue count and workgroups per launch
OUT=gpurun_out/r05_l
mkdir -p $OUT
run() { # label env... -- n k...
  echo "== $*" | tee -a $OUT/log.txt
  env "$@" 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
}
for q in 16 32 64; do
  run GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 tools/concurrent_fits.py 128 4 8 16
done
for q in 16 32; do
  for os in 112 75 56; do
    run GPU_MAX_HW_QUEUES=$q HBEGP_DAG_OVERSUB=$os timeout -k 10 200 python3 tools/concurrent_fits.py 1024 4
  done
done
run GPU_MAX_HW_QUEUES=32 HBEGP_DAG_OVERSUB=75 timeout -k 10 200 python3 tools/concurrent_fits.py 1024 6 8
