#!/bin/bash
# Round 5, call H: slot streams from a pool of their own; hardware queues 4 (default) vs 16 for fits side by side.
OUT=$PWD/gpurun_out/r5h
mkdir -p $OUT
HBEGP_TIMING=1 timeout 120 python3 tools/fit_rate.py 4 2>&1 | grep -v amdgpu | tail -5
GPU_MAX_HW_QUEUES=16 timeout 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s | sed 's/^/16 hw queues: /'
timeout 120 python3 tools/fit_rate.py 6 256 2>&1 | grep fits/s
for q in 4 16; do
  for spec in "128 1 4 16" "256 1 4" "1024 1 2 4 8" "2048 1 2 4"; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py $spec 2>&1 | grep -v amdgpu.ids | sed "s/^/hwq=$q /" | tee -a $OUT/concurrent.txt
  done
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py -x -q -p no:cacheprovider 2>&1 | tail -3
