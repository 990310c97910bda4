#!/bin/bash
# Round 5, call I: the runs of a small fit as workgroups of one launch; hardware queue counts for fits side by side.
OUT=$PWD/gpurun_out/r5i
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_parity.py tests/test_gpu_estimator.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 120 python3 tools/small_fit_rate.py 64 100 128 2>&1 | grep -v amdgpu
for q in 4 16 32 64; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 1 4 16 32 2>&1 | grep -v amdgpu.ids | sed "s/^/hwq=$q /" | tee -a $OUT/concurrent.txt
done
for q in 16 32; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 4 6 2>&1 | grep -v amdgpu.ids | sed "s/^/hwq=$q /" | tee -a $OUT/concurrent.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 512 1 4 2>&1 | grep -v amdgpu.ids | sed "s/^/hwq=$q /" | tee -a $OUT/concurrent.txt
done
