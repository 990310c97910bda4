#!/bin/bash
# size-aware small evaluation: same bits (n=64 lml -48.255586860566424, n=128 -60.60689101428038 before), rates per n
OUT=gpurun_out/r05_v
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_parity.py tests/test_gpu_estimator.py -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/tests.txt
for n in 24 48 64 80 100 112 128; do
  r=$(timeout -k 10 120 python3 tools/fit_rate.py 16 $n 2>&1 | grep fits/s); echo "n=$n: $r" | tee -a $OUT/log.txt
done
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 64 1 16 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
