#!/bin/bash
OUT=gpurun_out/r05_r
mkdir -p $OUT
rm -f $OUT/log.txt $OUT/tests.txt
for i in 1 2 3; do
  timeout -k 10 400 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -p no:cacheprovider > $OUT/suite_$i.txt 2>&1; tail -1 $OUT/suite_$i.txt | tee -a $OUT/tests.txt
done
HBEGP_NO_GRAPH=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -q -s -k fresh_pool -p no:cacheprovider 2>&1 | grep "old clear\|passed\|failed" | cut -c1-200 | tee -a $OUT/tests.txt
for q in 4 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/log.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 1 4 8 16 32 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
done
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 64 1 16 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 4 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
