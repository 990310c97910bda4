#!/bin/bash
# round 5: small fits of several host threads share their launches (SmallBatcher): tests, solo rates, concurrent rates
OUT=gpurun_out/r05_m
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/tests.txt
grep -q "failed\|error" $OUT/tests.txt && exit 1
for n in 64 128; do
  r=$(timeout -k 10 120 python3 tools/fit_rate.py 16 $n 2>&1 | grep fits/s); echo "solo n=$n: $r" | tee -a $OUT/log.txt
  r=$(HBEGP_SMALL_BATCH_US=0 timeout -k 10 120 python3 tools/fit_rate.py 16 $n 2>&1 | grep fits/s); echo "solo n=$n window 0: $r" | tee -a $OUT/log.txt
done
for q in 4 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/log.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 1 4 8 16 32 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
done
echo "== window 0 (every fit its own launch), 16 queues" | tee -a $OUT/log.txt
HBEGP_SMALL_BATCH_US=0 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 128 4 16 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
echo "== n=64, default queues" | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/concurrent_fits.py 64 1 16 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
