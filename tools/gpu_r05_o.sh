#!/bin/bash
OUT=gpurun_out/r05_o
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_api.py -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/tests.txt
for q in 4 16 32 64; do
  echo "== GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/log.txt
  FIT_PHASES=1 GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 8 16 2>&1 | grep "fits/s\|per fit" | tee -a $OUT/log.txt
done
echo "== GPU_MAX_HW_QUEUES=16, models with streams of their own" | tee -a $OUT/log.txt
HBEGP_MODEL_OWN_STREAM=1 FIT_PHASES=1 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 128 8 16 2>&1 | grep "fits/s\|per fit" | tee -a $OUT/log.txt
for v in 0 1; do
  r=$(HBEGP_MODEL_OWN_STREAM=$v timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s); echo "config M, model stream of its own = $v: $r" | tee -a $OUT/log.txt
done
