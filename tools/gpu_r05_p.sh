#!/bin/bash
OUT=gpurun_out/r05_p
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/tests.txt
for q in 4 16; do
  for pr in 1 0; do
    echo "== GPU_MAX_HW_QUEUES=$q batch streams of high priority: $pr" | tee -a $OUT/log.txt
    HBEGP_BATCH_STREAM_PRIORITY=$pr FIT_PHASES=1 GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 1 4 8 16 32 2>&1 | grep "fits/s\|per fit" | tee -a $OUT/log.txt
  done
done
