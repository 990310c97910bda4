#!/bin/bash
# after taking the stream adoption out again: headline, solo / concurrent mid-n, small-n concurrency
OUT=gpurun_out/r05_x
mkdir -p $OUT; rm -f $OUT/log.txt
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-side-lines 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench:', d['value'], d['ms_per_step'])" | tee -a $OUT/log.txt
for q in 4 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/log.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 1024 1 4 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 2048 1 2 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py 128 1 4 8 16 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
done
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-side-lines 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench:', d['value'], d['ms_per_step'])" | tee -a $OUT/log.txt
