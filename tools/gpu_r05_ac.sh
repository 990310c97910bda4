#!/bin/bash
OUT=gpurun_out/r05_ac
mkdir -p $OUT
for i in 1 2 3; do
  HBEGP_SMALL_BATCH_LOG=1 GPU_MAX_HW_QUEUES=16 FIT_PHASES=1 timeout -k 10 200 python3 tools/concurrent_fits.py 128 4 2> $OUT/py_$i.err | grep "fits/s\|per fit"
  grep "small-fit batch" $OUT/py_$i.err | awk '{print $3" runs, waited "$6" us"}' | sort | uniq -c | sort -rn | head -8
done
