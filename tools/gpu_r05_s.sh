#!/bin/bash
OUT=gpurun_out/r05_s
mkdir -p $OUT
rm -f $OUT/tests.txt
for i in 1 2 3 4; do
  timeout -k 10 400 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -p no:cacheprovider > $OUT/suite_$i.txt 2>&1; tail -1 $OUT/suite_$i.txt | tee -a $OUT/tests.txt
done
HBEGP_NO_GRAPH=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -q -s -k fresh_pool -p no:cacheprovider 2>&1 | grep "^old clear\|^.old clear\|passed\|failed" | cut -c1-250 | tee -a $OUT/tests.txt
