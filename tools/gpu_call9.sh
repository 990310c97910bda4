#!/bin/bash
OUT=gpurun_out/call9
mkdir -p $OUT
echo "[1] GPU suite" | tee $OUT/progress.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -4 $OUT/gputest.txt | tee -a $OUT/progress.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30 | tee -a $OUT/progress.txt; exit $rc; fi
echo "[2] small n probe: fused tail on / off" | tee -a $OUT/progress.txt
timeout -k 10 200 python3 tools/small_n_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/progress.txt
HBEGP_SMALLTAIL=0 timeout -k 10 200 python3 tools/small_n_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-130 | tee -a $OUT/progress.txt
echo "[3] fit rates at small n, fused tail on / off" | tee -a $OUT/progress.txt
for n in 100 128 200 256; do
  for st in 1 0; do
    HBEGP_SMALLTAIL=$st timeout -k 10 100 python3 tools/fit_rate.py 5 $n 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/n=$n smalltail=$st: /" | tee -a $OUT/progress.txt
  done
done
