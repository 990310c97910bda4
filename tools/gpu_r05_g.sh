#!/bin/bash
# Round 5, call G: no null-stream operation / device-wide synchronisation on the fit path any more, crowded-device launch sizes:
# the whole GPU suite, then concurrent fits.
OUT=$PWD/gpurun_out/r5g
mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -3 $OUT/gputest.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30; exit $rc; fi
for spec in "128 1 4 16" "256 1 4 8" "512 1 2 4" "1024 1 2 4 8" "2048 1 2 4" "4096 1 2"; do
  timeout -k 10 300 python3 tools/concurrent_fits.py $spec 2>&1 | grep -v amdgpu.ids | tee -a $OUT/concurrent.txt
done
timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s
