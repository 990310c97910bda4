#!/bin/bash
OUT=gpurun_out/call10
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -4 $OUT/gputest.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30; fi
timeout -k 10 300 python3 tools/profile_predict.py 2>&1 | grep -v amdgpu.ids | tail -5
exit $rc
