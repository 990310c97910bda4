#!/bin/bash
OUT=gpurun_out/r05_ag
mkdir -p $OUT; rm -f $OUT/log.txt
python3 tools/dump_workload.py 128 $OUT/wl128.bin
timeout -k 10 600 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_cpp.py tests/test_gpu_estimator.py tests/test_gpu_parity.py -q -p no:cacheprovider 2>&1 | tail -1 | tee -a $OUT/log.txt
HBEGP_TIMING=1 timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin 1 4 8 16 2> $OUT/err.txt | tee -a $OUT/log.txt
python3 - $OUT/err.txt <<'PY' | tee -a $OUT/log.txt
import re, sys
held = [float(m.group(1)) for m in re.finditer(r"turns held for ([\d.]+) ms", open(sys.argv[1]).read())]
print(f"  turns held: first 20 fits (alone) {sum(held[1:21]) / 20:.3f} ms, last 60 (16 threads) {sum(held[-60:]) / 60:.3f} ms")
PY
timeout -k 10 120 python3 tools/fit_rate.py 16 128 2>&1 | grep fits/s | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/batch_soak.py 16 200 2>&1 | grep "batch soak" | tee -a $OUT/log.txt
