#!/bin/bash
# host-side phases of small fits one thread at a time (HBEGP_SMALL_HOST_SERIAL: threads up to which the turn-taking applies; 0 = never)
OUT=gpurun_out/r05_ad
mkdir -p $OUT; rm -f $OUT/log.txt
python3 tools/dump_workload.py 128 $OUT/wl128.bin
timeout -k 10 400 python3 -m pytest tests/test_gpu_fit.py -q -p no:cacheprovider 2>&1 | tail -1 | tee -a $OUT/log.txt
for rep in 1 2; do
  for v in 16 0; do
    echo "== native threads, turn-taking up to $v threads (run $rep)" | tee -a $OUT/log.txt
    HBEGP_SMALL_HOST_SERIAL=$v timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin 1 2 4 8 16 24 32 2>/dev/null | tee -a $OUT/log.txt
  done
done
for v in 16 0; do
  echo "== python threads, turn-taking up to $v threads" | tee -a $OUT/log.txt
  HBEGP_SMALL_HOST_SERIAL=$v timeout -k 10 200 python3 tools/concurrent_fits.py 128 4 16 2>/dev/null | grep fits/s | tee -a $OUT/log.txt
done
timeout -k 10 300 python3 tools/batch_soak.py 16 300 2>&1 | grep "batch soak" | tee -a $OUT/log.txt
