#!/bin/bash
OUT=gpurun_out/r05_aa
mkdir -p $OUT; rm -f $OUT/log.txt
python3 tools/dump_workload.py 128 $OUT/wl128.bin
hist() { grep "small-fit batch" $1 | awk '{print $3}' | sort -n | uniq -c | sort -k2 -n | tr '\n' ';'; echo; }
for q in 4 16; do
  echo "== native GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/log.txt
  HBEGP_SMALL_BATCH_LOG=1 GPU_MAX_HW_QUEUES=$q timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin 1 2 4 8 16 32 2> $OUT/native_q$q.err | tee -a $OUT/log.txt
  hist $OUT/native_q$q.err | tee -a $OUT/log.txt
done
echo "== python GPU_MAX_HW_QUEUES=16" | tee -a $OUT/log.txt
HBEGP_SMALL_BATCH_LOG=1 GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python3 tools/concurrent_fits.py 128 4 16 2> $OUT/py.err | grep fits/s | tee -a $OUT/log.txt
hist $OUT/py.err | tee -a $OUT/log.txt
echo "== native, recent 1500 us, default queues" | tee -a $OUT/log.txt
HBEGP_SMALL_BATCH_RECENT_US=1500 timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin 4 16 2>/dev/null | tee -a $OUT/log.txt
echo "== native, recent 0 (as before), default queues" | tee -a $OUT/log.txt
HBEGP_SMALL_BATCH_RECENT_US=0 timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin 4 16 2>/dev/null | tee -a $OUT/log.txt
timeout -k 10 300 python3 tools/batch_soak.py 16 200 2>&1 | grep "batch soak" | tee -a $OUT/log.txt
