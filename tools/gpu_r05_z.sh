#!/bin/bash
# native host threads vs Python threads on the small-fit batcher; headline after the alpha loads were batched
OUT=gpurun_out/r05_z
mkdir -p $OUT; rm -f $OUT/log.txt
python3 tools/dump_workload.py 128 $OUT/wl128.bin
python3 tools/dump_workload.py 64 $OUT/wl64.bin
for q in 4 16; do
  for i in 1 2 3; do
    echo "== native, GPU_MAX_HW_QUEUES=$q, run $i" | tee -a $OUT/log.txt
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin 1 4 8 16 32 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
  done
done
echo "== native n=64, default queues" | tee -a $OUT/log.txt
timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl64.bin 1 16 32 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
for i in 1 2 3; do
  echo "== python threads, GPU_MAX_HW_QUEUES=16, run $i" | tee -a $OUT/log.txt
  GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python3 tools/concurrent_fits.py 128 16 2>&1 | grep "fits/s" | tee -a $OUT/log.txt
done
for i in 1 2; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-side-lines 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench:', d['value'], d['ms_per_step'])" | tee -a $OUT/log.txt
  HBEGP_FUSE_ALPHA=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-side-lines 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench, alpha_reduce in its own launch:', d['value'], d['ms_per_step'])" | tee -a $OUT/log.txt
done
