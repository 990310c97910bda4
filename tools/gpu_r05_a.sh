#!/bin/bash
# Round 5, call A: the tile function alone (stage-loop forms), then the fit rate of config M with the variant libraries,
# alternating, plus the bitwise task-queue tests under the variant that switches every tile shape over.
OUT=$PWD/gpurun_out/r5a
mkdir -p $OUT
echo "[1] tile_ubench" | tee $OUT/progress.txt
for depth in 2048 512 128; do
  timeout -k 5 120 ./tools/tile_ubench $depth 8 7 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/tile_ubench.txt
done
cat $OUT/tile_ubench.txt >> $OUT/progress.txt
echo "[2] fit rate, config M, alternating libraries" | tee -a $OUT/progress.txt
for round in 1 2 3; do
  for v in base p128 pbig pall; do
    if [ $v = base ]; then lib=hbetune_rs_amd/libhbegp.so; else lib=build/var/libhbegp_$v.so; fi
    [ -f $lib ] || continue
    r=$(HBEGP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s)
    echo "round $round $v: $r" | tee -a $OUT/fit_rates.txt
  done
done
echo "[3] bitwise tests of the task queue under pall" | tee -a $OUT/progress.txt
if [ -f build/var/libhbegp_pall.so ]; then
  HBEGP_LIB=$PWD/build/var/libhbegp_pall.so timeout -k 10 600 python3 -m pytest tests/test_gpu_dag.py -x -q -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT/progress.txt
fi
echo done | tee -a $OUT/progress.txt
