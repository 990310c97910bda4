#!/bin/bash
# Round 5, call C: three register sets (PIPE 2: loads three stages ahead) -- the tile alone, then the fit rate, alternating.
OUT=$PWD/gpurun_out/r5c
mkdir -p $OUT
for depth in 2048 512; do
  timeout -k 5 120 ./tools/tile_ubench $depth 8 7 256 2>&1 | grep -v amdgpu.ids | grep "depth\|DIFF" | tee -a $OUT/tile_ubench.txt
done
for round in 1 2 3; do
  for v in base p2128 p2big; do
    if [ $v = base ]; then lib=hbetune_rs_amd/libhbegp.so; else lib=build/var/libhbegp_$v.so; fi
    r=$(HBEGP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s)
    echo "round $round $v: $r" | tee -a $OUT/fit_rates.txt
  done
done
HBEGP_LIB=$PWD/build/var/libhbegp_p2big.so timeout -k 10 600 python3 -m pytest tests/test_gpu_dag.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $OUT/fit_rates.txt
