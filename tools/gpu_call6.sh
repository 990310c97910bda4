#!/bin/bash
OUT=gpurun_out/call6
mkdir -p $OUT
for lib in default oldloop; do
  for wg in 96 256; do
    if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
    echo "=== $lib wg=$wg" | tee -a $OUT/progress.txt
    HBEGP_DAG_LAUUM_SPLIT=0 HBEGP_DAG_WG=$wg HBEGP_DAG_TRACE=$OUT/trace_${lib}_$wg.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep -v "amdgpu.ids\|^gemm\|^gaps" | tee -a $OUT/progress.txt
    rm -f $OUT/trace_${lib}_$wg.txt
  done
done
