#!/bin/bash
OUT=gpurun_out/call7
mkdir -p $OUT
echo "=== trace default wg=96" | tee -a $OUT/progress.txt
HBEGP_DAG_LAUUM_SPLIT=0 HBEGP_DAG_WG=96 HBEGP_DAG_TRACE=$OUT/trace.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep -v "amdgpu.ids\|^gaps" | tee -a $OUT/progress.txt
rm -f $OUT/trace.txt
echo "=== small n probe again" | tee -a $OUT/progress.txt
timeout -k 10 200 python3 tools/small_n_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee -a $OUT/progress.txt
echo "=== oversubscription sweep (called-leaf build)" | tee -a $OUT/progress.txt
for ov in 106 112 118 125; do
  HBEGP_LIB=build/var/libhbegp_noinline.so HBEGP_DAG_OVERSUB=$ov timeout -k 10 200 python3 tools/fit_rate.py 6 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/oversub $ov: /" | tee -a $OUT/progress.txt
done
