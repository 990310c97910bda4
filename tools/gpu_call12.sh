#!/bin/bash
OUT=gpurun_out/call12
mkdir -p $OUT
echo "[1] GPU dag + fit + fullsize tests" | tee $OUT/progress.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_fit.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider > $OUT/t_default.txt 2>&1 || { tail -30 $OUT/t_default.txt | tee -a $OUT/progress.txt; exit 1; }
tail -2 $OUT/t_default.txt | tee -a $OUT/progress.txt
echo "[2] fit rate: next-tile preload (default) vs without (nopre), alternating" | tee -a $OUT/progress.txt
for rep in 1 2 3; do
for lib in default nopre; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  timeout -k 10 200 python3 tools/fit_rate.py 6 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$lib $rep: /" | tee -a $OUT/progress.txt
done
done
unset HBEGP_LIB
echo "[3] trace 96 wg (default, nopre)" | tee -a $OUT/progress.txt
for lib in default nopre; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  echo "== $lib" | tee -a $OUT/progress.txt
  HBEGP_DAG_LAUUM_SPLIT=0 HBEGP_DAG_WG=96 HBEGP_DAG_TRACE=$OUT/trace.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep -v "amdgpu.ids\|^gaps\|^gemm\|^leaf k" | tee -a $OUT/progress.txt
  rm -f $OUT/trace.txt
done
unset HBEGP_LIB
echo "[4] single evaluation 256 wg" | tee -a $OUT/progress.txt
timeout -k 10 200 python3 tools/profile_eval.py M 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/progress.txt
HBEGP_LIB=build/var/libhbegp_nopre.so timeout -k 10 200 python3 tools/profile_eval.py M 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
