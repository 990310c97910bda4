#!/bin/bash
OUT=gpurun_out/call15
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_dag.py -x -q -p no:cacheprovider > $OUT/t.txt 2>&1 || { tail -30 $OUT/t.txt; exit 1; }
tail -1 $OUT/t.txt
for rep in 1 2 3; do
for lib in default head; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  timeout -k 10 200 python3 tools/fit_rate.py 6 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$lib $rep: /"
done
done
for lib in default head; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  echo "== $lib"
  HBEGP_DAG_LAUUM_SPLIT=0 HBEGP_DAG_WG=96 HBEGP_DAG_TRACE=$OUT/trace.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep "dag_ms\|makespan\|tile tasks\|k=2048:\|k=128:\|k=1024:"
  rm -f $OUT/trace.txt
done
