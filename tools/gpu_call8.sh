#!/bin/bash
OUT=gpurun_out/call8
mkdir -p $OUT
echo "[1] GPU dag + fit + fullsize tests" | tee $OUT/progress.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_fit.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider > $OUT/t_default.txt 2>&1 || { tail -30 $OUT/t_default.txt | tee -a $OUT/progress.txt; exit 1; }
tail -2 $OUT/t_default.txt | tee -a $OUT/progress.txt
echo "[2] bench: 32-deep stages (default) vs 16-deep (k16)" | tee -a $OUT/progress.txt
for rep in 1 2; do
for lib in default k16; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err || { tail -5 $OUT/bench_${lib}_$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/bench_${lib}_$rep.json')); r=d['roofline']
print('$lib $rep value %.4f ms %.1f dag_ms %.3f single %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['single_stream_eval_ms']), [ (k['kernel'][:12], k['ms_per_eval']) for k in r['kernels']], 'C4?', d.get('small_n_side_line',{}).get('n=1024'))
" | tee -a $OUT/progress.txt
done
done
unset HBEGP_LIB
echo "[3] trace 96 wg (default)" | tee -a $OUT/progress.txt
HBEGP_DAG_LAUUM_SPLIT=0 HBEGP_DAG_WG=96 HBEGP_DAG_TRACE=$OUT/trace.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep -v "amdgpu.ids\|^gaps\|^gemm\|^leaf k" | tee -a $OUT/progress.txt
rm -f $OUT/trace.txt
echo "[4] n=8192 single evaluation, both" | tee -a $OUT/progress.txt
timeout -k 10 200 python3 tools/profile_eval.py C4 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $OUT/progress.txt
HBEGP_LIB=build/var/libhbegp_k16.so timeout -k 10 200 python3 tools/profile_eval.py C4 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
