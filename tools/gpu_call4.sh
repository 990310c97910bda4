#!/bin/bash
OUT=gpurun_out/call4
mkdir -p $OUT
echo "[1] fastmath probe" | tee $OUT/progress.txt
timeout -k 10 120 tools/fastmath_probe 2>&1 | tee -a $OUT/progress.txt
echo "[2] GPU suite" | tee -a $OUT/progress.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -4 $OUT/gputest.txt | tee -a $OUT/progress.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30 | tee -a $OUT/progress.txt; exit $rc; fi
echo "[3] GPU dag tests with the called-leaf build" | tee -a $OUT/progress.txt
HBEGP_LIB=build/var/libhbegp_noinline.so timeout -k 10 600 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_fit.py -x -q -p no:cacheprovider > $OUT/gputest_noinline.txt 2>&1 || { tail -30 $OUT/gputest_noinline.txt | tee -a $OUT/progress.txt; exit 1; }
tail -2 $OUT/gputest_noinline.txt | tee -a $OUT/progress.txt
echo "[4] bench, inlined vs called leaf" | tee -a $OUT/progress.txt
for rep in 1 2; do
for lib in default noinline; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err || { tail -5 $OUT/bench_${lib}_$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/bench_${lib}_$rep.json')); r=d['roofline']
print('$lib $rep value %.4f ms %.1f dag_ms %.3f single %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['single_stream_eval_ms']), [ (k['kernel'][:12], k['ms_per_eval']) for k in r['kernels']])
print('   f32', json.dumps(d.get('f32_side_line',{}).get('fits_per_s')), 'eval_ms', d.get('f32_side_line',{}).get('eval_ms'))
print('   small', json.dumps(d.get('small_n_side_line')))
" | tee -a $OUT/progress.txt
done
done
unset HBEGP_LIB
echo "[5] full bench with CPU baseline" | tee -a $OUT/progress.txt
timeout -k 10 900 python3 bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -25 $OUT/bench_full.err | tee -a $OUT/progress.txt
python3 -c "
import json; d=json.load(open('$OUT/bench_full.json')); print(json.dumps(d['cpu_baseline'])[:1500])" | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
