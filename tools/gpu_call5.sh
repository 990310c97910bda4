#!/bin/bash
OUT=gpurun_out/call5
mkdir -p $OUT
echo "[1] GPU dag + fit tests, both builds" | tee $OUT/progress.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_fit.py -x -q -p no:cacheprovider > $OUT/t_default.txt 2>&1 || { tail -30 $OUT/t_default.txt | tee -a $OUT/progress.txt; exit 1; }
tail -2 $OUT/t_default.txt | tee -a $OUT/progress.txt
HBEGP_LIB=build/var/libhbegp_noinline.so timeout -k 10 600 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_fit.py -x -q -p no:cacheprovider > $OUT/t_noinline.txt 2>&1 || { tail -30 $OUT/t_noinline.txt | tee -a $OUT/progress.txt; exit 1; }
tail -2 $OUT/t_noinline.txt | tee -a $OUT/progress.txt
echo "[2] bench, inlined vs called leaf" | tee -a $OUT/progress.txt
for rep in 1 2; do
for lib in default noinline; do
  if [ $lib = default ]; then unset HBEGP_LIB; else export HBEGP_LIB=build/var/libhbegp_$lib.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err || { tail -5 $OUT/bench_${lib}_$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/bench_${lib}_$rep.json')); r=d['roofline']
print('$lib $rep value %.4f ms %.1f dag_ms %.3f single %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['single_stream_eval_ms']), [ (k['kernel'][:12], k['ms_per_eval']) for k in r['kernels']])
" | tee -a $OUT/progress.txt
done
done
unset HBEGP_LIB
echo "[3] contention probe" | tee -a $OUT/progress.txt
timeout -k 10 300 python3 tools/contention_probe.py 4096 > $OUT/contention.json 2> $OUT/contention.err
python3 -c "
import json; d=json.load(open('$OUT/contention.json')); print(d['tflops'], d['slowdown_B_over_A'], d['A_one_slot_same_size']['dag_ms'], d['B_three_slots']['dag_ms'], d['C_one_slot_whole_chip']['dag_ms'])" | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
