#!/bin/bash
# round 5, early look + operand touch (HBEGP_DAG_EARLY: 0 = round 4's hand-off, 1 = early look only, 3 = + touch): the bitwise
# tests of the task queue, then fit rates of config M alternating the three settings, then one evaluation alone under each.
OUT=gpurun_out/r05_k
mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/tests.txt
grep -q "passed" $OUT/tests.txt || exit 1
grep -q "failed" $OUT/tests.txt && exit 1
for round in 1 2 3; do
  for e in 0 1 3; do
    r=$(HBEGP_DAG_EARLY=$e timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s)
    echo "round $round early=$e: $r" | tee -a $OUT/fit_rates.txt
  done
done
for e in 0 1 3; do
  cat > $OUT/one.py <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from hbetune_rs_amd import gpr, synth
for n in (1024, 4096):
    w = synth.make_workload("M", n=n)
    prob = gpr.Problem(w["X"], w["y"])
    ph = prob.time_eval(w["theta"], reps=5)
    print(f"early=$e n={n}: one evaluation {ph['eval_graph_ms']:.3f} ms", flush=True)
    prob.close()
PY
  HBEGP_DAG_EARLY=$e timeout -k 10 120 python3 $OUT/one.py 2>&1 | grep "one evaluation" | tee -a $OUT/fit_rates.txt
done
