#!/bin/bash
OUT=gpurun_out/call14
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_dag.py tests/test_gpu_fit.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider > $OUT/t.txt 2>&1 || { tail -30 $OUT/t.txt; exit 1; }
tail -2 $OUT/t.txt
for cfg in M C4 C5; do
  for inl in 1 0; do
    HBEGP_DAG_LEAF_INLINE=$inl timeout -k 10 200 python3 tools/profile_eval.py $cfg 2>&1 | grep -v amdgpu.ids | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg inline=$inl eval_graph_ms', d['eval_graph_ms'], 'dag_ms', d['dag_ms'])"
  done
done
timeout -k 10 200 python3 tools/fit_rate.py 6 2>&1 | grep -v amdgpu.ids | tail -1
