#!/bin/bash
# crossover of the progressive plan against the divide-and-conquer inverse: one evaluation alone + fit rate per n, f64 and f32
for n in ${NS:-1536 2048 2560 3072 3584}; do
  for p in 0 1; do
    echo "PROG=$p f64: $(HBEGP_DAG_PROG=$p HBEGP_DAG_PROG_SMALL=0 timeout -k 10 200 python3 tools/split_probe.py $n 2>&1 | grep -v amdgpu | cut -c1-110)"
  done
done
