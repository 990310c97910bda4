#!/bin/bash
# chain bias of the queue order (HBEGP_DAG_CHAIN_BIAS, microseconds of simulated time): one evaluation alone + three-run fits
for n in ${NS:-4096 2048}; do
  for p in 0 1; do
    for b in 0 5 10 20 40; do
      echo "n=$n PROG=$p BIAS=$b: $(HBEGP_DAG_PROG=$p HBEGP_DAG_CHAIN_BIAS=$b timeout -k 10 200 python3 tools/split_probe.py $n 2>&1 | grep -v amdgpu | sed 's/.*: one/one/' | cut -c1-80)"
    done
  done
done
