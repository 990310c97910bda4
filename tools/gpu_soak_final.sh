#!/bin/bash
# closing soak of the shipped build: every concurrent evaluation against the quiet device, bit for bit
OUT=gpurun_out/soak_final
mkdir -p $OUT
run() { name=$1; shift; echo "=== $name" | tee -a $OUT/progress.txt; env "$@" > $OUT/$name.txt 2>&1 || echo FAILED | tee -a $OUT/progress.txt; grep -v amdgpu.ids $OUT/$name.txt | head -3 | cut -c1-300 | tee -a $OUT/progress.txt; }
run dag_same_4096   timeout -k 10 300 python3 tools/nondet_hunt.py 4096 12000 same 0
run dag_rand_4096   timeout -k 10 300 python3 tools/nondet_hunt.py 4096 1500 rand 2
run dag_rand_2048   timeout -k 10 300 python3 tools/nondet_hunt.py 2048 4000 rand 2
run launch_same_2048 HBEGP_DAG=0 timeout -k 10 300 python3 tools/nondet_hunt.py 2048 20000 same 0
run launch_walk_2048 HBEGP_DAG=0 timeout -k 10 300 python3 tools/nondet_hunt.py 2048 4000 walk 2
echo done | tee -a $OUT/progress.txt
