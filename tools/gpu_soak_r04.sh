#!/bin/bash
# round-4 soak: concurrent evaluations (3 slots, different theta at once) against the quiet device, bit for bit -- the default
# paths after this round's changes: row-progressive plan with 32x64 chain tiles (n = 1536, 2048), divide-and-conquer plan with
# chain tiles (n = 4096), launch path driven through pinned memory (n = 512)
OUT=gpurun_out/soak_r04
mkdir -p $OUT
rm -f $OUT/progress.txt
run() { name=$1; shift; echo "=== $name" | tee -a $OUT/progress.txt; env "$@" > $OUT/$name.txt 2>&1 || echo FAILED | tee -a $OUT/progress.txt; grep -v amdgpu.ids $OUT/$name.txt | head -3 | cut -c1-300 | tee -a $OUT/progress.txt; }
run prog_rand_1536  timeout -k 10 200 python3 tools/nondet_hunt.py 1536 5000 rand 2
run prog_rand_2048  timeout -k 10 200 python3 tools/nondet_hunt.py 2048 4000 rand 2
run prog_same_2048  timeout -k 10 200 python3 tools/nondet_hunt.py 2048 12000 same 0
run dc_rand_4096    timeout -k 10 250 python3 tools/nondet_hunt.py 4096 1500 rand 2
run launch_rand_512 timeout -k 10 200 python3 tools/nondet_hunt.py 512 8000 rand 2
echo done | tee -a $OUT/progress.txt
