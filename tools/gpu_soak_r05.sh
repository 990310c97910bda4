#!/bin/bash
# round 5 closing soak of the shipped build: concurrent evaluations against the quiet device bit for bit (task queue, launch
# path), the small-fit batcher under 16-32 threads, fits of one size side by side, the GPU suite three more times
OUT=gpurun_out/soak_r05
mkdir -p $OUT; rm -f $OUT/progress.txt
run() { name=$1; shift; echo "=== $name" | tee -a $OUT/progress.txt; env "$@" > $OUT/$name.txt 2>&1 || echo FAILED | tee -a $OUT/progress.txt; grep -v amdgpu.ids $OUT/$name.txt | tail -2 | cut -c1-300 | tee -a $OUT/progress.txt; }
run dag_same_4096    timeout -k 10 200 python3 tools/nondet_hunt.py 4096 6000 same 0
run dag_rand_2048    timeout -k 10 200 python3 tools/nondet_hunt.py 2048 3000 rand 2
run launch_same_2048 HBEGP_DAG=0 timeout -k 10 200 python3 tools/nondet_hunt.py 2048 10000 same 0
run batch_soak_16    timeout -k 10 300 python3 tools/batch_soak.py 16 400
run batch_soak_32    timeout -k 10 300 python3 tools/batch_soak.py 32 200
run batch_soak_16q   GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/batch_soak.py 16 400
run side_by_side_900 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 900 4 6
run fit_bits_M       timeout -k 10 300 python3 tools/fit_bits.py 16
for i in 1 2 3; do
  run suite_$i timeout -k 10 400 python3 -m pytest tests -m gpu -q -p no:cacheprovider
done
echo done | tee -a $OUT/progress.txt
