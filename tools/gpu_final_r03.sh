#!/bin/bash
# Round-3 closing run: GPU suite, smoke, reproducibility evidence on both paths, C3 bench line, profiles (tools/refresh_profiles.sh r03).
OUT=$PWD/gpurun_out/final_r03
mkdir -p $OUT
echo "[1] GPU suite + smoke" | tee $OUT/progress.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -4 $OUT/gputest.txt | tee -a $OUT/progress.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30 | tee -a $OUT/progress.txt; exit $rc; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/progress.txt
echo "[2] fit_bits: 16 identical fixed-work fits, task queue (config M) and launch path (n=2048)" | tee -a $OUT/progress.txt
timeout -k 10 300 python3 tools/fit_bits.py 16 2>&1 | grep -v amdgpu.ids | tee $OUT/fit_bits_dag.txt | tee -a $OUT/progress.txt
HBEGP_DAG=0 timeout -k 10 300 python3 tools/fit_bits.py 16 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/fit_bits_launch.txt | tee -a $OUT/progress.txt
HBEGP_DAG=0 timeout -k 10 300 python3 tools/fit_bits.py 8 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/fit_bits_launch4096.txt | tee -a $OUT/progress.txt
echo "[3] bench --workload C3" | tee -a $OUT/progress.txt
timeout -k 10 600 python3 bench.py --workload C3 --steps 2 --warmup 1 > $OUT/bench_c3.json 2> $OUT/bench_c3.err || { tail -5 $OUT/bench_c3.err | tee -a $OUT/progress.txt; }
cut -c1-700 $OUT/bench_c3.json | tee -a $OUT/progress.txt
echo "[4] profiles" | tee -a $OUT/progress.txt
bash tools/refresh_profiles.sh r03 > $OUT/refresh.log 2>&1
tail -30 $OUT/refresh.log | cut -c1-400 | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
