#!/bin/bash
# Round-4 closing run: GPU suite + smoke, the diagonal block's benches, small-n rates, reproducibility, C3 bench line, profiles.
OUT=$PWD/gpurun_out/final_r04
mkdir -p $OUT
ROOT=$PWD
echo "[1] GPU suite + smoke" | tee $OUT/progress.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -4 $OUT/gputest.txt | tee -a $OUT/progress.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30 | tee -a $OUT/progress.txt; exit $rc; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/progress.txt
echo "[2] diagonal block: primitives, phases, the single-launch evaluation" | tee -a $OUT/progress.txt
timeout -k 5 60 ./tools/leaf_ubench > $OUT/r04_leaf_ubench.txt 2>&1
timeout -k 5 120 ./tools/leaf_bench > $OUT/r04_leaf_bench.txt 2>&1
head -1 $OUT/r04_leaf_bench.txt | tee -a $OUT/progress.txt
(timeout -k 5 60 ./tools/small_eval_bench 128 8; timeout -k 5 60 ./tools/small_eval_bench 100 2; timeout -k 5 60 ./tools/small_eval_bench 128 32) > $OUT/r04_small_eval_bench.txt 2>&1
echo "[3] fit rates at small n (persistent fit kernel up to 128 rows), host-driven for comparison" | tee -a $OUT/progress.txt
(timeout -k 10 300 python3 tools/small_fit_rate.py 64 100 128 200 256 512 768 1024 1536 2048; echo "--- HBEGP_SMALL_FIT=0 (host-driven optimiser, single-launch evaluation)"; HBEGP_SMALL_FIT=0 timeout -k 10 120 python3 tools/small_fit_rate.py 64 100 128; echo "--- HBEGP_SMALL=0 (general five-launch path)"; HBEGP_SMALL=0 timeout -k 10 120 python3 tools/small_fit_rate.py 64 100 128) 2>&1 | grep -v amdgpu.ids > $OUT/r04_small_fit_rates.txt
cat $OUT/r04_small_fit_rates.txt | tee -a $OUT/progress.txt
echo "[4] fit_bits: 16 identical fixed-work fits, task queue (config M), launch path (n=2048), persistent fit kernel (n=128)" | tee -a $OUT/progress.txt
timeout -k 10 300 python3 tools/fit_bits.py 16 2>&1 | grep -v amdgpu.ids | tee $OUT/fit_bits_dag.txt | tee -a $OUT/progress.txt
HBEGP_DAG=0 timeout -k 10 300 python3 tools/fit_bits.py 16 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/fit_bits_launch.txt | tee -a $OUT/progress.txt
timeout -k 10 300 python3 tools/fit_bits.py 32 128 2>&1 | grep -v amdgpu.ids | tee $OUT/fit_bits_small.txt | tee -a $OUT/progress.txt
echo "[5] bench --workload C3" | tee -a $OUT/progress.txt
timeout -k 10 600 python3 bench.py --workload C3 --steps 2 --warmup 1 > $OUT/r04_bench_c3.json 2> $OUT/bench_c3.err || { tail -5 $OUT/bench_c3.err | tee -a $OUT/progress.txt; }
cut -c1-700 $OUT/r04_bench_c3.json | tee -a $OUT/progress.txt
echo "[6] PMC traffic of the task-queue launch, three slots (rocprofv3 --pmc serialises dispatches: one 96-workgroup launch alone)" | tee -a $OUT/progress.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p_fetch -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p_write -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_write.log 2>&1 && \
python3 $ROOT/tools/pmc_summarize.py $OUT/p_fetch/run_counter_collection.csv $OUT/p_write/run_counter_collection.csv $OUT/r04_pmc_3slot.json "python3 tools/profile_3slot.py 6 (three slots evaluating at once; --pmc serialises the dispatches)" > $OUT/pmc3.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; rm -rf $OUT/p_fetch $OUT/p_write
cd $ROOT
echo "[7] kernel trace of a small fit (persistent fit kernel)" | tee -a $OUT/progress.txt
cd /tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smallfit -o run -- python3 $ROOT/tools/fit_timing_probe.py 128 > $OUT/smallfit.log 2>&1
cp $OUT/smallfit/run_kernel_stats.csv $OUT/r04_small_fit_kernel_stats.csv 2>/dev/null; rm -rf $OUT/smallfit
cd $ROOT
echo "[7b] where an evaluation's time outside the task-queue launch goes (three-run fit, config M and n=256)" | tee -a $OUT/progress.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/gapM -o run -- python3 $ROOT/tools/fit_rate.py 3 > $OUT/gapM.log 2>&1 && python3 $ROOT/tools/tail_gaps.py $OUT/gapM/run_kernel_trace.csv > $OUT/r04_tail_gaps_after_M.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/gap256 -o run -- python3 $ROOT/tools/fit_rate.py 6 256 > $OUT/gap256.log 2>&1 && python3 $ROOT/tools/tail_gaps.py $OUT/gap256/run_kernel_trace.csv > $OUT/r04_tail_gaps_after_256.txt 2>&1
rm -rf $OUT/gapM $OUT/gap256
cd $ROOT
tail -n 3 $OUT/r04_tail_gaps_after_M.txt | cut -c1-300 | tee -a $OUT/progress.txt
echo "[7c] task traces of one evaluation: n=4096 (divide-and-conquer inverse) and n=2048 (row-progressive plan)" | tee -a $OUT/progress.txt
for n in 4096 2048; do HBEGP_DAG_TRACE=$OUT/trace_$n.txt timeout -k 10 200 python3 tools/trace_eval.py $n 2>&1 | grep -v amdgpu.ids > $OUT/r04_task_trace_$n.txt; rm -f $OUT/trace_$n.txt; done
grep -E "makespan|leaf chain" $OUT/r04_task_trace_4096.txt $OUT/r04_task_trace_2048.txt | cut -c1-200 | tee -a $OUT/progress.txt
echo "[7d] f32 rates" | tee -a $OUT/progress.txt
timeout -k 10 200 python3 tools/f32_fit_rate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r04_f32_rates.txt | tee -a $OUT/progress.txt
echo "[8] profiles" | tee -a $OUT/progress.txt
cp $OUT/r04_pmc_3slot.json profiles/r04_pmc_3slot.json 2>/dev/null
bash tools/refresh_profiles.sh r04 > $OUT/refresh.log 2>&1
tail -30 $OUT/refresh.log | cut -c1-400 | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
