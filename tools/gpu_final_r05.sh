#!/bin/bash
# Round-5 closing run: GPU suite + smoke, reproducibility (alone, mixed with other fits, side by side), small-n rates, C3 bench line,
# PMC traffic of the task-queue launch, kernel traces, task traces, f32 rates, the profiles under profiles/r05_*.
OUT=$PWD/gpurun_out/final_r05
mkdir -p $OUT
ROOT=$PWD
echo "[1] GPU suite + smoke" | tee $OUT/progress.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -p no:cacheprovider -rP > $OUT/gputest.txt 2>&1
rc=$?
tail -2 $OUT/gputest.txt | tee -a $OUT/progress.txt
grep -E "trace replay|fitted model|persistent fit|device optimiser" $OUT/gputest.txt | cut -c1-300 > $OUT/r05_parity_rules_summary.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30 | tee -a $OUT/progress.txt; exit $rc; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/progress.txt
echo "[2] reproducibility: 16 identical fits (task queue, launch path, persistent kernel), 12 with other fits in between" | tee -a $OUT/progress.txt
(timeout -k 10 300 python3 tools/fit_bits.py 16; HBEGP_DAG=0 timeout -k 10 300 python3 tools/fit_bits.py 16 2048; timeout -k 10 300 python3 tools/fit_bits.py 32 128; FIT_BITS_MIX=1 timeout -k 10 400 python3 tools/fit_bits.py 12) 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_fit_bits.txt | tee -a $OUT/progress.txt
echo "[3] fits side by side on one context (GPU_MAX_HW_QUEUES = 4 / 16), every fit compared bit for bit with the solo fit" | tee -a $OUT/progress.txt
(for q in 4 16; do for spec in "128 1 4 16" "900 1 4" "1024 1 2 4" "2048 1 2 4"; do GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 tools/concurrent_fits.py $spec 2>&1 | grep "^n=" | sed "s/^/GPU_MAX_HW_QUEUES=$q /"; done; done; FIT_F32=1 GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python3 tools/concurrent_fits.py 1500 1 4 2>&1 | grep "^n=" | sed "s/^/GPU_MAX_HW_QUEUES=16 /") | tee $OUT/r05_concurrent_fits.txt | tee -a $OUT/progress.txt
echo "[4] fit rates at small n" | tee -a $OUT/progress.txt
timeout -k 10 300 python3 tools/small_fit_rate.py 64 100 128 200 256 512 768 1024 1536 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_small_fit_rates.txt | tee -a $OUT/progress.txt
echo "[5] bench --workload C3" | tee -a $OUT/progress.txt
timeout -k 10 600 python3 bench.py --workload C3 --steps 2 --warmup 1 > $OUT/r05_bench_c3.json 2> $OUT/bench_c3.err || { tail -5 $OUT/bench_c3.err | tee -a $OUT/progress.txt; }
cut -c1-400 $OUT/r05_bench_c3.json | tee -a $OUT/progress.txt
echo "[6] PMC traffic of the task-queue launch, three slots (rocprofv3 --pmc serialises dispatches: one 96-workgroup launch alone)" | tee -a $OUT/progress.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p_fetch -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p_write -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_write.log 2>&1 && \
python3 $ROOT/tools/pmc_summarize.py $OUT/p_fetch/run_counter_collection.csv $OUT/p_write/run_counter_collection.csv $OUT/r05_pmc_3slot.json "python3 tools/profile_3slot.py 6 (three slots evaluating at once; --pmc serialises the dispatches)" > $OUT/pmc3.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; rm -rf $OUT/p_fetch $OUT/p_write
cd $ROOT
echo "[7] where an evaluation's time outside the task-queue launch goes (three-run fit, config M)" | tee -a $OUT/progress.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gapM -o run -- python3 $ROOT/tools/fit_rate.py 3 > $OUT/gapM.log 2>&1 && python3 $ROOT/tools/tail_gaps.py $OUT/gapM/run_kernel_trace.csv > $OUT/r05_tail_gaps_M.txt 2>&1
python3 $ROOT/tools/kstats.py $OUT/gapM/run_kernel_stats.csv > $OUT/r05_fit_kernel_stats.txt 2>&1
rm -rf $OUT/gapM
cd $ROOT
tail -n 3 $OUT/r05_tail_gaps_M.txt | cut -c1-300 | tee -a $OUT/progress.txt
cat $OUT/r05_fit_kernel_stats.txt | tee -a $OUT/progress.txt
echo "[8] task traces: one evaluation n=4096 alone, one launch of 96 workgroups" | tee -a $OUT/progress.txt
HBEGP_DAG_TRACE=$OUT/trace_4096.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep -v amdgpu.ids > $OUT/r05_task_trace_4096.txt; rm -f $OUT/trace_4096.txt
HBEGP_DAG_WG=96 HBEGP_DAG_BIG128=2 HBEGP_DAG_TRACE=$OUT/trace_96.txt timeout -k 10 200 python3 tools/trace_eval.py 4096 2>&1 | grep -v amdgpu.ids > $OUT/r05_task_trace_4096_96wg.txt; rm -f $OUT/trace_96.txt
grep -E "makespan|leaf chain|tile efficiency|efficiency" $OUT/r05_task_trace_4096.txt $OUT/r05_task_trace_4096_96wg.txt | cut -c1-220 | tee -a $OUT/progress.txt
echo "[9] f32 rates" | tee -a $OUT/progress.txt
timeout -k 10 200 python3 tools/f32_fit_rate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_f32_rates.txt | tee -a $OUT/progress.txt
echo "[10] tile function alone (stage-loop forms, ablation)" | tee -a $OUT/progress.txt
(timeout -k 5 120 ./tools/tile_ubench 2048 8 7 256; TILE_UBENCH_ABL=1 timeout -k 5 200 ./tools/tile_ubench 2048 8 7 256) 2>&1 | grep "depth\|DIFF" > $OUT/r05_tile_ubench_final.txt
tail -4 $OUT/r05_tile_ubench_final.txt | cut -c1-200 | tee -a $OUT/progress.txt
echo "[11] profiles" | tee -a $OUT/progress.txt
cp $OUT/r05_pmc_3slot.json profiles/r05_pmc_3slot.json 2>/dev/null
bash tools/refresh_profiles.sh r05 > $OUT/refresh.log 2>&1
tail -30 $OUT/refresh.log | cut -c1-400 | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
