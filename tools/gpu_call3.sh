#!/bin/bash
# round 3, call 3: causal test of the helper-wave race (library without the fix must fail, shipped one must pass), the GPU
# suite, the contention probe and PMC passes over three concurrent slots.
OUT=$PWD/gpurun_out/call3
mkdir -p $OUT
ROOT=$PWD
echo "[1] helper-delay test, library WITHOUT the pivot-row copy (expected: FAIL)" | tee $OUT/progress.txt
HBEGP_LIB=build/var/libhbegp_nocopy.so timeout -k 10 300 python3 -m pytest tests/test_gpu_dag.py -q -k helper_waves -p no:cacheprovider > $OUT/delay_nocopy.txt 2>&1
tail -5 $OUT/delay_nocopy.txt | tee -a $OUT/progress.txt
echo "[2] GPU suite, shipped library" | tee -a $OUT/progress.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -5 $OUT/gputest.txt | tee -a $OUT/progress.txt
if [ $rc -ne 0 ]; then echo "GPU SUITE rc=$rc: stopping" | tee -a $OUT/progress.txt; exit $rc; fi
echo "[3] contention probe" | tee -a $OUT/progress.txt
timeout -k 10 300 python3 tools/contention_probe.py 4096 > $OUT/contention.json 2> $OUT/contention.err || exit 1
cat $OUT/contention.json | tee -a $OUT/progress.txt
echo "[4] PMC passes over three concurrent slots" | tee -a $OUT/progress.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_trace -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_trace.log 2>&1 || exit 1
echo trace done | tee -a $OUT/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p_fetch -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_fetch.log 2>&1 || exit 1
echo fetch done | tee -a $OUT/progress.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p_write -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_write.log 2>&1 || exit 1
echo write done | tee -a $OUT/progress.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p_sq -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_sq.log 2>&1 || exit 1
echo sq done | tee -a $OUT/progress.txt
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p_tcc -o run -- python3 $ROOT/tools/profile_3slot.py 6 > $OUT/p_tcc.log 2>&1 || echo "tcc pass failed" | tee -a $OUT/progress.txt
# keep the merge small: drop everything but the csv files we read
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT | tee -a $OUT/progress.txt
echo done | tee -a $OUT/progress.txt
