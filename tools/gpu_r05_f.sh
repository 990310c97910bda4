#!/bin/bash
# Round 5, call F: tail kernels, one workgroup per tile again, occupancy capped through the LDS request; concurrent fits as they are.
OUT=$PWD/gpurun_out/r5f
ROOT=$PWD
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fit.py -x -q -p no:cacheprovider -rP > $OUT/gputest.txt 2>&1
rc=$?
tail -3 $OUT/gputest.txt; grep "device optimiser vs host" $OUT/gputest.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
for wgs in 3 4 5 8; do
  HBEGP_TILE_WGS_PER_CU=$wgs HBEGP_DAG=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval_$wgs -o run -- python3 $ROOT/tools/profile_eval.py M > $OUT/eval_$wgs.log 2>&1
  cp $OUT/eval_$wgs/run_kernel_stats.csv $OUT/single_eval_kernel_stats_wgs$wgs.csv; rm -rf $OUT/eval_$wgs
  echo "--- alone, at most $wgs workgroups per CU"; python3 $ROOT/tools/kstats.py $OUT/single_eval_kernel_stats_wgs$wgs.csv
done
cd $ROOT
for wgs in 3 4 8; do
  r=$(HBEGP_TILE_WGS_PER_CU=$wgs timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s)
  echo "fit, at most $wgs per CU: $r"
done
echo "--- concurrent fits, as the code stands"
timeout -k 10 200 python3 tools/concurrent_fits.py 128 1 4 16 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/concurrent_fits.py 1024 1 2 4 2>&1 | grep -v amdgpu.ids
