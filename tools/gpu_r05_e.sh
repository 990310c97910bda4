#!/bin/bash
# Round 5, call E: the rewritten tail kernels -- parity first, then their times alone (kernel trace of single evaluations) and
# inside the three-run fit, for 4 / 6 / 8 queue-fed workgroups per CU.
OUT=$PWD/gpurun_out/r5e
ROOT=$PWD
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dag.py -x -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -3 $OUT/gputest.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
for wgs in 4 6 8; do
  HBEGP_TILE_WGS_PER_CU=$wgs HBEGP_DAG=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval_$wgs -o run -- python3 $ROOT/tools/profile_eval.py M > $OUT/eval_$wgs.log 2>&1
  echo "--- alone, $wgs workgroups per CU"; grep -E "kmat|gradtrace|trmv|alpha_reduce" $OUT/eval_$wgs/run_kernel_stats.csv | awk -F'","' '{printf "%-40.40s calls %s avg %.1f us\n", $1, $2, $4/1000}'
  cp $OUT/eval_$wgs/run_kernel_stats.csv $OUT/single_eval_kernel_stats_wgs$wgs.csv; rm -rf $OUT/eval_$wgs
done
cd $ROOT
for wgs in 4 6 8; do
  for fuse in 512 1000000; do
    r=$(HBEGP_TILE_WGS_PER_CU=$wgs HBEGP_FUSE_GRAD_MAX_TILES=$fuse timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s)
    echo "fit, $wgs per CU, fuse<=$fuse: $r"
  done
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit -o run -- python3 $ROOT/tools/fit_rate.py 3 > $OUT/fit.log 2>&1
echo "--- inside the three-run fit (default)"; grep -E "kmat|gradtrace|trmv|alpha_reduce|dag_kernel" $OUT/fit/run_kernel_stats.csv | awk -F'","' '{printf "%-40.40s calls %s avg %.1f us\n", $1, $2, $4/1000}'
cp $OUT/fit/run_kernel_stats.csv $OUT/fit_kernel_stats.csv; rm -rf $OUT/fit
