#!/bin/bash
# alpha / lml reductions inside the X^T w launch (HBEGP_FUSE_ALPHA=0: alpha_reduce_kernel in a launch of its own, as before)
OUT=gpurun_out/r05_w
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/tests.txt
for round in 1 2; do
  for f in 1 0; do
    for n in 256 512 1024; do
      r=$(HBEGP_FUSE_ALPHA=$f timeout -k 10 120 python3 tools/fit_rate.py 8 $n 2>&1 | grep fits/s); echo "round $round fused=$f n=$n: $r" | tee -a $OUT/log.txt
    done
    r=$(HBEGP_FUSE_ALPHA=$f timeout -k 10 120 python3 tools/fit_rate.py 4 2>&1 | grep fits/s); echo "round $round fused=$f M: $r" | tee -a $OUT/log.txt
  done
done
