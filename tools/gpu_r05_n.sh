#!/bin/bash
# where a round of 16 concurrent small fits spends its time
OUT=gpurun_out/r05_n
mkdir -p $OUT
for k in 1 16; do
  FIT_PHASES=1 HBEGP_TIMING=1 GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python3 tools/concurrent_fits.py 128 $k > $OUT/k$k.out 2> $OUT/k$k.err
  grep "fits/s\|per fit" $OUT/k$k.out | tee -a $OUT/log.txt
  python3 - $OUT/k$k.err <<'PY' | tee -a $OUT/log.txt
import re, sys
rows = [tuple(map(float, m.groups())) for m in re.finditer(r"fit: problem ([\d.]+) ms, optimiser runs ([\d.]+) ms \(\d+ evaluations\), model ([\d.]+) ms", open(sys.argv[1]).read())]
rows = rows[len(rows) // 2:]
n = len(rows)
print(f"  inside new(), last {n} fits: problem {sum(r[0] for r in rows) / n:.2f} ms, runs {sum(r[1] for r in rows) / n:.2f} ms, model {sum(r[2] for r in rows) / n:.2f} ms")
PY
done
