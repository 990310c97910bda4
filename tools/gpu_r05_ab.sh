#!/bin/bash
OUT=gpurun_out/r05_ab
mkdir -p $OUT
python3 tools/dump_workload.py 128 $OUT/wl128.bin
for k in 1 16; do
  HBEGP_TIMING=1 timeout -k 10 200 ./build/concurrent_fits_native $OUT/wl128.bin $k > $OUT/k$k.out 2> $OUT/k$k.err
  cat $OUT/k$k.out
  python3 - $OUT/k$k.err <<'PY'
import re, sys
rows = [tuple(map(float, m.groups())) for m in re.finditer(r"fit: problem ([\d.]+) ms, optimiser runs ([\d.]+) ms \(\d+ evaluations\), model ([\d.]+) ms", open(sys.argv[1]).read())]
rows = rows[len(rows) // 2:]
n = len(rows)
print(f"  inside hbegp_fit, last {n} fits: problem {sum(r[0] for r in rows) / n:.2f} ms, runs {sum(r[1] for r in rows) / n:.2f} ms, model {sum(r[2] for r in rows) / n:.2f} ms")
PY
done
