#!/bin/bash
# kernel statistics of the headline workload only (bench.py --no-side-lines) + the bench line of the same command
ROOT=$PWD
OUT=$ROOT/gpurun_out/call13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-side-lines --steps 3 --warmup 1 > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
cp "$OUT/bench/run_kernel_stats.csv" "$OUT/r03_bench_3stream_kernel_stats.csv"
python3 "$ROOT/tools/trace_timeline.py" "$OUT/bench/run_kernel_trace.csv" > "$OUT/r03_bench_3stream_timeline.txt" 2>&1
rm -rf "$OUT/bench"
head -4 "$OUT/r03_bench_3stream_kernel_stats.csv" | cut -c1-160
python3 -c "
import json; d=json.loads(open('$OUT/bench_trace.json').read().strip().splitlines()[-1]); r=d['roofline']; print('under rocprofv3: value', d['value'], 'avg_launch_ms', r['avg_launch_ms'], 'frac', r['frac'])"
cat "$OUT/r03_bench_3stream_timeline.txt" | tail -8
