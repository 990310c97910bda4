#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh r01
# Writes gpurun_out/profiles_<tag>/; copy the files into profiles/ afterwards.
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[1/5] kernel trace of bench.py (3 concurrent optimiser runs)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/bench_trace.log" 2>&1
cp "$OUT/bench/run_kernel_stats.csv" "$OUT/${TAG}_bench_3stream_kernel_stats.csv"
python3 "$ROOT/tools/trace_timeline.py" "$OUT/bench/run_kernel_trace.csv" > "$OUT/${TAG}_bench_3stream_timeline.txt" 2>&1 || true
rm -rf "$OUT/bench"
echo "[2/5] kernel trace of single evaluations"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/eval" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/eval_trace.log" 2>&1
cp "$OUT/eval/run_kernel_stats.csv" "$OUT/${TAG}_single_eval_kernel_stats.csv"
rm -rf "$OUT/eval"
echo "[3/5] PMC FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/fetch.log" 2>&1
echo "[4/5] PMC WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/write.log" 2>&1
python3 "$ROOT/tools/pmc_summarize.py" "$OUT/fetch/run_counter_collection.csv" "$OUT/write/run_counter_collection.csv" "$OUT/${TAG}_pmc_traffic.json" > "$OUT/pmc_summary.log"
rm -rf "$OUT/fetch" "$OUT/write"
cd "$ROOT"
echo "[5/5] bench.py (full, with CPU baseline)"
cp "$OUT/${TAG}_pmc_traffic.json" "$ROOT/profiles/${TAG}_pmc_traffic.json"
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
cut -c1-300 "$OUT/${TAG}_bench.json"
ls -la "$OUT"
