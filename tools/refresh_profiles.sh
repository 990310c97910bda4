#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh r02
# Writes gpurun_out/profiles_<tag>/; copy the files into profiles/ afterwards.  rocprofv3: the program itself follows `--`
# (python3), counters are collected in passes of their own (never together with trace domains other than --kernel-trace).
set -e
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[1/7] kernel trace of bench.py (3 concurrent optimiser runs, task-queue launches)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-side-lines --steps 3 --warmup 1 > "$OUT/bench_trace.log" 2>&1
cp "$OUT/bench/run_kernel_stats.csv" "$OUT/${TAG}_bench_3stream_kernel_stats.csv"
python3 "$ROOT/tools/trace_timeline.py" "$OUT/bench/run_kernel_trace.csv" > "$OUT/${TAG}_bench_3stream_timeline.txt" 2>&1 || true
rm -rf "$OUT/bench"
echo "[2/7] kernel trace of single evaluations: task queue (256 workgroups) and launch-per-product path"
HBEGP_DAG=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/eval1" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/eval_dag.log" 2>&1
cp "$OUT/eval1/run_kernel_stats.csv" "$OUT/${TAG}_single_eval_dag_kernel_stats.csv"
HBEGP_DAG=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/eval0" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/eval_launches.log" 2>&1
cp "$OUT/eval0/run_kernel_stats.csv" "$OUT/${TAG}_single_eval_launches_kernel_stats.csv"
rm -rf "$OUT/eval1" "$OUT/eval0"
echo "[3/7] PMC FETCH_SIZE (task queue)"
HBEGP_DAG=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/fetch.log" 2>&1
echo "[4/7] PMC WRITE_SIZE (task queue)"
HBEGP_DAG=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/write.log" 2>&1
python3 "$ROOT/tools/pmc_summarize.py" "$OUT/fetch/run_counter_collection.csv" "$OUT/write/run_counter_collection.csv" "$OUT/${TAG}_pmc_traffic.json" > "$OUT/pmc_summary.log"
rm -rf "$OUT/fetch" "$OUT/write"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
echo "[5/7] PMC SQ counters, task queue"
HBEGP_DAG=1 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$OUT/sq1" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/sq_dag.log" 2>&1
echo "[6/7] PMC SQ counters, launch-per-product path (gemm_kernel 64- and 32-tile launches in context)"
HBEGP_DAG=0 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$OUT/sq0" -o run -- python3 "$ROOT/tools/profile_eval.py" M > "$OUT/sq_launches.log" 2>&1
python3 "$ROOT/tools/pmc_sq_summarize.py" "$OUT/${TAG}_pmc_sq.json" dag="$OUT/sq1/run_counter_collection.csv" launches="$OUT/sq0/run_counter_collection.csv" > "$OUT/pmc_sq_summary.log"
rm -rf "$OUT/sq1" "$OUT/sq0"
cd "$ROOT"
echo "[7/7] bench.py (full, with CPU baseline)"
cp "$OUT/${TAG}_pmc_traffic.json" "$ROOT/profiles/${TAG}_pmc_traffic.json"
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
cut -c1-300 "$OUT/${TAG}_bench.json"
cat "$OUT/pmc_sq_summary.log"
ls -la "$OUT"
echo "[extra] kernel trace of one f32 evaluation (C5: himmelblau n=2048, --use-32)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o run -- python3 "$ROOT/tools/profile_eval.py" C5 > "$OUT/eval_c5.log" 2>&1
cp "$OUT/c5/run_kernel_stats.csv" "$OUT/${TAG}_c5_f32_kernel_stats.csv"
rm -rf "$OUT/c5"
cd "$ROOT"
