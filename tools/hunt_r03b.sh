#!/bin/bash
# Round-3, second call: rate of deviating evaluations with and without the helper waves' pivot-row copy (LEAF_DIAG_COPY), in
# the called-leaf and inlined task-queue builds and on the launch path; then the fit rate of each build.
set -e
OUT=gpurun_out/hunt2
mkdir -p $OUT
run() {
  name=$1; shift
  echo "=== $name" | tee -a $OUT/progress.txt
  env "$@" > $OUT/$name.txt 2>&1 || echo "FAILED $name" | tee -a $OUT/progress.txt
  grep -v amdgpu.ids $OUT/$name.txt | head -4 | cut -c1-400 | tee -a $OUT/progress.txt
}
V=build/var
run C_dag_noinline      HBEGP_LIB=$V/libhbegp_noinline.so          HUNT_OUT=$OUT/C.json timeout -k 10 300 python3 tools/nondet_hunt.py 4096 12000 same 0
run D_dag_noinline_copy HBEGP_LIB=$V/libhbegp_noinline_diagcopy.so HUNT_OUT=$OUT/D.json timeout -k 10 300 python3 tools/nondet_hunt.py 4096 12000 same 0
run E_dag_inline        HBEGP_LIB=$V/libhbegp_nocopy.so            HUNT_OUT=$OUT/E.json timeout -k 10 300 python3 tools/nondet_hunt.py 4096 12000 same 0
run F_dag_inline_copy   HUNT_OUT=$OUT/F.json timeout -k 10 300 python3 tools/nondet_hunt.py 4096 12000 same 0
run A_launch            HBEGP_DAG=0 HBEGP_LIB=$V/libhbegp_nocopy.so HUNT_OUT=$OUT/A.json timeout -k 10 300 python3 tools/nondet_hunt.py 2048 20000 same 0
run B_launch_copy       HBEGP_DAG=0 HUNT_OUT=$OUT/B.json timeout -k 10 300 python3 tools/nondet_hunt.py 2048 20000 same 0
for lib in nocopy noinline_diagcopy; do
  run bench_$lib HBEGP_LIB=$V/libhbegp_$lib.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2
done
run bench_default timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2
echo done | tee -a $OUT/progress.txt
