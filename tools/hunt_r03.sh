#!/bin/bash
# Round-3 bisect of the run-to-run non-reproducibility (one gpurun call): launch path and called-leaf task queue, with and
# without the diagonal-row copy for the helper waves (LEAF_DIAG_COPY).  Output: gpurun_out/hunt/*.txt|json
set -e
OUT=gpurun_out/hunt
mkdir -p $OUT
run() {  # name, env..., -- args
  name=$1; shift
  echo "=== $name" | tee -a $OUT/progress.txt
  env "$@" > $OUT/$name.txt 2>&1 || echo "FAILED $name" | tee -a $OUT/progress.txt
  head -3 $OUT/$name.txt | tee -a $OUT/progress.txt
}
V=build/var
run A_launch_rand      HBEGP_DAG=0 HUNT_OUT=$OUT/A_launch_rand.json timeout -k 10 300 python3 tools/nondet_hunt.py 2048 3000 rand 3
run B_launch_rand_copy HBEGP_DAG=0 HBEGP_LIB=$V/libhbegp_diagcopy.so HUNT_OUT=$OUT/B_launch_rand_copy.json timeout -k 10 300 python3 tools/nondet_hunt.py 2048 3000 rand 3
run A_launch_slotconst HBEGP_DAG=0 HUNT_OUT=$OUT/A_launch_slotconst.json timeout -k 10 300 python3 tools/nondet_hunt.py 2048 3000 slotconst 2
run A_launch_walk      HBEGP_DAG=0 HUNT_OUT=$OUT/A_launch_walk.json timeout -k 10 300 python3 tools/nondet_hunt.py 2048 3000 walk 2
run C_dag_noinline      HBEGP_LIB=$V/libhbegp_noinline.so HUNT_OUT=$OUT/C_dag_noinline.json timeout -k 10 400 python3 tools/nondet_hunt.py 4096 1200 rand 2
run D_dag_noinline_copy HBEGP_LIB=$V/libhbegp_noinline_diagcopy.so HUNT_OUT=$OUT/D_dag_noinline_copy.json timeout -k 10 400 python3 tools/nondet_hunt.py 4096 1200 rand 2
run E_dag_default       HUNT_OUT=$OUT/E_dag_default.json timeout -k 10 400 python3 tools/nondet_hunt.py 4096 1200 rand 2
echo done | tee -a $OUT/progress.txt
