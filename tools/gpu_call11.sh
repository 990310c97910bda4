#!/bin/bash
OUT=gpurun_out/call11
mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_dag.py -q -k full_mode -p no:cacheprovider 2>&1 | tail -15
echo "--- fit phases (HBEGP_TIMING=1)"
HBEGP_TIMING=1 timeout -k 10 200 python3 tools/fit_rate.py 4 2>&1 | grep -v amdgpu.ids | tail -8
echo "--- graphs off"
HBEGP_NO_GRAPH=1 HBEGP_TIMING=1 timeout -k 10 200 python3 tools/fit_rate.py 4 2>&1 | grep -v amdgpu.ids | tail -6
echo "--- extend timing n=4096"
timeout -k 10 200 python3 - <<'PY'
import time, sys
sys.path.insert(0, '.')
from hbetune_rs_amd import gpr, synth
w = synth.make_workload("M")
for i in range(4):
    t0 = time.perf_counter(); fk = gpr.FittedKernel.extend(w["X"], w["y"], w["theta"]); dt = time.perf_counter() - t0
    print(f"extend n=4096 call {i}: {dt*1e3:.2f} ms"); fk.release()
PY
