#!/bin/bash
# GPU suite + default bench line (the driver's round-end sequence, in small): gpurun_out/<tag>/
TAG=${1:-suite}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -4 $OUT/gputest.txt
if [ $rc -ne 0 ]; then echo "GPU SUITE rc=$rc"; grep -n "Error\|assert" $OUT/gputest.txt | head -20; exit $rc; fi
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 || { tail -5 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
timeout -k 10 900 python3 bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"frac",r["frac"],"whole_fit",r["whole_fit_frac_of_peak"],"per_launch",r["per_launch"])
print("f32",json.dumps(d.get("f32_side_line"))[:600])
print("small",json.dumps(d.get("small_n_side_line")))
print("cpu",json.dumps(d.get("cpu_baseline"))[:900])
PY
