#!/bin/bash
# Round 5, call J: full GPU suite + smoke + the bench line (with the concurrent-fits side line and the referee).
OUT=$PWD/gpurun_out/r5j
mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/gputest.txt 2>&1
rc=$?
tail -3 $OUT/gputest.txt
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5j/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("concurrent_fits_side_line"), indent=1)[:1500])
print(json.dumps(d["parity_in_run"]["timed_model"]["referee"], indent=1)[:1200])
for k in ("f32_side_line", "small_n_side_line", "baseline_configs_side_line"):
    print(k, json.dumps(d.get(k))[:700])
PY
