#!/bin/bash
# Round 5, call D: the whole GPU suite with the referee-based bars, the pool-race test, the f32 kernel without spills; then the bench line.
OUT=$PWD/gpurun_out/r5d
mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -p no:cacheprovider -rP > $OUT/gputest.txt 2>&1
rc=$?
tail -5 $OUT/gputest.txt
grep -E "trace replay|fitted model|persistent fit" $OUT/gputest.txt | cut -c1-260
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/gputest.txt | head -30; exit $rc; fi
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
print(json.dumps(d["parity_in_run"], indent=1)[:3000])
PY
