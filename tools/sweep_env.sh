#!/bin/bash
# usage: tools/sweep_env.sh OUTFILE "ENV1=.. ENV2=.." "ENV=.." ...   (each argument = one bench configuration)
out=$1; shift
: > "$out"
for cfg in "$@"; do
  echo "== $cfg" >> "$out"
  env $cfg python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>>"$out.err" | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        j = json.loads(l); r = j.get('roofline', {})
        print(j['value'], j['ms_per_step'], r.get('whole_eval_ms'), r.get('phases_ms'))
" >> "$out"
done
