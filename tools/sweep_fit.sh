#!/bin/bash
# usage: tools/sweep_fit.sh OUTFILE "ENV1=.. ENV2=.." "ENV=.." ...   (each argument = one configuration of tools/fit_rate.py)
out=$1; shift
: > "$out"
for cfg in "$@"; do
  echo "== $cfg" >> "$out"
  env $cfg timeout -k 10 120 python3 tools/fit_rate.py 3 >> "$out" 2>> "$out.err" || echo "FAILED rc=$?" >> "$out"
done
cat "$out"
