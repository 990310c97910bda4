#!/bin/bash
# usage: tools/sweep_fit_n.sh "n1 n2 ..." "ENV=.. ENV=.." "ENV=.." ...   fit rates (tools/fit_rate.py) per size and configuration
sizes=$1; shift
for n in $sizes; do
  for cfg in "$@"; do
    echo "== n=$n $cfg"; env $cfg python3 tools/fit_rate.py 3 $n 2>/dev/null
  done
done
