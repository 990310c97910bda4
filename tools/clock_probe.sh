#!/bin/bash
# Samples the GPU's shader clock and power while three-run fits of config M run (is the fit bound by the power limit?)
OUT=${1:-gpurun_out/clock}
mkdir -p $OUT
(rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -iE "sclk|power|Temperature|edge|junction" | head -12) > $OUT/idle.txt
python3 tools/fit_rate.py 12 > $OUT/fit.txt 2>&1 &
FIT=$!
sleep 6
for i in 1 2 3 4 5 6; do
  (rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|Power" | head -4) >> $OUT/busy.txt
  sleep 0.7
done
wait $FIT
(rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -3) >> $OUT/idle.txt
echo "--- idle"; cat $OUT/idle.txt; echo "--- during the fits"; cat $OUT/busy.txt; grep fits $OUT/fit.txt
