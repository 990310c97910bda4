#!/usr/bin/env python3
"""Prints the engine's kernels from a rocprofv3 kernel-stats CSV: calls, average and minimum duration in microseconds."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "hbegp::" in n:
        print(f'{n.replace("void hbegp::", "")[:52]:52s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1000:9.1f} us  min {float(r["MinNs"]) / 1000:9.1f}')
