#!/usr/bin/env python3
"""Three slots evaluating at once (the fit's configuration) for rocprofv3 passes: python3 tools/profile_3slot.py [reps=6]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hbetune_rs_amd import gpr, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
w = synth.make_workload("M")
prob = gpr.Problem(w["X"], w["y"], n_slots=3)
r = prob.time_concurrent(w["theta"], reps=reps)
print(r)
prob.close()
