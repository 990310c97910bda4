import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hbetune_rs_amd import gpr, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = synth.make_workload("M", n=n)
st = synth.restart_points("M", w["lo"], w["hi"], 2)
for i in range(4):
    t0 = time.perf_counter()
    f = gpr.FittedKernel.new(w["X"], w["y"], w["theta0"], w["lo"], w["hi"], st, nu=2.5, maxeval=150, fixed_work=True)
    dt = time.perf_counter() - t0
    print(f"fit {i}: {dt*1e3:.2f} ms, evals {f.n_evals}, lml {f.lml:.6f}")
    f.release()
