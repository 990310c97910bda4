#!/usr/bin/env python3
"""bench.py — GP fit+predict throughput of the MI355X engine on BASELINE.json's metric configuration.

One "step" = one fixed-work fit + one batched predict on config M (rosenbrock d=8, n=4096, f64):
  * fit: (1 + R) = 3 bounded L-BFGS runs x 150 log-marginal-likelihood+gradient evaluations (the reference's cap:
    gpr.rs:222 n_restarts_optimizer = 2, gradmin.rs:54 maxeval = 150), K^-1 of the captured best evaluation,
  * predict: mean + variance at m = 1600 candidates (about one generation of the caller's predict calls).
Inputs are synthetic (hbetune_rs_amd/synth.py), uploaded to HBM inside the C ABI call; the timed region covers the
whole fit+predict call chain (H2D of X, y is ~300 KB and included).

Multi-GPU: the path shards as independent units with no exchange step: every rank runs whole
fit+predict steps on its own GPU ("weak" scaling, no data-path collective); torch.distributed (RCCL) is only used
for the barrier and the max-over-ranks of the timings.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak (vendor); measured 77.1 with tools/ubench.hip
EVALS_PER_RUN = 150
N_RESTARTS = 2
M_CANDIDATES = 1600


def cpu_baseline(w, theta, Xs):
    """Reference-faithful CPU port (oracle/gpr_oracle.py: materialised dK tensor, LAPACK potrf/potrs/potri), 1 thread
    (the reference builds OpenBLAS with USE_THREAD=0, Makefile:3-4).  Bounded sample: ONE evaluation + one predict."""
    import numpy as np
    from threadpoolctl import threadpool_limits

    from oracle import gpr_oracle as O

    s2, c, ell = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:])
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        res = O.lml_with_gradient(w["X"], w["y"], s2, c, ell, 2.5)
        t_eval = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.predict(Xs, w["X"], res["alpha"], res["k_inv"], c, ell, 2.5)
        t_pred = time.perf_counter() - t0
    n_evals = (1 + N_RESTARTS) * EVALS_PER_RUN
    fit_predict_s = n_evals * t_eval + t_pred
    # best-effort variant beside it: the same port with every host core given to BLAS/LAPACK
    import os as _os

    t0 = time.perf_counter()
    O.lml_with_gradient(w["X"], w["y"], s2, c, ell, 2.5)
    t_eval_all = time.perf_counter() - t0
    # and the floor of any CPU implementation on this box: LAPACK dpotrf + dpotri alone (n^3 flops), all cores
    from scipy.linalg import lapack

    Kc = res["kernel_matrix"].copy()
    t0 = time.perf_counter()
    ch, _ = lapack.dpotrf(Kc, lower=1)
    lapack.dpotri(ch, lower=1)
    t_lapack_all = time.perf_counter() - t0
    return {
        "value": 1.0 / fit_predict_s,
        "unit": "fit+predict/s",
        "cores": 1,
        "kind": "port",
        "sample": f"1 lml+gradient evaluation ({t_eval:.2f} s, x{n_evals} per fit) + 1 predict m={len(Xs)} ({t_pred:.2f} s), "
                  f"n={w['n']} d={w['d']} f64, numpy + LAPACK dpotrf/dpotrs/dpotri, 1 thread",
        "eval_s": t_eval,
        "predict_s": t_pred,
        "best_effort": {"cores": _os.cpu_count(), "eval_s": t_eval_all,
                        "value": 1.0 / (n_evals * t_eval_all + t_pred), "note": "same port, all host cores for BLAS/LAPACK",
                        "lapack_only_eval_s": t_lapack_all,
                        "lapack_only_value": 1.0 / (n_evals * t_lapack_all + t_pred),
                        "lapack_note": "dpotrf + dpotri alone on all cores: a floor for any CPU implementation of one evaluation"},
    }, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=None, help="override n (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np  # noqa: F401
    import torch  # noqa: F401  (loaded before libhbegp.so so the process holds one HIP runtime)

    from hbetune_rs_amd import dist as D
    from hbetune_rs_amd import gpr, synth

    rank, local_rank, world = D.rank_info()
    if args.gpus != world:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
    dist = D.init("nccl") if world > 1 else None

    w = synth.make_workload("M", n=args.n)
    X, y, theta = w["X"], w["y"], w["theta"]
    n, d = X.shape
    starts = synth.restart_points("M", w["lo"], w["hi"], N_RESTARTS)
    Xs = synth.candidates("M", M_CANDIDATES, d)

    ctx = gpr.Context(device_ids=[local_rank])

    def step():
        fk = gpr.FittedKernel.new(X, y, w["theta0"], w["lo"], w["hi"], starts, nu=2.5, ctx=ctx, maxeval=EVALS_PER_RUN,
                                  fixed_work=True)
        mean, var, _ = fk.predict(Xs)
        fk.release()
        return mean, var

    for _ in range(args.warmup):
        step()
    D.barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    D.barrier(dist)
    elapsed = D.max_over_ranks(dist, time.perf_counter() - t0)

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed  # whole-job fit+predict per second
        # roofline of the dominant kernel (gemm_kernel<double,128>: Cholesky/TRTRI/LAUUM tile GEMMs), measured live with
        # hipEvents on the evaluation stream: algorithmic GFLOP of its launches in one evaluation / their summed duration.
        prob = gpr.Problem(X, y, nu=2.5, ctx=ctx)
        ph = prob.time_eval(theta, reps=5)
        big = "gemm128" if ph["gemm128_ms"] > 0 else "gemm64"
        achieved = ph[f"{big}_gflop"] / ph[f"{big}_ms"] if ph[f"{big}_ms"] > 0 else 0.0  # GFLOP/ms = TFLOP/s
        eval_tflops = (n ** 3) * 1e-12 / (ph["eval_graph_ms"] * 1e-3)
        # HBM/L2-miss bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate runs, gfx950 read correction applied) -- counters cannot be collected from inside this process
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            k = pmc.get(f"hbegp::gemm_kernel<double, {big[4:]}, 0>") or pmc[f"hbegp::gemm_kernel<double, {big[4:]}>"]
            if args.n is None:
                traffic = k["fetch_bytes_per_dispatch"] + k["write_bytes_per_dispatch"]
        except Exception:
            traffic = None
        roofline = {
            "bound": "mfma",
            "kernel": f"hbegp::gemm_kernel<double, {big[4:]}, 0>",
            "achieved": achieved,
            "peak": PEAK_FP64_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (mean over the kernel's launches in one evaluation; profiles/r01_pmc_traffic.json)",
            "launches_per_eval": ph[f"n_{big}"],
            "gflop_per_launch": ph[f"{big}_gflop"] / max(ph[f"n_{big}"], 1.0),
            "avg_launch_ms": ph[f"{big}_ms"] / max(ph[f"n_{big}"], 1.0),
            "algorithmic_gflop_per_eval": ph[f"{big}_gflop"],
            "kernel_ms_per_eval": ph[f"{big}_ms"],
            "whole_eval_ms": ph["eval_graph_ms"],
            "whole_eval_frac_of_peak": eval_tflops / PEAK_FP64_MFMA_TFLOPS,
            "whole_fit_frac_of_peak": ((1 + N_RESTARTS) * EVALS_PER_RUN * n ** 3 + 2.0 * M_CANDIDATES * n * n) * 1e-12
                                      / (ms_per_step * 1e-3) / PEAK_FP64_MFMA_TFLOPS,
            "phases_ms": {k: round(v, 4) for k, v in ph.items()},
        }
        prob.close()
        out = {
            "metric": "GP fit+predict/sec (n=4096,d=8,f64)",
            "value": value,
            "unit": "fit+predict/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"M: rosenbrock d={d} n={n} f64, Matern nu=5/2 x constant + white noise; fixed-work fit = "
                            f"{1 + N_RESTARTS} L-BFGS runs x {EVALS_PER_RUN} lml+gradient evaluations + K^-1, then predict "
                            f"mean+variance at m={M_CANDIDATES}",
                "evals_per_fit": (1 + N_RESTARTS) * EVALS_PER_RUN,
                "m_candidates": M_CANDIDATES,
                "parallelism": f"replicas x{world} (independent fits per GPU, no collective)",
            },
            "roofline": roofline,
        }
        # the same fit with the optimiser's own stopping rule (not timed above): evaluations it actually needs
        t0 = time.perf_counter()
        fk = gpr.FittedKernel.new(X, y, w["theta0"], w["lo"], w["hi"], starts, nu=2.5, ctx=ctx, maxeval=EVALS_PER_RUN, trace=True)
        t_conv = time.perf_counter() - t0
        out["converged_fit"] = {"seconds": t_conv, "evaluations": int(len(fk.trace["lml"])), "lml": fk.lml,
                                "note": "early-stopping fit (projected-gradient / progress tolerances of csrc/lbfgsb.hpp), same data and starts"}
        fk.release()
        if world == 1 and not args.no_cpu_baseline:
            cb, ref = cpu_baseline(w, theta, Xs)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_port"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        D.barrier(dist)
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
