/*
 * hbegp.h — C ABI of the MI355X-native Gaussian-process surrogate engine (libhbegp.so).
 *
 * Drop-in boundary for hbetune's `src/gpr` hot path.  The reference has no C ABI of its own; its seam
 * is the pair of Rust traits `Estimator<A>` / `SurrogateModel<A>` (src/core/surrogate_model.rs:6-65),
 * implemented for the CPU by `EstimatorGPR` / `SurrogateModelGPR<A>` (src/core/gpr.rs:54-63, 215-338),
 * which call `FittedKernel::new/extend` (src/gpr/fit.rs:18-68) and `predict` (src/gpr/predict.rs:7-52).
 * Every entry point below replaces one of those calls; the Rust-side binding a maintainer would add is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - Plain C: host pointers and sizes only, no callbacks, no exceptions cross the boundary.
 *   - Element type A is f64 (default) or f32 (`--use-32`, src/bin/hbetune/main.rs:240-244); entry points
 *     that move arrays of A come in `_f64` / `_f32` pairs.  Hyper-parameters, bounds, lml and its
 *     gradient are always f64 (src/gpr/lml.rs:9-10, src/util/bounded_value.rs).
 *   - X is n x d row-major, features in [0,1] (src/core/space.rs:141-159); y has n entries and is
 *     already y-normalised by the caller (src/core/ynormalize.rs stays in the adapter).
 *   - Kernel = ConstantKernel(c) * Matern(nu, ell_1..ell_d) + white noise s2 (src/core/gpr.rs:51, 402-427),
 *     nu in {0.5, 1.5, 2.5} (src/gpr/matern_kernel.rs:65-80).  Extension: nu = +infinity selects the squared-exponential
 *     kernel exp(-r^2/2) (the reference has none: matern_kernel.rs:79 is unimplemented! for other nu; oracle: sklearn RBF).
 *   - theta is log-space, p = d + 2 entries ordered [ln s2, ln c, ln ell_1 .. ln ell_d]
 *     (src/gpr/fit.rs:140-144; gradient order src/gpr/lml.rs:67-68).  `lo`/`hi` are the linear-space
 *     bounds in the same order.  Kernel parameters are clamped into their bounds after exp()
 *     (`with_clamped_theta`, src/gpr/fit.rs:95); the noise is not (src/gpr/fit.rs:96).
 *   - Return value: one of HBEGP_* below.  Negative = usage/runtime error (see hbegp_last_error()).
 *   - All compute runs in hand-written HIP kernels on gfx950; there is no CPU fallback.  Without a GPU
 *     hbegp_ctx_create() fails with HBEGP_ENODEV.
 *   - Accuracy against the reference's CPU arithmetic (ndarray + LAPACK potrf/potrs/potri), relative to max(1, scale), the
 *     predictive variance relative to the amplitude: f64 1e-8 on lml, gradient, alpha, K^-1, mean and variance.  Where cond(K)
 *     puts LAPACK's own digits beyond that (the corners of the box an optimiser visits: cond(K) ~ 1e9 .. 1e12) the statement is
 *     made against the truth (an extended-precision referee, oracle/referee.c: kernel matrix in binary128, refined solves):
 *     |result - truth| <= max(1e-8, 2 |LAPACK - truth|) -- the engine is at most twice as far from the truth as the reference's
 *     own arithmetic.  Measured at the end of a fit of config M (cond(K) = 6.7e11): lml 7e-9 (LAPACK 1e-9), mean 1.9e-8 (2.5e-8),
 *     variance 2e-14 (LAPACK's K^-1 form: 3.9e-6).
 *     f32 (`--use-32`) 1e-4 on lml, gradient, mean and variance; alpha and K^-1 meet 1e-4 up to cond(K) ~ 7e4 on the paths a
 *     caller gets by default (the task queue's right-looking order from n = 641 on, for fits and single evaluations alike;
 *     the single launch up to n = 128), and max(1e-4, 2 x the deviation of LAPACK's own f32 path) otherwise -- between
 *     n = 129 and n = 640 f32 panel solves go through the explicit inverse of the whole left half, which costs K^-1 ~20 %
 *     more deviation at cond(K) = 7e4 (1.2e-4; LAPACK f32: 1.7e-4).  tests/test_gpu_fullsize.py holds both statements.
 *     Beyond cond(K) ~ 1e5 every f32 result is dominated by the rounding of the kernel matrix itself (tests/parity_rules.py).
 *   - Threads: the library is re-entrant per context.  Any number of host threads may call hbegp_fit_*, hbegp_extend_* and
 *     hbegp_predict_* on ONE context at the same time (replicas: independent fits side by side on one GPU); every fit owns its
 *     workspaces, streams and graphs, nothing on those paths uses the null stream or synchronises the whole device, and the
 *     launches are sized for the optimiser runs in flight over all fits of the process.  A concurrent fit returns the same
 *     bits as the same fit alone.  One hbegp_problem / one hbegp_model must not be used from two threads at once (a model
 *     serialises its own predictions).
 *   - Problems of at most 128 rows and 32 features (the reference's own regime, src/core/minimize.rs:118-120) take a path of
 *     their own: one evaluation is one launch that keeps K, L, L^-1, K^-1 and alpha in a compute unit's LDS, and one optimiser
 *     run of a fit is one persistent launch (evaluation + bounded L-BFGS step + capture on the device); the host only starts
 *     the runs side by side and collects.  Same entry points, same contracts; HBEGP_SMALL=0 / HBEGP_SMALL_FIT=0 select the
 *     general path / the host-driven optimiser for comparison.  Such fits issued by several threads at the same time share
 *     their launches (one grid carries the runs of all fits that arrive within a few milliseconds of each other; a thread
 *     alone never waits), and the host-side phases of such fits take turns while at most 16 threads are inside them (the HIP
 *     runtime's locks do worse): 16 native threads reach ~1,140 fits/s at n = 128 where one reaches 95, each fit bit for bit its
 *     solo result.
 */
#ifndef HBEGP_H
#define HBEGP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HBEGP_VERSION 200 /* 0.2.0: hbegp_fit_options starts with struct_size (the struct may grow at its end without breaking
                             callers built against an older header); hbegp_problem_debug_get which = 3, 4 */

enum {
  HBEGP_OK = 0,
  HBEGP_NOT_PD = 1,     /* Cholesky failed: reference returns None -> objective +inf, gradient 0 (lml.rs:47-50, fit.rs:105-112) */
  HBEGP_ALL_FAILED = 2, /* every evaluation of a fit failed: the reference would panic (fit.rs:161) */
  HBEGP_EINVAL = -1,
  HBEGP_EHIP = -2,      /* a HIP call failed */
  HBEGP_ENODEV = -3,    /* no usable gfx950 device */
  HBEGP_ENOMEM = -4
};

typedef struct hbegp_ctx hbegp_ctx;         /* devices, streams, workspace pools */
typedef struct hbegp_problem hbegp_problem; /* X, y resident on the device(s) + evaluation workspaces */
typedef struct hbegp_model hbegp_model;     /* fitted model: mirrors FittedKernel (fit.rs:6-12) + x_train */

/* ---- context ------------------------------------------------------------------------------------ */
int hbegp_version(void);
/* Number of visible gfx950 devices (0 when there is none); does not initialise a context. */
int hbegp_device_count(void);
/* n_devices > 0; device_ids may be NULL (= 0..n_devices-1).  One process may own several GPUs: optimiser
 * runs of one fit are sharded run r -> device r mod n_devices (restart axis, src/util/gradmin.rs:19-31). */
int hbegp_ctx_create(int n_devices, const int* device_ids, hbegp_ctx** out);
void hbegp_ctx_destroy(hbegp_ctx* ctx);
/* Thread-local text of the last error raised through any handle (never NULL). */
const char* hbegp_last_error(void);

/* ---- problem: upload X, y once, evaluate many theta (the L-BFGS inner loop, fit.rs:93-134) ------- */
/* n_slots >= 1 independent evaluation workspaces per device (one optimiser run uses one slot). */
int hbegp_problem_create_f64(hbegp_ctx* ctx, const double* X, const double* y, int n, int d, double nu,
                             int n_slots, hbegp_problem** out);
int hbegp_problem_create_f32(hbegp_ctx* ctx, const float* X, const float* y, int n, int d, double nu,
                             int n_slots, hbegp_problem** out);
void hbegp_problem_destroy(hbegp_problem* prob);

/* One evaluation of lml_with_gradient (src/gpr/lml.rs:29-79) on slot `slot` of device index `dev`.
 * theta/lo/hi as above (lo/hi may be NULL = no clamping).  grad may be NULL (skips the gradient pass).
 * Returns HBEGP_OK, or HBEGP_NOT_PD (then *lml = -inf and grad = 0). */
int hbegp_problem_eval(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo,
                       const double* hi, double* lml, double* grad);

/* Copy out device results of the most recent successful evaluation on (dev, slot); any pointer may be NULL.
 * alpha[n]; kinv[n*n] full symmetric row-major (what invc() returns, lml.rs:62); ldiag[n] = diag(L). */
int hbegp_problem_get_f64(hbegp_problem* prob, int dev, int slot, double* alpha, double* kinv, double* ldiag);
int hbegp_problem_get_f32(hbegp_problem* prob, int dev, int slot, float* alpha, float* kinv, float* ldiag);

/* Kernel-matrix assembly only: K = c*Matern(X,X) + s2*I, full symmetric n*n row-major (lml.rs:40-44). */
int hbegp_problem_kmat_f64(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo,
                           const double* hi, double* K);
int hbegp_problem_kmat_f32(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo,
                           const double* hi, float* K);

/* Timed evaluations for bench.py: runs `reps` evaluations at theta on (dev, slot) bracketed by hipEvents on the
 * slot's stream.  phase_ms (may be NULL) receives per-phase averages measured with events in eager mode:
 * [0] kmat, [1] chol+trtri GEMM launches (sum), [2] leaf (diag-block) launches (sum), [3] lauum GEMM,
 * [4] alpha/lml reductions, [5] gradtrace, [6] whole evaluation (graph replay), [7] number of GEMM launches/eval,
 * [8..13] (ms, algorithmic GFLOP) of the 128-, 64- and 32-tile GEMM launches, [14] eager evaluation, [15] leaf launches,
 * [16..18] launches per evaluation of the 128-, 64-, 32-tile GEMM, [19] the factorisation's task-queue launch (ms),
 * [20] its algorithmic GFLOP.  phase_ms must have room for 24 doubles. */
int hbegp_problem_time_eval(hbegp_problem* prob, int dev, int slot, const double* theta, int reps,
                            double* phase_ms);

/* Timed evaluations in the configuration a fit runs in: `reps` evaluations at theta on EVERY slot of device index `dev` at
 * once, one host thread per slot.  out must have room for 16 doubles:
 * [0] wall ms per round (every slot finishes one evaluation) with graph replay, [1] number of slots,
 * [2] the same with eager launches and one hipEvent pair per launch group, on each slot's stream; per evaluation, mean over
 * slots and repetitions, from those events: [3] kmat ms, [4] factorisation ms (task-queue launch, or diagonal-block +
 * tile-GEMM launches), [5] its algorithmic GFLOP, [6] lauum ms, [7] its GFLOP, [8] alpha/lml ms, [9] gradtrace ms,
 * [10] workgroups of the task-queue launch (0: launch-per-product path), [11] launches of [4] per evaluation. */
int hbegp_problem_time_concurrent(hbegp_problem* prob, int dev, const double* theta, int reps, double* out);

/* ---- fit / extend: mirrors FittedKernel::new / ::extend (fit.rs:18-68, 71-176) --------------------- */
typedef struct hbegp_fit_options {
  /* sizeof(hbegp_fit_options) in the header the CALLER was compiled against (use HBEGP_FIT_OPTIONS_INIT).  The library reads
   * and writes through min(struct_size, its own sizeof) bytes only: members beyond the caller's struct are taken as zero /
   * NULL, so the struct can grow at its end.  0 or a size that does not cover `maxeval` is HBEGP_EINVAL. */
  size_t struct_size;
  int maxeval;      /* evaluations per optimiser run; the reference uses 150 (gradmin.rs:54) */
  int fixed_work;   /* 0: stop a run when the optimiser converges; 1: keep evaluating up to maxeval (bench) */
  int lbfgs_memory; /* history pairs, 0 = default (10) */
  int trace_cap;    /* capacity (in evaluations) of the trace buffers below, 0 = no trace */
  /* optional trace of the evaluations, for replay parity against the oracle: trace_theta[trace_cap*p],
   * trace_lml[trace_cap] (-inf when not PD), trace_grad[trace_cap*p], trace_run[trace_cap].  Concurrent runs record as
   * their evaluations complete (the first trace_cap of them are kept); on return the records are sorted by run, and
   * within a run they are in evaluation order (problems of at most 128 rows: the first trace_cap records in that order).  Sorting needs trace_run; without it the order is completion order. */
  double* trace_theta;
  double* trace_lml;
  double* trace_grad;
  int* trace_run;
  int* trace_count; /* out: number of evaluations recorded */
  int* n_evals;     /* out (may be NULL): evaluations run, all optimiser runs together */
  int* n_not_pd;    /* out (may be NULL): evaluations whose kernel matrix was not positive definite (objective +inf, fit.rs:105-112) */
} hbegp_fit_options;
#define HBEGP_FIT_OPTIONS_INIT { sizeof(hbegp_fit_options) }

/* Maximise the log marginal likelihood over theta in [ln lo, ln hi] with 1 + n_restarts bounded L-BFGS runs:
 * run 0 starts at theta0, run r>0 at starts[(r-1)*p .. ] (uniform draws in log-bounds made by the caller's RNG,
 * gradmin.rs:22-24).  Capture rule = arg-max lml over every evaluation of every run (fit.rs:116-125), ties broken
 * by the lowest (run, eval) index.  On success *model owns X, y, alpha, K^-1, theta_best (clamped, fit.rs:155-164).
 * theta_best[p] and lml_best may be NULL.  Returns HBEGP_ALL_FAILED when no evaluation succeeded. */
int hbegp_fit_f64(hbegp_ctx* ctx, const double* X, const double* y, int n, int d, double nu, const double* theta0,
                  const double* lo, const double* hi, const double* starts, int n_restarts,
                  const hbegp_fit_options* opt, double* theta_best, double* lml_best, hbegp_model** model);
int hbegp_fit_f32(hbegp_ctx* ctx, const float* X, const float* y, int n, int d, double nu, const double* theta0,
                  const double* lo, const double* hi, const double* starts, int n_restarts,
                  const hbegp_fit_options* opt, double* theta_best, double* lml_best, hbegp_model** model);

/* One evaluation at fixed theta + K^-1 (fit.rs:33-68).  HBEGP_NOT_PD where the reference panics (fit.rs:55).  For f64 the result is
 * bit for bit what an evaluation of the same theta inside hbegp_fit_f64 produces (one order of operations for one theta). */
int hbegp_extend_f64(hbegp_ctx* ctx, const double* X, const double* y, int n, int d, double nu, const double* theta,
                     const double* lo, const double* hi, hbegp_model** model);
int hbegp_extend_f32(hbegp_ctx* ctx, const float* X, const float* y, int n, int d, double nu, const double* theta,
                     const double* lo, const double* hi, hbegp_model** model);

/* extend at the theta of `prior`, a model fitted on a PREFIX of these rows (the caller appends its validation samples
 * to the data the last model was built from, minimize.rs:629-644).  The leading floor(n_prior/128) diagonal blocks of L,
 * L^-1 and K^-1 are reused: O(n^2 k) instead of the reference's O(n^3) refactorisation (fit.rs:33-68); same results to
 * rounding.  Falls back to the full path when the prefix differs, n < n_prior or n_prior < 128.  *incremental (may be
 * NULL) reports which path ran.  HBEGP_NOT_PD where the reference panics (fit.rs:55). */
int hbegp_extend_from_f64(hbegp_ctx* ctx, hbegp_model* prior, const double* X, const double* y, int n, hbegp_model** model,
                          int* incremental);
int hbegp_extend_from_f32(hbegp_ctx* ctx, hbegp_model* prior, const float* X, const float* y, int n, hbegp_model** model,
                          int* incremental);

/* ---- model: mirrors predict() (predict.rs:7-52) and the FittedKernel fields -------------------------- */
/* mean[m]; var[m] or NULL.  var = c + 1e-5 - rowsum((K* K^-1) o K*), negatives clamped to 0 (predict.rs:25-48);
 * it excludes s2 (the reference predicts the latent function).  *n_warn (may be NULL) = number of variances below
 * -sqrt(1e-5) before clamping (the reference prints a warning for those, predict.rs:39-48). */
int hbegp_predict_f64(hbegp_model* model, const double* Xs, int m, double* mean, double* var, int* n_warn);
int hbegp_predict_f32(hbegp_model* model, const float* Xs, int m, float* mean, float* var, int* n_warn);

int hbegp_model_info(const hbegp_model* model, int* n, int* d, int* is_f32, double* nu, double* lml);
/* theta[p] (log space, clamped), alpha[n], kinv[n*n] full symmetric; any pointer may be NULL. */
int hbegp_model_get_f64(hbegp_model* model, double* theta, double* alpha, double* kinv);
int hbegp_model_get_f32(hbegp_model* model, double* theta, float* alpha, float* kinv);
/* Reference counting for `Clone`/`Drop` of the Rust wrapper (gpr.rs:53; models are kept for the whole run,
 * minimize.rs:331).  Device memory is freed on the last release. */
void hbegp_model_retain(hbegp_model* model);
void hbegp_model_release(hbegp_model* model);

/* ---- host-side optimiser, exposed for its own tests (mirrors minimize_by_gradient, gradmin.rs:35-60) --------- */
typedef double (*hbegp_objective_fn)(const double* x, double* grad, void* user);
/* Bounded L-BFGS minimisation; x is updated in place; returns the best objective value found. */
double hbegp_minimize_by_gradient(hbegp_objective_fn f, void* user, double* x, const double* lo, const double* hi,
                                  int n, int maxeval);

/* ---- test hook: raw copy of a slot's work matrix after the last evaluation (np x np row-major, np = n rounded up to 128;
 * which = 1: W1 (Schur complements / U), 2: W2 (X = L^-1), 3: W3 (the factor L; right-looking task queue and f32
 * refinement only), 4: the K^-1 buffer the last evaluation wrote (lower tiles, not mirrored).  out holds np*np elements of
 * the problem's type. */
int hbegp_problem_debug_get_f64(hbegp_problem* prob, int dev, int slot, int which, double* out);
int hbegp_problem_debug_get_f32(hbegp_problem* prob, int dev, int slot, int which, float* out);

/* ---- test hook (host only, no GPU): build the task queue of the device-scheduled factorisation for `nblocks`
 * 128-blocks (bk = contraction elements per stage: 16 for f64, 32 for f32; nodes up to small_h blocks wide use 64x64
 * tiles; nwg > 0: order the queue by a list schedule simulated for nwg workgroups; fine bit 0: per-row-block dependencies,
 * bit 1: must be 0 (a plan variant removed in round 5), bit 2: the tiles of K^-1 = X^T X follow in the same queue,
 * bit 3: the right-looking plan instead of the recursion, bit 4: its row-progressive inverse and K^-1) and
 * check it: queue order topological (=> deadlock-free for any number of resident workgroups), every wait for a full
 * count, no unordered access to a tile.  crit_us / sim_us: critical path and simulated makespan under the host's task
 * time estimates.  Returns HBEGP_OK or HBEGP_EINVAL with the reason in err. */
int hbegp_debug_dag_plan(int nblocks, int bk, int small_h, int nwg, int fine, int* ntasks, int* ncounters, int* nleaf,
                         double* gflop, double* crit_us, double* sim_us, char* err, int errlen);

/* ---- test hook (host only, no GPU): the host's bounded L-BFGS state machine (csrc/lbfgs_step.hpp) fed with a RECORDED sequence
 * of evaluations -- f[i] (objective, +inf for a failed evaluation) and g[i*n ..] (its gradient) are what evaluation i returned;
 * requested[i*n ..] receives the point the state machine asks for as evaluation i (requested[0] = the clipped start point),
 * n_requested how many points it asked for (<= count).  lo / hi: the box in the optimiser's coordinates.  memory <= 0: the
 * default.  The GPU tests replay the trace of a run of the persistent fit kernel (a wave-wide transcription of the same method,
 * gradmin.rs:35-60) through it and compare every point. */
int hbegp_debug_lbfgs_replay(int n, const double* x0, const double* lo, const double* hi, int maxeval, int memory, int fixed_work,
                             int count, const double* f, const double* g, double* requested, int* n_requested);

#ifdef __cplusplus
}
#endif
#endif /* HBEGP_H */
