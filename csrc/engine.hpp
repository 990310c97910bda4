// engine.hpp — device-side engine interface shared by kernels.hip (HIP kernels + launchers) and
// hbegp.cpp (host runtime, C ABI).  Everything here is plain C++; no torch, no oracle.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace hbegp {

constexpr int NB = 128;     // leaf (diagonal block) size; all matrices are padded to a multiple of NB
constexpr int MAXD = 64;    // max number of features
constexpr int MAXP = MAXD + 2;

// Per-evaluation hyper-parameters, linear space, already clamped (fit.rs:94-96).  Lives in device memory;
// the host refreshes it with one hipMemcpyAsync before each evaluation so captured graphs stay valid.
struct EvalParams {
  double noise;      // sigma^2
  double amp;        // c
  double ell[MAXD];  // length scales
  unsigned long long seq;  // serial number of the evaluation (host); the last kernel echoes it into EvalOut::seq of the pinned result block
};

// Per-evaluation scalar results (device -> pinned host).
struct EvalOut {
  double lml;
  double yalpha;   // y^T alpha
  double logdet;   // sum_i log L_ii
  double grad[MAXP];
  int info;        // 0 ok, else 1 + index of the failing pivot (not positive definite)
  int n_warn;      // predict: variances below -sqrt(1e-5)
  int done;        // bit 0: lml written by this evaluation, bit 1: gradient written (a kernel that never ran leaves them clear)
  int pad_;
  unsigned long long seq;  // pinned copy only: EvalParams::seq of the evaluation these results belong to, stored LAST (system scope)
};

// What the first kernel of an evaluation (the kernel-matrix assembly) does besides its tiles when the evaluation is driven
// through pinned host memory (hbegp.cpp, HBEGP_HOSTIO): its workgroup 0 copies the parameters it reads from the host block to
// device memory for the later kernels, poisons the result block and clears the task queue's control words -- one graph node in
// front of the long kernel instead of four (parameter copy, reset kernel, assembly, memset), each of which the GPU spent
// ~30-50 us waiting for while three host threads were enqueuing their graphs.
struct EvalPrologue {
  EvalParams* dP = nullptr;   // device copy of the parameters (null: not the first kernel -- `info` is honoured as usual)
  EvalOut* out = nullptr;     // device result block to poison
  int* ctrl = nullptr;        // task-queue control words to clear (may be null)
  int ctrl_words = 0;
};

// One tile-GEMM operation on row-major matrices, in units of TILE x TILE tiles (global tile coordinates).
//   C(ti,tj) = beta*C(ti,tj) + alpha * sum_{tk in range(ti,tj)} opA(ti,tk) * opB(tk,tj)
// a_kmajor = 0: A tile (ti,tk) is read from A[ti*T.., tk*T..] (rows = output rows, contraction along columns)
// a_kmajor = 1: A tile is read from A[tk*T.., ti*T..] (rows = contraction index)        (same for B with tj)
struct GemmOp {
  const void* A;
  const void* B;
  void* C;
  int lda, ldb, ldc;
  int a_kmajor, b_kmajor;
  int ci0, cj0;   // first output tile (row, col)
  int mi, nj;     // output tiles (rows, cols)
  int c_lower;    // only tiles with global ti >= tj
  int k0, k1;     // contraction tile range [k0, k1)
  int klim;       // 0 none | 1: k <= tj | 2: k >= tj | 3: k <= ti | 4: k >= ti   (global tile coordinates)
  int maskA, maskB;  // operand is lower triangular (its strict upper part is zero IN MEMORY; only used for flop accounting)
  int ntiles;     // output tiles of this op (set by the launcher)
  int reverse;    // walk the tile list backwards (set by the launcher)
  int alpha_neg;  // alpha = -1 instead of +1
  int beta_one;   // beta = 1 instead of 0
};

constexpr int MAXOPS = 4;  // independent tile-GEMM operations per launch

struct GemmLaunch {
  GemmOp op[MAXOPS];
  int nops;
  const int* info;  // device flag: kernels return immediately when *info != 0
  // optional static schedule (device memory): workgroup b runs items [sched_off[b], sched_off[b+1]);
  // item = op << 30 | local tile row << 16 | local tile col, in units of the launch tile size
  const int* sched_off;
  const unsigned* sched_items;
  int sched_nwg;
};

// ---- device-scheduled factorisation: one persistent launch pulls leaf / tile tasks from an ordered queue ----------
// (dag_kernel.inc.hpp; plan built on the host by dag_plan.hpp).  A task waits until its counters have reached their
// values, runs, publishes its results write-through and bumps its own counter.  The queue order is a topological
// order of the dependency graph, so any number of resident workgroups >= 1 makes progress (no co-residency needed).
enum : uint16_t {
  DAG_GEMM_128x64 = 0, DAG_GEMM_64x64 = 1, DAG_LEAF = 2,
  // (3..7 were the kernel-matrix tiles and the alpha / lml reductions as tasks of the same queue, rounds 2-4: measured slower
  // than the launches around the queue, removed in round 5)
  // round 4: a 32x64 output tile whose operands (at most 128 contraction elements at a time, both stored [outer][k]) are
  // fetched in ONE shot -- the two products between consecutive diagonal blocks of the right-looking plan (L(k+1,k) =
  // A(k+1,k) X_kk^T and A(k+1,k+1) -= L(k+1,k) L(k+1,k)^T), where a tile's latency counts and its throughput does not.
  // Same accumulation order per element as the other tiles (k ascending in MFMA steps of four): same bits.
  DAG_GEMM_32x64 = 8,
  // 128x128 output tile for deep products without beta = 1 (no room in the LDS for the old values beside the stage buffers): half
  // the tasks, 8 MFMAs per 6 fragment reads instead of 4 per 4.  Same accumulation order per element: same bits.
  DAG_GEMM_128x128 = 9,
};
enum : uint16_t {
  DAGF_ABUF = 1,   // operand A lives in W2 (else W1)
  DAGF_BBUF = 2,
  DAGF_CBUF = 4,
  DAGF_AKM = 8,    // A tile is read with the contraction index along rows
  DAGF_BKM = 16,
  DAGF_NEG = 32,   // alpha = -1
  DAGF_ACC = 64,   // beta = 1
  DAGF_CKINV = 128,  // the result goes to the launch's K^-1 buffer (tiles of K^-1 = X^T X behind the recursion); skipped when
                     // the launch asks for the factorisation only
  // right-looking plan: the Cholesky factor L itself lives in a third work matrix (W3) until the inverse is complete
  DAGF_A3 = 256,   // operand A lives in W3 (overrides DAGF_ABUF)
  DAGF_B3 = 512,
  DAGF_C3 = 1024,  // the result goes to W3 (overrides DAGF_CBUF)
  // with DAGF_ACC (f64 only, no DAGF_NEG): the old values of the output tile START the accumulation instead of being added at the
  // end -- the second part of a contraction split in two then continues the first part's MFMA chain bit for bit
  DAGF_CINIT = 2048,
};
constexpr int DAG_MAXWAIT = 4;
constexpr int DAG_MAXSIG = 3;
constexpr uint16_t DAG_NOSIG = 0xffff;
struct DagTask {
  uint16_t kind, flags;
  int32_t row0, col0;  // origin of the output tile (elements); leaf: row0 = index of the 128-block
  int32_t kbeg, kend;  // contraction range (elements, whole stages)
  uint16_t nwait;
  uint16_t sig[DAG_MAXSIG];    // counters bumped once the task's results are visible (DAG_NOSIG: unused)
  uint16_t wcnt[DAG_MAXWAIT];  // counters waited for ...
  uint16_t wval[DAG_MAXWAIT];  // ... to reach these values (always the counter's full count)
  uint16_t cost;               // host-side estimate (tenths of a microsecond) used to order the queue
  uint16_t pad_[1];
};
static_assert(DAG_MAXSIG == 3, "DagTask::sig has three entries: dword 5 high half, dword 6 low half, dword 6 high half");
static_assert(sizeof(DagTask) == 48, "DagTask layout");  // dag_kernel decodes it dword by dword: keep the field order
constexpr int DAG_CTRL_WORDS = 4;     // ctrl[0] queue head, [1] first task that gave up waiting (+1), [2..3] spare; counters follow
constexpr int DAG_INFO_TIMEOUT = -2;  // written to EvalOut::info when a wait exceeded its bound (a bug, never a data property)
struct DagLaunch {
  const DagTask* tasks;
  int ntasks;
  int* ctrl;
  void* W1;
  void* W2;
  void* W3;         // right-looking plan: the factor L (DAGF_A3 / B3 / C3)
  void* Kinv;       // target of the DAGF_CKINV tiles (null: skip them)
  int ld;
  void* ldiag;
  int* info;
  unsigned long long* trace;  // optional (diagnostics): per task [pulled, inputs ready, computed, published] on the 100 MHz clock, then the CU id
  unsigned long long wait_ticks;  // bound of one dependency wait in ticks of the 100 MHz clock (host: from the plan's simulated makespan)
  int leaf_dbg;                   // debug bits of the diagonal-block tasks (tests: 16 = the helper wave starts late, HBEGP_LEAF_DBG)
};
template <typename T>
void launch_dag(const DagLaunch& g, int nwg, hipStream_t s);
int dag_stage_depth(bool is_f32);  // contraction elements per pipeline stage of the task-queue tiles (task ranges are whole stages)

// ---- launchers (kernels.hip), T in {double, float} ------------------------------------------------------------
template <typename T>
void launch_gemm(const GemmLaunch& g, int tile, hipStream_t s);  // tile in {32, 64, 128}

template <typename T>
void launch_kmat(const T* X, int n, int d, int np, int nu2, const EvalParams* P, T* W, const int* info, hipStream_t s,
                 const EvalPrologue* pro = nullptr);

// Factor the 128x128 diagonal block `blk` of W1 (lower) in place -> X_blk = L_blk^-1 into W2's block, diag(L) -> ldiag;
// W3 (optional): the lower triangle of L_blk itself.
template <typename T>
void launch_leaf(T* W1, T* W2, int ld, int blk, T* ldiag, int* info, hipStream_t s, int dbg = 0, T* W3 = nullptr);

// alpha = X^T (X y), lml pieces.  X lower-triangular np x np in W2.  part: [np/256][np] scratch.
template <typename T>
// ticket (device int, zero between launches): the last workgroup of the reduction forms the lml itself (no lml_final launch).
void launch_alpha_lml(const T* Xinv, int np, int n, const T* y, const T* ldiag, T* wbuf, double* part, T* alpha,
                      EvalOut* out, const int* info, hipStream_t s, int* ticket = nullptr);

// lml gradient: g_j = 1/2 sum_ik (alpha alpha^T - Kinv)_ik dK_ik/dtheta_j without materialising dK.
template <typename T>
// ticket (device int, zero between launches): the launch's last workgroup finalises the gradient itself and, with hout, copies
// the result block to the pinned one and publishes the evaluation's serial number -- no finalize / publish launches behind it
void launch_gradtrace(const T* X, int n, int d, int np, int nu2, const EvalParams* P, const T* Kinv, const T* alpha,
                      double* part, EvalOut* out, const int* info, hipStream_t s, int* ticket = nullptr, EvalOut* hout = nullptr);
size_t gradtrace_part_elems(int np, int d);

template <typename T>
void launch_symmetrize(T* A, int np, hipStream_t s);  // mirror lower -> upper

// predict: Kstar [mp x np] (rows = candidates), mean, var
template <typename T>
void launch_kstar(const T* Xs, int m, int mp, const T* X, int n, int d, int np, int nu2, const EvalParams* P, T* Ks,
                  hipStream_t s);
template <typename T>
void launch_pred_mean(const T* Ks, int m, int np, const T* alpha, T* mean, hipStream_t s);
template <typename T>
void launch_pred_var(const T* Ks, const T* Q, int m, int np, const EvalParams* P, T* var, EvalOut* out, hipStream_t s);

// predict for m <= PRED_SMALL_MAX candidates without the 128-row padding: reads L^-1 once (row dots against the m
// cross-kernel vectors).  Ks: [PRED_SMALL_MAX][np] scratch, pmean: [(np+255)/256][PRED_SMALL_MAX], w: [n][PRED_SMALL_MAX].
constexpr int PRED_SMALL_MAX = 16;
template <typename T>
void launch_predict_small(const T* Xs, int m, const T* X, int n, int d, int np, int nu2, const EvalParams* P, const T* alpha, const T* Linv,
                          T* Ks, double* pmean, double* w, int want_var, T* mean, T* var, int* n_warn, hipStream_t s);

// One evaluation of a problem of at most 128 rows (np = NB, d <= SMALL_EVAL_MAXD) in ONE launch, everything in the LDS
// (kernels.hip: small_eval_kernel).  mode bit 0: alpha + lml, bit 1: K^-1, bit 2: gradient.
constexpr int SMALL_EVAL_MAXD = 32;
struct SmallEval {
  const void* X;
  const void* y;
  int n, d;
  const EvalParams* P;
  void* W2;
  void* ldiag;
  void* Kinv;
  void* alpha;
  EvalOut* out;    // device memory: the diagonal block's positive-definite flag (atomics)
  EvalOut* hout;   // where the results go: the slot's pinned host block (written by the kernel itself: no copy node) or `out`
  int mode;
};
template <typename T>
void launch_small_eval(const SmallEval& g, int nu2, hipStream_t s);

// One optimiser run of a fit of at most 128 rows in ONE launch (kernels.hip: small_fit_kernel): the evaluation above in a loop
// with the bounded L-BFGS step and the fit's objective wrapper (clamping, failed evaluations, capture of the best) on the device.
struct LbfgsState;
struct SmallFitResult {
  double best_lml;
  double best_theta[MAXP];   // as the optimiser passed it (log space, unclamped noise)
  double best_params[MAXP];  // the clamped linear-space parameters the captured evaluation ran with (the DEVICE's exp of best_theta: the
                             // model's L^-1 is refactored from exactly these, not from the host's exp of the same theta)
  int best_idx;              // ping-pong buffer (K^-1, alpha) that holds the captured evaluation; -1: every evaluation failed
  int best_eval;             // its index within the run
  int n_evals, n_not_pd;
};
struct SmallFit {
  SmallEval ev;              // X, y, n, d, P (device workspace, written by the kernel), W2, ldiag, out = hout (device workspace)
  void* Kinv[2];
  void* alpha[2];
  void* Xinv[2];             // L^-1 and diag(L) ping-pong with K^-1 and alpha (null: ev.W2 / ev.ldiag every time): the model is built
  void* ldiag[2];            // from the captured evaluation's own factor instead of factoring once more at the captured theta
  LbfgsState* st;            // device workspace
  const double* x0;          // start point (log space), p = d + 2 entries
  const double* lo;          // linear-space box
  const double* hi;
  int maxeval, memory, fixed_work;
  double pgtol, ftol;
  SmallFitResult* res;       // device
  SmallFitResult* hres;      // pinned host copy of *res, written at the end of the run (null: none) ...
  unsigned long long* hdone; // ... followed, behind a system-scope fence, by 1 in this pinned word: the host thread that owns the
                             // run polls it -- the launch may carry other fits' runs (hbegp.cpp: SmallBatcher) and need not be over
  double* trace_theta;       // [trace_cap][p] (device; may be null with trace_cap = 0)
  double* trace_lml;         // [trace_cap]
  double* trace_grad;        // [trace_cap][p]
  int trace_cap;
};
// fs_dev: device array of nruns descriptors; run r is workgroup r of one launch
template <typename T>
void launch_small_fit(const SmallFit* fs_dev, int nruns, int nu2, hipStream_t s);

void launch_set_info(int* info, int value, hipStream_t s);
// start of an evaluation: info = n_warn = done = 0, lml and gradient poisoned with NaN (a launch that was rejected or
// skipped can then never pass for a result)
void launch_reset_out(EvalOut* out, hipStream_t s);
// Last kernel of an evaluation driven through pinned host memory: copies the device result block to the pinned one and then,
// behind a system-scope fence, stores P->seq into hout->seq -- the host thread spins on that word instead of sleeping in
// hipStreamSynchronize (interrupt wake-up: ~50 us of every evaluation).
void launch_publish_out(const EvalOut* out, EvalOut* hout, const EvalParams* P, hipStream_t s);
// throws nothing: returns the first pending launch error of the calling thread's device (hipGetLastError)
void init_kernels();

// *flag |= 1 when the two device arrays differ anywhere in [0, count)
template <typename T>
void launch_prefix_differs(const T* a, const T* b, size_t count, int* flag, hipStream_t s);

}  // namespace hbegp
