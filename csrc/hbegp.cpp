// hbegp.cpp — host runtime and C ABI of the MI355X GP engine (see include/hbegp.h).
//
// Layout in HBM (per evaluation slot, all row-major with leading dimension np = n rounded up to 128):
//   W1   np x np   kernel matrix K (lower) -> trailing Schur complements / scratch during the factorisation
//   W2   np x np   X = L^-1 (lower), built block by block while the Cholesky recursion runs
//   Kinv np x np   x2 (ping-pong): K^-1 = X^T X (lower); the copy holding the best lml so far is never overwritten
//   alpha np       x2 (ping-pong)
// The padding rows/cols carry an identity block, so every kernel works on whole 128-tiles.
//
// One evaluation (lml.rs:29-79) = kmat -> chol_inv recursion (leaf + tile GEMMs) -> alpha/lml -> lauum -> gradtrace,
// captured once per slot into a hipGraph and replayed for every theta the optimiser asks for.
#include "../include/hbegp.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <queue>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "dag_plan.hpp"
#include "engine.hpp"
#include "lbfgsb.hpp"
#include "lbfgs_step.hpp"


using namespace hbegp;

// ---------------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error = "";
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
struct HipError {
  hipError_t e;
  const char* what;
  int line;
};
#define HIPCHECK(x)                                \
  do {                                             \
    hipError_t _e = (x);                           \
    if (_e != hipSuccess) throw HipError{_e, #x, __LINE__}; \
  } while (0)

// A rejected launch (bad configuration, LDS request refused, ...) is only reported through hipGetLastError: checked after
// every group of launches, also while a stream is being captured.
#define CHECK_LAUNCHES()                                                        \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) throw HipError{_e, "kernel launch", __LINE__};        \
  } while (0)

static int hip_fail(const HipError& he) {
  return fail(he.e == hipErrorOutOfMemory ? HBEGP_ENOMEM : HBEGP_EHIP, "HIP error %d (%s) in %s at hbegp.cpp:%d", (int)he.e,
              hipGetErrorString(he.e), he.what, he.line);
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

struct hbegp_ctx {
  std::vector<int> devs;
};
static std::atomic<int> g_live_ctx{0};  // the block pool is process-global: it is trimmed when the last context goes

// ---------------------------------------------------------------------------------------------------------------
// Device-memory pool for the large work matrices (np x np).  hipMalloc/hipFree of 128 MiB blocks costs tens of
// milliseconds per fit; the caller fits one model per generation with slowly growing n, so blocks are recycled by
// exact size.  Recycled blocks hold finite numbers from their previous life, which is all the engine requires of
// never-written regions (strict upper triangles).
#include <chrono>
#include <condition_variable>
#include <map>
#include <tuple>
struct DevPool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_list;
  // Fresh blocks are cleared on a NON-BLOCKING stream of the pool's own (one per device), never on the null stream: several
  // host threads may be fitting on one context (hbegp.h: "re-entrant per ctx"), and any null-stream operation issued while
  // another thread captures a graph fails with hipErrorStreamCaptureImplicit -- besides synchronising with nothing the engine
  // runs on (its streams are non-blocking), which is how round 4's late clear came about.
  std::map<int, hipStream_t> clear_stream;
  hipStream_t stream_of(int dev) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = clear_stream.find(dev);
    if (it != clear_stream.end()) return it->second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) throw HipError{hipErrorOutOfMemory, "hipStreamCreateWithFlags (pool)", __LINE__};
    clear_stream[dev] = st;
    return st;
  }
  size_t cached = 0;
  static constexpr size_t MAX_CACHED = (size_t)24 << 30;
  void* get(int dev, size_t bytes, bool* fresh) {
    // Test switches (tests/test_gpu_parity.py::test_fresh_pool_blocks_are_cleared_before_their_first_writer):
    //   HBEGP_POOL_FRESH=1         never recycle: every block is a fresh hipMalloc, as in a fresh process
    //   HBEGP_POOL_NULL_DELAY_MB=N queue an N MiB fill on the null stream in front of every fresh block's clear (makes the
    //                              clear LATE on purpose: a first writer on a non-blocking stream then runs before it)
    //   HBEGP_POOL_OLD_CLEAR=1     round 1-4's clear: hipMemset with no synchronisation (the race the test must see fail)
    const bool always_fresh = getenv("HBEGP_POOL_FRESH") != nullptr && atoi(getenv("HBEGP_POOL_FRESH")) != 0;
    if (!always_fresh) {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_list.find({dev, bytes});
      if (it != free_list.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        cached -= bytes;
        *fresh = false;
        return p;
      }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      trim();  // give cached blocks back and retry once
      e = hipMalloc(&p, bytes);
      if (e != hipSuccess) throw HipError{e, "hipMalloc (pool)", __LINE__};
    }
    // a fresh block is cleared once, here: every later owner may rely on "finite numbers everywhere" (recycled blocks
    // hold finite results of their previous life)
    // hipMemset on device memory returns before the fill has run (it is queued on the null stream), and the engine's streams are
    // non-blocking ones that do not wait for the null stream: without the synchronisation below the fill runs CONCURRENTLY with,
    // or after, the block's first writer on such a stream (the model's copy of L^-1 in make_model, the kernel-matrix tiles of an
    // evaluation) and zeroes part or all of what was written -- seen once in round 4 as predictive variances that were off by
    // O(1) on a fresh process, where every block is a fresh one (profiles/r04_memset_race.txt).
    // (the work matrices only: with every small array delayed as well the backlog on the null stream outlasts the whole test)
    const int delay_mb = (bytes >= ((size_t)1 << 20) && getenv("HBEGP_POOL_NULL_DELAY_MB")) ? atoi(getenv("HBEGP_POOL_NULL_DELAY_MB")) : 0;
    if (delay_mb > 0) {
      static std::mutex dmu;
      static std::map<int, std::pair<void*, size_t>> scratch;  // per device, kept for the life of the process (a test switch)
      std::lock_guard<std::mutex> lk(dmu);
      auto& sc = scratch[dev];
      const size_t want = (size_t)delay_mb << 20;
      if (sc.second < want) {
        if (sc.first) (void)hipFree(sc.first);
        sc = {nullptr, 0};
        if (hipMalloc(&sc.first, want) == hipSuccess) sc.second = want;
      }
      if (sc.first) (void)hipMemsetAsync(sc.first, 0x5a, sc.second, nullptr);
    }
    if (bytes >= ((size_t)1 << 20) && getenv("HBEGP_POOL_OLD_CLEAR") != nullptr && atoi(getenv("HBEGP_POOL_OLD_CLEAR")) != 0) {
      e = hipMemset(p, 0, bytes);
    } else if (delay_mb > 0) {
      // the test's delayed form: the clear queues behind the long null-stream fill, and the synchronisation covers both
      e = hipMemsetAsync(p, 0, bytes, nullptr);
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    } else {
      hipStream_t st = stream_of(dev);
      e = hipMemsetAsync(p, 0, bytes, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
      (void)hipFree(p);
      throw HipError{e, "hipMemset (pool)", __LINE__};
    }
    *fresh = true;
    return p;
  }
  void put(int dev, void* p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    if (cached + bytes > MAX_CACHED) {
      (void)hipFree(p);
      return;
    }
    free_list[{dev, bytes}].push_back(p);
    cached += bytes;
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_list)
      for (void* p : kv.second) (void)hipFree(p);
    free_list.clear();
    cached = 0;
  }
};
static DevPool g_pool;

// Pinned host blocks and streams are recycled as well: a fit of a small problem lasts ~10 ms, and creating + destroying its
// slots' streams, pinned parameter / result blocks and ~8 small device arrays per slot cost ~2 ms of it (every hipFree waits for
// the device).  Streams are handed back only after they have been synchronised.
struct HostPool {
  std::mutex mu;
  std::map<size_t, std::vector<void*>> free_list;
  void* get(size_t bytes) {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_list.find(bytes);
      if (it != free_list.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        return p;
      }
    }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);  // fine-grained: a host thread spins on words the GPU writes (Problem::wait_eval)
    if (e != hipSuccess) throw HipError{e, "hipHostMalloc (pool)", __LINE__};
    return p;
  }
  void put(void* p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    auto& v = free_list[bytes];
    if (v.size() >= 64) { (void)hipHostFree(p); return; }
    v.push_back(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_list)
      for (void* p : kv.second) (void)hipHostFree(p);
    free_list.clear();
  }
};
static HostPool g_host_pool;
// HIP maps its streams onto a few hardware queues per device (GPU_MAX_HW_QUEUES, default 4) in the order the streams are
// created, and two streams that share a queue run one after the other.  So the pool is keyed by what a stream is FOR: the
// evaluation slots of a fit always get the same streams back (created first, on distinct queues), whatever a model's or the
// block pool's stream does in between -- with ONE free list a fit's three slots got, every other fit, two streams of one queue
// (measured: optimiser runs of config M 561 / 900 / 561 / 900 ms).
enum { STREAM_SLOT = 0, STREAM_MODEL = 1, STREAM_BATCH = 2 };
struct StreamPool {
  std::mutex mu;
  std::map<std::pair<int, int>, std::vector<hipStream_t>> free_list;
  hipStream_t get(int dev, int kind = STREAM_SLOT) {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_list.find({dev, kind});
      if (it != free_list.end() && !it->second.empty()) {
        hipStream_t s = it->second.back();
        it->second.pop_back();
        return s;
      }
    }
    hipStream_t s = nullptr;
    hipError_t e;
    static const bool batch_prio = env_int("HBEGP_BATCH_STREAM_PRIORITY", 1) != 0;
    if (kind == STREAM_BATCH && batch_prio) {
      // the runtime keeps a pool of hardware queues PER PRIORITY: a stream of another priority never shares its queue with the
      // normal-priority streams of the fits -- whose copies and small kernels would otherwise wait for the ~10 ms persistent grid
      // of a small-fit batch whenever they land on its queue
      int least = 0, greatest = 0;
      e = hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (e != hipSuccess) throw HipError{e, "hipDeviceGetStreamPriorityRange", __LINE__};
      e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    } else {
      e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    }
    if (e != hipSuccess) throw HipError{e, "hipStreamCreate (pool)", __LINE__};
    return s;
  }
  void put(int dev, hipStream_t s, int kind = STREAM_SLOT) {  // the caller has synchronised it
    if (!s) return;
    std::lock_guard<std::mutex> lk(mu);
    auto& v = free_list[{dev, kind}];
    if (v.size() >= 64) { (void)hipStreamDestroy(s); return; }
    v.push_back(s);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_list)
      for (hipStream_t s : kv.second) (void)hipStreamDestroy(s);
    free_list.clear();
  }
};
static StreamPool g_stream_pool;

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Fits of up to 128 rows from several host threads share their launches.  An optimiser run of such a fit is one workgroup of a
// persistent kernel (small_fit_kernel, ~10 ms); with a launch per fit the fits of different threads sit on different streams, HIP
// maps streams onto a handful of hardware queues, and a stream that shares its queue with another fit's persistent kernel waits
// for all of it (measured: 16 threads, 16 hardware queues: 560 fits/s where 16 / 10.5 ms = 1,500 would fit the chip; more
// queues made it worse).  So: a thread that arrives with its runs joins the batch that is open for its (device, kernel) or opens
// one; whoever opened it waits until no other thread is on its way (threads announce themselves when they enter the small-fit
// path: SmallBatcher::arriving) and none has just come back from a small fit (g_small_recent below), at most 3 ms of a ~10 ms fit (HBEGP_SMALL_BATCH_US; measured at 16 threads: 0.3 ms 724, 1 ms 959, 3 ms
// 1,113 fits/s -- under load a thread needs 2-3 ms to set its problem up), and launches all runs in ONE grid; every thread then polls the pinned words of its
// OWN runs (SmallFit::hdone) -- it neither waits for the other fits of the launch nor synchronises a stream.  One thread alone
// launches at once.  The workgroups of a launch do not depend on each other, so a run's bits do not depend on its company.
struct SmallBatch {
  std::vector<SmallFit> fits;
  bool closed = false, launched = false;
  int refs = 0;
  std::exception_ptr err;  // launch failure: every participant rethrows it
  void* fs_dev = nullptr;
  size_t fs_bytes = 0;
  hipStream_t stream = nullptr;
  int dev = 0;
};
struct SmallBatcher {
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::tuple<int, int, bool>, std::shared_ptr<SmallBatch>> open;  // (device id, nu2, f32)
  std::atomic<int> arriving{0};  // threads inside the small-fit path that have not handed in their runs yet
};
static SmallBatcher g_small_batcher;
static std::mutex g_small_host_mu;      // turn-taking of the host-side phases of small fits (do_fit)
static std::atomic<int> g_small_active{0};  // threads inside a small fit
// Threads that have just come back from a small fit are the ones most likely to bring the next one: each thread keeps the time of
// its last return in a slot of this table (0 while it is inside a fit), and the thread that opens a batch also waits for those
// whose return is younger than HBEGP_SMALL_BATCH_RECENT_US (4 ms) -- without this, threads that finish together drift apart
// again (whoever is back first sees nobody on the way and launches alone: measured, 16 threads, 68 of 101 grids carried one fit).
constexpr int SMALL_RECENT_SLOTS = 64;
static std::atomic<long long> g_small_recent[SMALL_RECENT_SLOTS];
static std::atomic<int> g_small_recent_next{0};
static int small_recent_slot() {
  static thread_local int slot = -1;
  if (slot < 0) slot = g_small_recent_next.fetch_add(1) % SMALL_RECENT_SLOTS;
  return slot;
}
static long long steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int small_recent_others(long long window_ns) {
  const long long now = steady_ns();
  const int mine = small_recent_slot();
  int c = 0;
  for (int i = 0; i < SMALL_RECENT_SLOTS; ++i) {
    const long long ts = g_small_recent[i].load(std::memory_order_relaxed);
    if (i != mine && ts != 0 && now - ts < window_ns) ++c;
  }
  return c;
}
struct SmallArrival {  // RAII: "I am on my way with runs"; on destruction: "I am back" (the time of the return is left in the table)
  bool counted = false, entered = false;
  void announce() {
    if (!counted) {
      g_small_batcher.arriving.fetch_add(1);
      counted = true;
      entered = true;
      g_small_recent[small_recent_slot()].store(0, std::memory_order_relaxed);
    }
  }
  void arrived() {
    if (counted) {
      g_small_batcher.arriving.fetch_sub(1);
      counted = false;
      g_small_batcher.cv.notify_all();
    }
  }
  ~SmallArrival() {
    arrived();
    if (entered) g_small_recent[small_recent_slot()].store(steady_ns(), std::memory_order_relaxed);
  }
};
static void small_batch_unref_locked(SmallBatch& b);
// Hands in `mine` (runs of one fit on device `dev`); returns the batch once its grid has been launched.  The caller polls its runs'
// SmallFit::hdone words and then calls small_batch_leave.  Throws if the launch failed.
template <typename T>
static std::shared_ptr<SmallBatch> small_batch_submit(int dev, int nu2, const std::vector<SmallFit>& mine, SmallArrival& arrival) {
  SmallBatcher& B = g_small_batcher;
  static const int window_us = env_int("HBEGP_SMALL_BATCH_US", 3000);
  std::unique_lock<std::mutex> lk(B.mu);
  const auto key = std::make_tuple(dev, nu2, sizeof(T) == 4);
  std::shared_ptr<SmallBatch> b = B.open[key];
  const bool leader = !b;
  if (leader) {
    b = std::make_shared<SmallBatch>();
    b->dev = dev;
    B.open[key] = b;
  }
  b->fits.insert(b->fits.end(), mine.begin(), mine.end());
  ++b->refs;
  arrival.arrived();  // (notifies: a leader waiting for the stragglers looks again)
  if (leader) {
    const auto t_open = std::chrono::steady_clock::now();
    static const long long recent_ns = 1000ll * env_int("HBEGP_SMALL_BATCH_RECENT_US", 4000);
    if (window_us > 0) {
      // until nobody is on the way and nobody has just come back (their marks expire by themselves: look again every 200 us)
      const auto deadline = t_open + std::chrono::microseconds(window_us);
      while (std::chrono::steady_clock::now() < deadline && (B.arriving.load() != 0 || small_recent_others(recent_ns) != 0))
        B.cv.wait_for(lk, std::chrono::microseconds(200));
    }
    static const bool log_batches = env_int("HBEGP_SMALL_BATCH_LOG", 0) != 0;
    if (log_batches)
      fprintf(stderr, "small-fit batch: %zu runs, waited %.0f us for stragglers (%d still on their way, %d just back)\n", b->fits.size(),
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_open).count(), B.arriving.load(), small_recent_others(recent_ns));
    b->closed = true;
    B.open.erase(key);
    lk.unlock();
    try {
      HIPCHECK(hipSetDevice(dev));
      b->fs_bytes = sizeof(SmallFit) * (size_t)round_up((int)b->fits.size(), 16);
      bool fresh = false;
      b->fs_dev = g_pool.get(dev, b->fs_bytes, &fresh);
      b->stream = g_stream_pool.get(dev, STREAM_BATCH);
      HIPCHECK(hipMemcpyAsync(b->fs_dev, b->fits.data(), sizeof(SmallFit) * b->fits.size(), hipMemcpyHostToDevice, b->stream));  // (the batch outlives the copy)
      launch_small_fit<T>(static_cast<const SmallFit*>(b->fs_dev), (int)b->fits.size(), nu2, b->stream);
      CHECK_LAUNCHES();
    } catch (...) {
      b->err = std::current_exception();
    }
    lk.lock();
    b->launched = true;
    B.cv.notify_all();
  } else {
    B.cv.wait(lk, [&] { return b->launched; });
  }
  if (b->err) {
    std::exception_ptr e = b->err;
    if (b->stream) (void)hipStreamSynchronize(b->stream);  // (a copy may have been queued before the launch failed)
    small_batch_unref_locked(*b);
    lk.unlock();
    std::rethrow_exception(e);
  }
  return b;
}
static void small_batch_unref_locked(SmallBatch& b) {
  if (--b.refs == 0) {  // every run of the grid is over (each participant saw its words, or waited for the stream)
    if (b.fs_dev) g_pool.put(b.dev, b.fs_dev, b.fs_bytes);
    if (b.stream) g_stream_pool.put(b.dev, b.stream, STREAM_BATCH);
    b.fs_dev = nullptr;
    b.stream = nullptr;
  }
}
// `finished`: the caller has seen every one of its runs' hdone words (false: it gave up -- the grid is waited for here, its
// workspaces are about to be released)
static void small_batch_leave(const std::shared_ptr<SmallBatch>& b, bool finished) {
  if (!b) return;
  if (!finished && b->stream) {
    (void)hipSetDevice(b->dev);
    (void)hipStreamSynchronize(b->stream);
  }
  std::lock_guard<std::mutex> lk(g_small_batcher.mu);
  small_batch_unref_locked(*b);
}
constexpr int DAG_PROG_MAX_BLOCKS = 20;  // right-looking plan: the inverse and K^-1 follow the chain row by row up to this many 128-blocks
constexpr int DAG_MIN_BLOCKS_FIT = 6;   // evaluations (a fit's, `extend`'s, single ones) use the task queue from this many 128-blocks on

// theta (log space) -> clamped linear-space parameters (fit.rs:94-96)
static void theta_to_params(const double* theta, const double* lo, const double* hi, int d, EvalParams* P) {
  auto clampv = [&](double v, int i) {
    if (!lo || !hi) return v;
    if (v < lo[i]) return lo[i];  // bounded_value.rs:43-56
    if (hi[i] < v) return hi[i];
    return v;
  };
  P->noise = std::exp(theta[0]);  // not clamped (fit.rs:96)
  P->amp = clampv(std::exp(theta[1]), 1);
  for (int k = 0; k < d; ++k) P->ell[k] = clampv(std::exp(theta[2 + k]), 2 + k);
}

// ---------------------------------------------------------------------------------------------------------------
// timing support for hbegp_problem_time_eval
struct PhaseTimer {
  enum Kind { KMAT = 0, GEMM = 1, LEAF = 2, LAUUM = 3, ALPHA = 4, GRAD = 5, DAG = 6, NKIND = 7 };
  struct Rec {
    hipEvent_t a, b;
    int kind, tile;
    double gflop;
  };
  std::vector<Rec> recs;
  hipStream_t s;
  void begin(int kind, int tile = 0, double gflop = 0) {
    Rec r;
    HIPCHECK(hipEventCreate(&r.a));
    HIPCHECK(hipEventCreate(&r.b));
    r.kind = kind; r.tile = tile; r.gflop = gflop;
    HIPCHECK(hipEventRecord(r.a, s));
    recs.push_back(r);
  }
  void end() { HIPCHECK(hipEventRecord(recs.back().b, s)); }
};

static int pick_tile(int tiles128) {
  static const int forced = env_int("HBEGP_TILE", 0);
  if (forced == 32 || forced == 64 || forced == 128) return forced;
  // 128-tiles only when a launch has >= 600 of them (n >= 8192): measured n=8192 15.1 -> 13.7 ms/evaluation with them,
  // n=4096 LAUUM (528 tiles) 0.44 ms with 64-tiles vs 0.77 ms with 128-tiles
  if (tiles128 >= env_int("HBEGP_T128_MIN", 600)) return 128;
  if (tiles128 >= env_int("HBEGP_T64_MIN", 128)) return 64;  // h=1024 SYRK+U (100 tiles): 58 us with 32-tiles vs 83 us with 64-tiles
  return 32;
}

// algorithmic flops of one op (in 128-tile units): 2*128^3 per (tile, k-tile) pair, half for pairs that only
// touch a triangle (storage-diagonal tile of a triangular operand, or a diagonal tile of a symmetric result)
static double op_gflop(const GemmOp& op) {
  double pairs = 0;
  for (int i = 0; i < op.mi; ++i)
    for (int j = 0; j < (op.c_lower ? i + 1 : op.nj); ++j) {
      const int ti = op.ci0 + i, tj = op.cj0 + j;
      int ka = op.k0, kb = op.k1;
      if (op.klim == 1) kb = std::min(kb, tj + 1);
      if (op.klim == 2) ka = std::max(ka, tj);
      if (op.klim == 3) kb = std::min(kb, ti + 1);
      if (op.klim == 4) ka = std::max(ka, ti);
      for (int k = ka; k < kb; ++k) {
        double w = 1.0;
        if ((op.maskA && k == ti) || (op.maskB && k == tj)) w = 0.5;
        if (op.c_lower && ti == tj) w = std::min(w, 0.5);
        pairs += w;
      }
    }
  return pairs * 2.0 * 128.0 * 128.0 * 128.0 * 1e-9;
}

// ---------------------------------------------------------------------------------------------------------------
// Static schedule of one large GEMM launch: tiles are assigned to a fixed number of resident workgroups by
// longest-processing-time-first on their contraction depth, so that triangular operands (depth 1..D per tile) do not
// leave most of the chip waiting for the few deepest tiles.  Built once per problem (shapes repeat every evaluation).
struct Sched {
  int* d_off = nullptr;
  unsigned* d_items = nullptr;
  int nwg = 0;   // 0: plain launch
  int tile = 0;
};

static void tile_krange(const GemmOp& op, int ti, int tj, int* ka, int* kb) {
  *ka = op.k0; *kb = op.k1;
  if (op.klim == 1) *kb = std::min(*kb, tj + 1);
  if (op.klim == 2) *ka = std::max(*ka, tj);
  if (op.klim == 3) *kb = std::min(*kb, ti + 1);
  if (op.klim == 4) *ka = std::max(*ka, ti);
}

// ops in NB units; returns host arrays for `tile`
static bool build_sched(const GemmLaunch& g, int tile, std::vector<int>* off, std::vector<unsigned>* items, int* nwg_out) {
  const int f = NB / tile;
  struct It { double w; unsigned code; };
  std::vector<It> its;
  for (int oi = 0; oi < g.nops; ++oi) {
    GemmOp op = g.op[oi];
    op.ci0 *= f; op.cj0 *= f; op.mi *= f; op.nj *= f; op.k0 *= f; op.k1 *= f;
    for (int i = 0; i < op.mi; ++i)
      for (int j = 0; j < (op.c_lower ? i + 1 : op.nj); ++j) {
        int ka, kb;
        tile_krange(op, op.ci0 + i, op.cj0 + j, &ka, &kb);
        its.push_back({(double)(kb - ka) + 0.75, ((unsigned)oi << 30) | ((unsigned)i << 16) | (unsigned)j});
      }
  }
  const int ntiles = (int)its.size();
  // resident workgroups per CU the kernel can reach (registers: 206 / 168 VGPRs for the 128- / 64-tile kernels)
  static const int occ_env = env_int("HBEGP_SCHED_OCC", 0);
  const int occmax = occ_env > 0 ? occ_env : (tile == 128 ? 2 : 3);  // residency the kernels reach (VGPRs); pinned via the LDS request
  // Above ~2 tiles per resident slot the hardware dispatcher (tiles are listed deepest-first) balances better than a
  // static list (measured: LAUUM at n=4096, 2080 tiles: 50 vs 47 TFLOP/s); below it the static list wins (TRSM 29 -> 44).
  int nwg = 0;
  static const int sched_max_tiles = env_int("HBEGP_SCHED_MAXTILES", 2048);
  if (ntiles < sched_max_tiles)
    for (int occ = occmax; occ >= 1; --occ)
      if (ntiles >= 2 * 256 * occ) { nwg = 256 * occ; break; }
  static const int sched_on = env_int("HBEGP_SCHED", 1);
  static const int sched_t32 = env_int("HBEGP_SCHED_T32", 1);
  if (!sched_on || nwg == 0 || (tile == 32 && !sched_t32)) return false;
  std::stable_sort(its.begin(), its.end(), [](const It& a, const It& b) { return a.w > b.w; });
  typedef std::pair<double, int> Load;  // (load, wg)
  std::priority_queue<Load, std::vector<Load>, std::greater<Load>> pq;
  for (int w = 0; w < nwg; ++w) pq.push({0.0, w});
  std::vector<std::vector<unsigned>> per(nwg);
  for (const It& it : its) {
    Load l = pq.top();
    pq.pop();
    per[l.second].push_back(it.code);
    pq.push({l.first + it.w, l.second});
  }
  off->assign(nwg + 1, 0);
  items->clear();
  for (int w = 0; w < nwg; ++w) {
    (*off)[w] = (int)items->size();
    items->insert(items->end(), per[w].begin(), per[w].end());
  }
  (*off)[nwg] = (int)items->size();
  *nwg_out = nwg;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int DAG_MAX_VARIANTS = 4;  // task-queue launches sized for 1..4 busy slots; with more slots the last one is shared
// ... and, behind those, launch sizes for a CROWDED device: optimiser runs of OTHER fits in flight on the same GPU (several host
// threads fitting side by side on one context -- replicas).  Sized for 6 / 12 / 24 runs in flight.
constexpr int DAG_CROWD_LEVELS = 3;
constexpr int DAG_CROWD_BUSY[DAG_CROWD_LEVELS] = {6, 12, 24};
constexpr int MAX_DEVS = 64;
static std::atomic<int> g_dev_busy[MAX_DEVS];  // optimiser runs in flight per device id, over all fits of the process

template <typename T>
struct Slot {
  int dev = 0;
  hipStream_t stream = nullptr;
  T *Xalt = nullptr, *ldalt = nullptr;  // device-driven small fit: second buffers for L^-1 / diag(L) (they ping-pong with K^-1 / alpha)
  bool x_captured = false;              // ... and after such a fit: (best_idx ? Xalt : W2) IS the captured evaluation's factor
  T *W1 = nullptr, *W2 = nullptr, *Kinv[2] = {nullptr, nullptr}, *alpha[2] = {nullptr, nullptr};
  T* W3 = nullptr;  // f32 problems only: the Cholesky factor L (lower), kept for the refinement of the panel solves
  T *ldiag = nullptr, *wbuf = nullptr;
  double *part_t = nullptr, *part_g = nullptr;
  EvalParams* dP = nullptr;
  EvalOut* dOut = nullptr;
  int* tickets = nullptr;    // [0] alpha / lml reduction, [1] gradient: "last workgroup finishes the launch's job" counters (zero between launches)
  unsigned long long* dag_trace = nullptr;  // HBEGP_DAG_TRACE: per-task time stamps of the last evaluation
  int* dag_ctrl = nullptr;   // queue head + dependency counters of the task-queue kernel (cleared before every launch)
  EvalParams* hP = nullptr;  // pinned
  EvalOut* hOut = nullptr;   // pinned
  void* small_slab = nullptr;   // one pooled device block behind alpha[2], ldiag, wbuf, part_t, part_g, dP, dOut
  size_t small_slab_bytes = 0;
  void* host_slab = nullptr;    // one pooled pinned block behind hP, hOut
  size_t host_slab_bytes = 0;
  hipGraphExec_t graph[DAG_MAX_VARIANTS + 1 + DAG_CROWD_LEVELS][2][2] = {};  // [task-queue variant][target][want_grad]
  // capture state (fit.rs:116-125)
  int best_idx = -1;  // which ping-pong buffer holds the best evaluation so far
  double best_lml = -std::numeric_limits<double>::infinity();
  int best_run = 0, best_eval = 0;
  std::vector<double> best_theta;
  std::vector<double> best_params;  // persistent fit kernel only: the clamped linear parameters the device evaluated the captured theta with
  int last_target = 0;  // buffer written by the most recent evaluation
  int dag_variant = 1;  // which ordering / launch size the next task-queue launch uses (Problem::DagVariant)
  T* dag_kinv = nullptr;  // K^-1 buffer the task-queue launch writes (its X^T X tiles); null: factorisation only
  int gemm_ord = 0;     // ordinal of the next GEMM launch inside the current evaluation (indexes the static schedules)
  unsigned long long seq = 0;  // serial number of the last evaluation handed to the device (EvalParams::seq)
  bool ctrl_cleared = false;   // the evaluation's first kernel clears dag_ctrl itself (EvalPrologue): no memset node
  bool published = false;      // the evaluation ends with publish_out_kernel: the host may spin on hOut->seq
  long long expect_ns = 0;     // how long the last published evaluation took from launch to publication (wait_eval sleeps through most of it)
};

struct ProblemBase {
  hbegp_ctx* ctx = nullptr;
  int n = 0, d = 0, np = 0, nu2 = 5, n_slots = 1;
  bool is_f32 = false;
  virtual ~ProblemBase() {}
  virtual int eval(int dev, int slot, const double* theta, const double* lo, const double* hi, double* lml, double* grad) = 0;
  virtual int time_eval(int dev, int slot, const double* theta, int reps, double* phase_ms) = 0;
  virtual int time_concurrent(int dev, const double* theta, int reps, double* out) = 0;
};
struct hbegp_problem {
  std::unique_ptr<ProblemBase> impl;
};

template <typename T>
struct Problem : ProblemBase {
  std::vector<T*> Xd, yd;                 // per device
  std::vector<std::vector<Slot<T>>> slots;  // [device][slot]
  std::vector<std::vector<Sched>> scheds;   // [device][gemm launch ordinal]
  // device-scheduled factorisation (dag_kernel.inc.hpp): one plan per problem, the same on every device
  bool dag_ = false;
  bool dag_lauum_ = false;                  // the tiles of K^-1 = X^T X are tasks of the queue too (no LAUUM launch)
  bool dag_rl_ = false;                     // right-looking plan (dag_plan.hpp build_rl): the factor L lives in W3
  double dag_gflop_lauum = 0;
  // One ordering of the same task set per number of slots that are busy at the same time (variant v = 1..n_slots): a launch
  // gets the share of the CUs that fits v concurrent launches, and its queue is ordered for that many workgroups.  Same
  // tasks, same arithmetic, same bits for every variant -- only the order in which workgroups pull them differs.  A fit whose
  // optimiser runs end at different times (early stopping, 8 runs over 3 slots) gives the remaining runs the freed CUs.
  struct DagVariant {
    int nwg = 0;
    std::vector<DagTask*> tasks;            // per device
    std::vector<DagTask> host_tasks;        // kept for the trace dump
  };
  std::vector<DagVariant> dag_var;          // [0] unused, [v] for v busy slots
  std::unique_ptr<std::atomic<int>[]> busy_slots_;  // per device: slots inside an optimiser run (0: not known -> all of them)
  int dag_ntasks = 0, dag_nwg = 0;          // dag_nwg: workgroups of the default variant (all slots busy)
  int dag_nvar_ = 1, dag_ncrowd_ = 0;       // variants [1..dag_nvar_] for this fit's own busy slots, then dag_ncrowd_ crowded-device levels
  unsigned long long dag_wait_ticks_ = 200000000ull;  // bound of one dependency wait (100 MHz ticks), see init()
  size_t dag_ctrl_bytes = 0;
  double dag_gflop = 0;
  // f32 (--use-32): the panel solve T = A21 L11^-T is a product with the explicit inverse X11, whose residual grows with
  // cond(L11) * eps -- harmless in f64, but in f32 it makes the lml gradient ~8x less accurate than LAPACK's substitution.
  // One step of iterative refinement against the factor itself (kept in W3) restores a small residual:
  //   T = A21 X11^T;  A21 -= T L11^T;  T += A21 X11^T          (three products of the same shape, all fp32 MFMA)
  bool refine_ = false;
  bool dry_ = false;                        // walk the evaluation without launching (schedule construction)
  bool adhoc_ = false;                      // GEMM launches bypass the per-evaluation schedule table
  bool small_ = false;                      // np = 128, d <= 32: one evaluation = ONE launch (small_eval_kernel), everything in the LDS
  bool like_fit_ = false;                   // path selection of a fit (task queue from DAG_MIN_BLOCKS_FIT blocks on) although there is one slot
  // HBEGP_HOSTIO (default 1): an evaluation is driven through the slot's pinned blocks -- the first kernel reads the parameters
  // there and prepares the device-side blocks (EvalPrologue), the last one copies the results back and publishes the
  // evaluation's serial number, which the host thread spins on.  0: parameter copy + reset kernel + memset in front, a result
  // copy behind, hipStreamSynchronize (round 1-3).  Measured on config M (three optimiser runs, rocprofv3 kernel trace): the GPU
  // waited 52 + 54 + 29 us per evaluation for the host to enqueue the three extra nodes in front of the long kernel, and 84 us
  // between two evaluations.
  bool hostio_ = true;
  // ONE rule for "this problem's evaluations end by publishing their serial number into the pinned result block": enqueue_eval
  // acts on it when it runs, run_eval when a graph replay skips enqueue_eval (two copies of the condition could drift apart:
  // every evaluation would then sit out wait_eval's two-second fallback, or silently fall back to hipStreamSynchronize)
  bool eval_published() const { return hostio_ && !small_; }
  int leaf_dbg_ = 0;                        // HBEGP_LEAF_DBG: debug bits of the diagonal-block kernel (16: helper waves start late)

  // Small device arrays of the problem (features, targets, task queues, schedule tables, control words) come from the block
  // pool too and go back to it in release(): hipMalloc / hipFree synchronise the whole device, which serialises host threads
  // that fit side by side on one context (and cost ~2 ms of a 10 ms fit even alone).
  struct Pooled { int dev; void* p; size_t bytes; };
  std::vector<Pooled> pooled_;
  template <typename U>
  U* palloc(int dev, size_t count) {
    bool fresh = false;
    const size_t bytes = std::max<size_t>(16, sizeof(U) * count);
    void* q = g_pool.get(dev, bytes, &fresh);
    pooled_.push_back({dev, q, bytes});
    return static_cast<U*>(q);
  }

  // single_shot: the problem runs one evaluation (extend): skip the static schedule tables, every GEMM launch is ad hoc
  // like_fit: choose the evaluation path (launches / task queue) as a fit of this size does, whatever the slot count --
  // `extend` then repeats the fit's own evaluation of a theta bit for bit
  Problem(hbegp_ctx* c, const T* X, const T* y, int n_, int d_, double nu, int n_slots_, bool single_shot = false, bool like_fit = false) {
    try {
      adhoc_ = single_shot;
      like_fit_ = like_fit;
      init(c, X, y, n_, d_, nu, n_slots_);
    } catch (...) {
      release();  // a constructor that throws never runs the destructor: give back what was allocated so far
      throw;
    }
  }

  void init(hbegp_ctx* c, const T* X, const T* y, int n_, int d_, double nu, int n_slots_) {
    ctx = c; n = n_; d = d_; np = round_up(n_, NB); n_slots = n_slots_;
    nu2 = std::isinf(nu) ? 0 : (int)std::lround(2 * nu);  // 0 = squared exponential (nu = infinity)
    is_f32 = sizeof(T) == 4;
    // off by default: with the chunked fp64 totals of the f32 tile GEMM (kernels.hip) the plain recursion is already 2-7x
    // closer to the f64 result than LAPACK's f32 path (n=2048, cond 7e4: gradient 3e-6 vs 2e-5); refinement buys another
    // 1.3x on alpha / K^-1 for 26 % more time (measured), for kernel matrices beyond cond ~1e5
    refine_ = is_f32 && env_int("HBEGP_F32_REFINE", 0) != 0;
    leaf_dbg_ = env_int("HBEGP_LEAF_DBG", 0) & 16;  // tests only; the bits that skip work stay with tools/leaf_bench
    // The reference's own regime (minimize.rs:118-120: n stays at 100-200): up to 128 rows the five launches of the general path
    // cost more in launch gaps and HBM round trips than in arithmetic; one workgroup does the whole evaluation in its LDS
    // instead (HBEGP_SMALL=0: the general path, which the tests compare it with).
    small_ = np == NB && d <= SMALL_EVAL_MAXD && !refine_ && env_int("HBEGP_SMALL", 1) != 0;
    hostio_ = env_int("HBEGP_HOSTIO", 1) != 0;
    const size_t nn = (size_t)np * np;
    Xd.assign(c->devs.size(), nullptr);
    yd.assign(c->devs.size(), nullptr);
    slots.resize(c->devs.size());
    for (size_t di = 0; di < c->devs.size(); ++di) {
      HIPCHECK(hipSetDevice(c->devs[di]));
      slots[di].resize(n_slots);
      for (auto& s : slots[di]) {
        s.dev = c->devs[di];
        s.stream = g_stream_pool.get(s.dev);
      }
      // everything this constructor queues goes to the first slot's (non-blocking) stream and is waited for on THAT stream:
      // no null-stream operation, no device-wide synchronisation (other host threads may be fitting on this device)
      hipStream_t st0 = slots[di][0].stream;
      Xd[di] = palloc<T>(c->devs[di], (size_t)n * d);
      yd[di] = palloc<T>(c->devs[di], np);
      HIPCHECK(hipMemsetAsync(yd[di], 0, sizeof(T) * np, st0));
      HIPCHECK(hipMemcpyAsync(Xd[di], X, sizeof(T) * (size_t)n * d, hipMemcpyHostToDevice, st0));
      HIPCHECK(hipMemcpyAsync(yd[di], y, sizeof(T) * n, hipMemcpyHostToDevice, st0));
      for (auto& s : slots[di]) {
        bool fresh1 = false, fresh2 = false, fk = false;
        s.W1 = static_cast<T*>(g_pool.get(s.dev, sizeof(T) * nn, &fresh1));
        s.W2 = static_cast<T*>(g_pool.get(s.dev, sizeof(T) * nn, &fresh2));
        for (int b = 0; b < 2; ++b) s.Kinv[b] = static_cast<T*>(g_pool.get(s.dev, sizeof(T) * nn, &fk));
        // W2 carries the triangular operand X = L^-1.  The GEMM loader does not mask: the strict upper triangle of W2 must
        // BE zero (tiles on the diagonal are loaded whole).  Nothing in the engine writes there, so clearing the buffer
        // once per slot is enough -- also when it is recycled from the pool (it may have held a full symmetric K^-1).
        // W1's strict upper part is only ever multiplied by those zeros or ignored: it just has to be finite.
        HIPCHECK(hipMemsetAsync(s.W2, 0, sizeof(T) * nn, st0));
        if (refine_) {
          bool f3 = false;
          s.W3 = static_cast<T*>(g_pool.get(s.dev, sizeof(T) * nn, &f3));
          HIPCHECK(hipMemsetAsync(s.W3, 0, sizeof(T) * nn, st0));  // strict upper triangle must be zero, like W2's
        }
        (void)fresh1; (void)fresh2; (void)fk;  // fresh blocks were cleared by the pool
        // the small per-slot arrays: one pooled block
        {
          auto up = [](size_t v) { return (v + 255) / 256 * 256; };
          const size_t b_vec = up(sizeof(T) * np);
          const size_t b_part_t = up(sizeof(double) * ((size_t)((np + 255) / 256) * np + 2 * (size_t)((np + 255) / 256) + 64));
          const size_t b_part_g = up(sizeof(double) * gradtrace_part_elems(np, d));
          const size_t b_p = up(sizeof(EvalParams)), b_o = up(sizeof(EvalOut));
          s.small_slab_bytes = 4 * b_vec + b_part_t + b_part_g + b_p + b_o + 256;
          bool fs = false;
          s.small_slab = g_pool.get(s.dev, s.small_slab_bytes, &fs);
          char* q = static_cast<char*>(s.small_slab);
          s.alpha[0] = reinterpret_cast<T*>(q); q += b_vec;
          s.alpha[1] = reinterpret_cast<T*>(q); q += b_vec;
          s.ldiag = reinterpret_cast<T*>(q); q += b_vec;
          s.wbuf = reinterpret_cast<T*>(q); q += b_vec;
          s.part_t = reinterpret_cast<double*>(q); q += b_part_t;
          s.part_g = reinterpret_cast<double*>(q); q += b_part_g;
          s.dP = reinterpret_cast<EvalParams*>(q); q += b_p;
          s.dOut = reinterpret_cast<EvalOut*>(q); q += b_o;
          s.tickets = reinterpret_cast<int*>(q);
          HIPCHECK(hipMemsetAsync(s.dOut, 0, b_o + 256, st0));  // the result block and, right behind it, the tickets: one fill
          const size_t h_p = up(sizeof(EvalParams));
          s.host_slab_bytes = h_p + up(sizeof(EvalOut));
          s.host_slab = g_host_pool.get(s.host_slab_bytes);
          s.hP = reinterpret_cast<EvalParams*>(s.host_slab);
          s.hOut = reinterpret_cast<EvalOut*>(static_cast<char*>(s.host_slab) + h_p);
          memset(s.hP, 0, sizeof(EvalParams));
          memset(s.hOut, 0, sizeof(EvalOut));
        }
      }
      HIPCHECK(hipStreamSynchronize(st0));
    }
    // Task queue of the factorisation.  The workgroups of one launch hold a CU each while they wait for the diagonal
    // blocks, so a problem whose slots run concurrently shares the CUs between its slots.
    // Default: on for problems whose slots run concurrently (a fit: the optimiser runs share the chip, and a resident
    // task-queue kernel keeps its CUs while another run's tile GEMMs fill the rest: measured 1.30 -> 1.56 fit+predict/s
    // on config M), and for any problem of n > 4608, where the look-ahead hides the chain behind the bulk tiles (one
    // evaluation alone, launches vs task queue: n=4096 2.92 vs 2.98 ms, 6144 7.10 vs 5.90, 8192 12.7 vs 10.6, 16384 86.3 vs
    // 72.1); below that a single evaluation stream is a few per cent faster as a chain of launches.
    // HBEGP_DAG=0/1 forces it (read per problem: the parity tests flip it inside one process).
    const int dag_env = env_int("HBEGP_DAG", -1);
    // measured (config M data, launches vs task queue): one evaluation alone n=1536: 0.75 / 0.78 ms, 2048: 1.05 / 1.02, 4096: 2.89 / 2.19,
    // 8192: 12.7 / 9.9; three concurrent optimiser runs (fits/s) n=512: 21.1 / 19.6, 1024: 11.35 / 11.48, 1536: 6.80 / 7.95, 4096: 1.29 / 1.58
    // round 4 (faster diagonal block, evaluations driven through pinned memory): three runs side by side, fits/s, launches / task
    // queue: n=1024: 14.4 / 13.4, 1536: 7.95 / 8.7, 2048: 5.3 / 6.0 -- the queue from 12 blocks on (round 3: 8)
    // with the row-progressive plan (dag_plan.hpp rl_progressive): n=512: 29.8 / 28.6, 768: 19.9 / 20.7, 896: 17.0 / 18.3, 1024: 14.5 / 15.6,
    // 1280: 10.3 / 12.3; one evaluation alone (ms): n=512 0.202 / 0.218, 768 0.301 / 0.302, 896 0.357 / 0.347, 1024 0.406 / 0.385,
    // 1536 0.650 / 0.572 -- the queue from 6 blocks on, for fits and for single evaluations alike
    const int dag_min_blocks = env_int("HBEGP_DAG_MIN_BLOCKS", DAG_MIN_BLOCKS_FIT);
    dag_ = (dag_env < 0 ? np / NB >= dag_min_blocks : dag_env != 0) && !adhoc_ && np / NB >= 2;
    if (refine_) dag_ = false;  // the refined panel solve exists as launches only (the task queue carries the f64 recursion)
    if (dag_) {
      int cus = 1 << 30;  // the launch sizes are shared by the devices of the context: size them for the smallest one
      for (int dev : c->devs) {
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, dev));
        cus = std::min(cus, std::max(1, prop.multiProcessorCount));
      }
      const int forced = env_int("HBEGP_DAG_WG", 0);
      // Concurrent slots: each launch gets a little more than its share of the CUs (a multiple of 8: the dispatcher deals
      // workgroups round-robin to the 8 XCCs).  The surplus workgroups of a launch start on the CUs another slot's launch has
      // just given back (that slot is in its short kmat / alpha / gradient launches) and leave when their own queue is empty.
      // Measured, 3 slots, n=4096, fit+predict/s: 80 -> 1.58, 85 -> 1.62, 88 -> 1.64, 96 -> 1.68, 104 -> 1.64, 112 -> 1.68,
      // 120 -> 1.54, 128 -> 1.60, 170 -> 1.43, 256 -> 1.14.
      auto share_of = [&](int busy) {
        const int share = busy <= 1 ? cus : std::max(8, (cus * env_int("HBEGP_DAG_OVERSUB", 112) / 100 / busy + 4) / 8 * 8);  // 3 slots: 96
        return forced > 0 ? forced : std::max(1, std::min(cus, share));
      };
      // plans depend only on (blocks, stage depth, tiling and ordering knobs): the caller fits one model per generation with
      // slowly growing n, so they are kept (building + simulating the n=4096 queue costs ~15 ms of host time per fit)
      // HBEGP_DAG_LAUUM (default 1): the tiles of K^-1 = X^T X follow the recursion in the same queue, as 128x64 tile tasks,
      // instead of a gemm_kernel launch behind the task-queue launch
      dag_lauum_ = env_int("HBEGP_DAG_LAUUM", 1) != 0;
      // HBEGP_DAG_RL: right-looking tile Cholesky + divide-and-conquer inverse instead of the recursion that carries the
      // inverse: two 128-deep tiles between consecutive diagonal blocks instead of products as deep as the node is wide
      // (critical path of one evaluation at n = 4096: 2.83 -> 2.02 ms, simulated).  Not bitwise equal to the launch path
      // (another order of operations); the recursion plan stays available (0) and is what the bitwise tests pin.
      // Above ~10k rows one evaluation is bound by the tile work, where the recursion's deeper tiles win again (n=8192: 10.3 /
      // 9.9 ms recursion / right-looking, 12288: 31.4 / 32.0, 16384: 72.3 / 75.5).
      dag_rl_ = env_int("HBEGP_DAG_RL", np / NB <= 80 ? 1 : 0) != 0;
      static std::mutex cache_mu;
      static std::map<std::array<int, 20>, std::shared_ptr<const DagPlan>> cache;
      auto plan_for = [&](int nwg) {
        std::array<int, 20> key = {np / NB, dag_stage_depth(is_f32), env_int("HBEGP_DAG_SMALLH", dag_rl_ ? 4 : 8), env_int("HBEGP_DAG_ORDER", 1) ? env_int("HBEGP_DAG_ORDER_WG", nwg) : 0,
                                   env_int("HBEGP_DAG_FINE", 1), env_int("HBEGP_DAG_CRIT", 1), 0, dag_lauum_ ? 1 : 0, dag_rl_ ? 1 : 0,
                                   env_int("HBEGP_DAG_RL_GROUP", 32), env_int("HBEGP_DAG_RL_NEAR", 1),
                                   env_int("HBEGP_DAG_LAUUM_SPLIT", n_slots <= 1 ? 1 : 0),
                                   env_int("HBEGP_DAG_CHAIN32", 1),
                                   // row-progressive inverse and K^-1 (dag_plan.hpp rl_progressive) up to 20 blocks.  Measured (divide and
                                   // conquer / progressive; one evaluation alone in ms, three-run fits per s): n=1536 0.65/0.65, 9.0/10.1;
                                   // 2048 0.88/0.80, 6.4/7.1; 2560 1.12/1.05, 4.5/4.7; 3072 1.41/1.30, 3.37/3.21; 3584 1.75/1.67, 2.38/2.21;
                                   // 4096 2.08/2.08, 1.67/1.55; C5 (n=2048 f32): 0.85/0.75 ms, 8.9/12.5 fits/s.  Above ~22 blocks the bulk
                                   // tiles fill every CU and the chain's tasks wait for a free workgroup; a function of n alone, so that
                                   // `extend` repeats a fit's evaluation bit for bit.
                                   env_int("HBEGP_DAG_PROG", np / NB <= DAG_PROG_MAX_BLOCKS ? 1 : 0),
                                   env_int("HBEGP_DAG_PROG_UNEAR", -1), env_int("HBEGP_DAG_PROG_KNEAR", -1), env_int("HBEGP_DAG_PROG_SMALL", 0),
                                   0, 0,  // (round 4: geometric ranges, chain tasks moved forward in the queue -- measured, no gain, removed)
                                   // 128x128 tiles for the deep products without beta = 1 when several slots share the chip (throughput: half the
                                   // tasks, fewer fragment reads per MFMA): three-run fits M f64 1.727 -> 1.740, M f32 2.64 -> 2.77, C4 0.233 ->
                                   // 0.245; one evaluation alone gets SLOWER (fewer, longer tasks on 256 workgroups: n=4096 2.08 -> 2.22 ms), so
                                   // single-slot problems keep 128x64.  The bits do not depend on the tile shape (one k-ascending chain of MFMA
                                   // accumulations per element), so `extend` still repeats a fit's evaluation bit for bit.
                                   // Below 32 blocks the fits lose (n=1536 10.4 -> 8.5, 2048 7.0 -> 6.4, 3072 3.36 -> 3.13 fits/s: too few deep
                                   // products, the longer tasks only unbalance the end of the launch).
                                   // 2: also the tiles with beta = 1 (the trailing updates; the old values are then fetched in the epilogue): M f64
                                   // 1.711 / 1.737 / 1.739 -> 1.753 / 1.754 / 1.756 (alternating), M f32 2.76 -> 2.80, C4 0.242 -> 0.249 fits/s.
                                   // One evaluation alone: n=6144 4.58 -> 4.73 ms (worse), n=8192 9.69 -> 9.31 ms: from 64 blocks on there too.
                                   env_int("HBEGP_DAG_BIG128", ((n_slots >= 2 && np / NB >= 32) || np / NB >= 64) ? 2 : 0)};  // one evaluation alone: 2.21 -> 2.17 ms at n=4096, 1.02 -> 0.97 at 2048; a fit: 1.67 -> 1.66
        std::shared_ptr<const DagPlan> cached;
        {
          std::lock_guard<std::mutex> lk(cache_mu);
          auto it = cache.find(key);
          if (it != cache.end()) cached = it->second;
        }
        if (!cached) {
          DagBuilder builder(key[1], key[2], key[3], key[4] != 0, key[5]);
          builder.set_rl(key[9], key[10], key[11] != 0, key[12] != 0);
          builder.set_rl_progressive(key[13] != 0, key[14], key[15], key[16] != 0);
          builder.set_big128(key[19] != 0, key[19] >= 2);
          cached = std::make_shared<const DagPlan>(builder.build(0, np / NB, dag_lauum_, dag_rl_));
          std::lock_guard<std::mutex> lk(cache_mu);
          if (cache.size() > 64) cache.clear();
          cache[key] = cached;
        }
        return cached;
      };
      const int nvar = std::min(n_slots, DAG_MAX_VARIANTS);
      dag_nwg = share_of(n_slots);  // the default variant: every slot busy
      std::shared_ptr<const DagPlan> cached = plan_for(dag_nwg);
      if (cached->tasks.empty() && dag_rl_) {  // too many counters for 16-bit ids (n > ~12k): the recursion plan needs far fewer
        dag_rl_ = false;
        cached = plan_for(dag_nwg);
      }
      const DagPlan& plan = *cached;
      if (plan.tasks.empty()) dag_ = dag_lauum_ = false;  // too many counters for 16-bit ids (n > 32k): launch-per-product path
      if (dag_ && env_int("HBEGP_DAG_VALIDATE", 0)) {
        const std::string why = dag_plan_validate(plan, np / NB);
        if (!why.empty()) throw std::runtime_error("task queue of the factorisation is unsound: " + why);
      }
      if (dag_ && env_int("HBEGP_DAG_VERBOSE", 0))
        fprintf(stderr, "dag plan: %zu tasks (%d K^-1 tiles), %zu counters, %d workgroups, critical path %.0f us, simulated %.0f us, %.2f GFLOP (%.2f in K^-1)\n",
                plan.tasks.size(), plan.n_lauum, plan.totals.size(), dag_nwg, plan.crit_us, plan.sim_us, plan.gflop, plan.gflop_lauum);
      if (dag_) {
        dag_ntasks = (int)plan.tasks.size();
        dag_nwg = std::min(dag_nwg, dag_ntasks);
        dag_gflop = plan.gflop;
        dag_gflop_lauum = plan.gflop_lauum;
        dag_ctrl_bytes = (sizeof(int) * (DAG_CTRL_WORDS + plan.totals.size()) + 15) / 16 * 16;
        // A dependency wait longer than this is reported as a scheduling bug (info = -2).  The clock runs on while the queue is
        // preempted or time-sliced (another process, a profiler serialising dispatches) and single waits grow with the plan, so
        // the bound follows the plan: 200 x its simulated makespan with every slot sharing the chip, at least 2 s.
        {
          const double wait_s = std::max(2.0, 200.0 * plan.sim_us * 1e-6 * std::max(1, n_slots));
          const double forced_s = getenv("HBEGP_DAG_WAIT_S") ? atof(getenv("HBEGP_DAG_WAIT_S")) : 0.0;
          dag_wait_ticks_ = (unsigned long long)((forced_s > 0 ? forced_s : wait_s) * 1e8);
        }
        busy_slots_.reset(new std::atomic<int>[c->devs.size()]);
        for (size_t di = 0; di < c->devs.size(); ++di) busy_slots_[di].store(0);
        // the variants: [nvar] = the default (every slot busy), [v < nvar] for v busy slots (HBEGP_DAG_ADAPT=0: default only)
        const bool adapt = env_int("HBEGP_DAG_ADAPT", 1) != 0 && forced <= 0;
        // [nvar + 1 + l]: a crowded device, DAG_CROWD_BUSY[l] runs in flight over all fits (only for problems with several slots: a fit;
        // same tasks, same bits -- fewer workgroups per launch, so that all the launches in flight are resident)
        dag_nvar_ = nvar;
        dag_ncrowd_ = (adapt && n_slots > 1) ? DAG_CROWD_LEVELS : 0;
        dag_var.assign(nvar + 1 + dag_ncrowd_, DagVariant());
        for (int v = 1; v <= nvar + dag_ncrowd_; ++v) {
          DagVariant& var = dag_var[v];
          std::shared_ptr<const DagPlan> pv = cached;
          var.nwg = dag_nwg;
          if (v != nvar && adapt) {
            const int busy = v < nvar ? v : DAG_CROWD_BUSY[v - nvar - 1];
            var.nwg = std::min(share_of(busy), dag_ntasks);
            if (v > nvar) var.nwg = std::min(var.nwg, dag_nwg);
            pv = plan_for(var.nwg);
            if (pv->tasks.size() != plan.tasks.size()) { pv = cached; var.nwg = dag_nwg; }  // cannot happen: same task set
          }
          var.host_tasks = pv->tasks;
          var.tasks.assign(c->devs.size(), nullptr);
          for (size_t di = 0; di < c->devs.size(); ++di) {
            HIPCHECK(hipSetDevice(c->devs[di]));
            var.tasks[di] = palloc<DagTask>(c->devs[di], pv->tasks.size());
            HIPCHECK(hipMemcpyAsync(var.tasks[di], var.host_tasks.data(), sizeof(DagTask) * var.host_tasks.size(), hipMemcpyHostToDevice, slots[di][0].stream));  // (the source outlives the copy: it is the problem's own)
          }
        }
        for (size_t di = 0; di < c->devs.size(); ++di) {
          HIPCHECK(hipSetDevice(c->devs[di]));
          for (auto& s : slots[di]) {
            if (dag_rl_ && !s.W3) {
              bool f3 = false;
              s.W3 = static_cast<T*>(g_pool.get(s.dev, sizeof(T) * nn, &f3));  // the factor L: every tile read has been written
            }
            s.dag_ctrl = palloc<int>(s.dev, dag_ctrl_bytes / sizeof(int));
            HIPCHECK(hipMemsetAsync(s.dag_ctrl, 0, dag_ctrl_bytes, slots[di][0].stream));
            if (getenv("HBEGP_DAG_TRACE")) {
              s.dag_trace = palloc<unsigned long long>(s.dev, 5 * plan.tasks.size());
              HIPCHECK(hipMemsetAsync(s.dag_trace, 0, sizeof(unsigned long long) * 5 * plan.tasks.size(), slots[di][0].stream));
            }
          }
          HIPCHECK(hipStreamSynchronize(slots[di][0].stream));  // the copies and fills above; the other slots' streams do not wait for this one
        }
      }
    }
    // build the static GEMM schedules (per device; shared by its slots) by walking one evaluation without launching
    scheds.resize(c->devs.size());
    if (adhoc_) return;
    dry_ = true;
    for (size_t di = 0; di < c->devs.size(); ++di) {
      HIPCHECK(hipSetDevice(c->devs[di]));
      enqueue_eval(slots[di][0], di, 0, true, nullptr);
    }
    dry_ = false;
  }
  ~Problem() override { release(); }

  void release() {
    for (size_t di = 0; di < slots.size(); ++di) {
      (void)hipSetDevice(ctx->devs[di]);
      for (auto& s : slots[di]) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        for (int v = 0; v <= DAG_MAX_VARIANTS + DAG_CROWD_LEVELS; ++v)
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
              if (s.graph[v][a][b]) (void)hipGraphExecDestroy(s.graph[v][a][b]);
        const size_t nnb = sizeof(T) * (size_t)np * np;
        g_pool.put(s.dev, s.W1, nnb); g_pool.put(s.dev, s.W2, nnb); g_pool.put(s.dev, s.W3, nnb);
        for (int b = 0; b < 2; ++b) g_pool.put(s.dev, s.Kinv[b], nnb);
        g_pool.put(s.dev, s.small_slab, s.small_slab_bytes);
        g_host_pool.put(s.host_slab, s.host_slab_bytes);
        g_stream_pool.put(s.dev, s.stream);  // synchronised above
      }
    }
    for (const Pooled& q : pooled_) g_pool.put(q.dev, q.p, q.bytes);  // features, targets, task queues, control words, schedule tables
    pooled_.clear();
    slots.clear();
    scheds.clear();
    dag_var.clear();
    Xd.clear();
    yd.clear();
  }

  void gemm(Slot<T>& s, size_t di, GemmLaunch& g, PhaseTimer* tm, int kind) {
    hipStream_t stream = s.stream;
    g.info = &s.dOut->info;
    const int ord = s.gemm_ord++;
    if (dry_) {
      int tiles = 0;
      for (int i = 0; i < g.nops; ++i) {
        const GemmOp& op = g.op[i];
        tiles += op.c_lower ? op.mi * (op.mi + 1) / 2 : op.mi * op.nj;
      }
      Sched sc;
      sc.tile = pick_tile(tiles);
      std::vector<int> off;
      std::vector<unsigned> items;
      int nwg = 0;
      if (build_sched(g, sc.tile, &off, &items, &nwg)) {
        sc.nwg = nwg;
        sc.d_off = palloc<int>(s.dev, off.size());
        sc.d_items = palloc<unsigned>(s.dev, items.size());
        HIPCHECK(hipMemcpyAsync(sc.d_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice, stream));
        HIPCHECK(hipMemcpyAsync(sc.d_items, items.data(), sizeof(unsigned) * items.size(), hipMemcpyHostToDevice, stream));
        HIPCHECK(hipStreamSynchronize(stream));  // off / items are locals
      }
      if ((int)scheds[di].size() <= ord) scheds[di].resize(ord + 1);
      scheds[di][ord] = sc;
      return;
    }
    if (adhoc_) {  // launch sequences other than the evaluation's (incremental extend): hardware dispatch, no table
      g.sched_off = nullptr; g.sched_items = nullptr; g.sched_nwg = 0;
      int tiles = 0;
      for (int i = 0; i < g.nops; ++i) tiles += g.op[i].c_lower ? g.op[i].mi * (g.op[i].mi + 1) / 2 : g.op[i].mi * g.op[i].nj;
      launch_gemm<T>(g, pick_tile(tiles), stream);
      return;
    }
    const Sched& sc = scheds[di][ord];
    g.sched_off = sc.nwg ? sc.d_off : nullptr;
    g.sched_items = sc.nwg ? sc.d_items : nullptr;
    g.sched_nwg = sc.nwg;
    double gf = 0;
    if (tm)
      for (int i = 0; i < g.nops; ++i) gf += op_gflop(g.op[i]);
    if (tm) tm->begin(kind, sc.tile, gf);
    launch_gemm<T>(g, sc.tile, stream);
    if (tm) tm->end();
  }

  // Cholesky + inverse of the factor on the diagonal block range [lo, hi) (units of 128): on return W2 holds
  // X = L^-1 on that range (lower), ldiag the diagonal of L.
  void chol_inv_rec(Slot<T>& s, size_t di, int lo, int hi, PhaseTimer* tm) {
    if (hi - lo == 1) {
      if (dry_) return;
      if (tm) tm->begin(PhaseTimer::LEAF);
      launch_leaf<T>(s.W1, s.W2, np, lo, s.ldiag, &s.dOut->info, s.stream, leaf_dbg_, refine_ ? s.W3 : nullptr);
      if (tm) tm->end();
      return;
    }
    chol_inv_split(s, di, lo, lo + (hi - lo) / 2, hi, tm, false);
  }

  // One node of the recursion with an explicit split point.  left_done: X (and L's diagonal) of [lo, mid) are already in
  // place (incremental extend), only the border and the right part are computed.
  void chol_inv_split(Slot<T>& s, size_t di, int lo, int mid, int hi, PhaseTimer* tm, bool left_done) {
    if (!left_done) chol_inv_rec(s, di, lo, mid, tm);
    GemmOp base{};
    base.lda = base.ldb = base.ldc = np;
    T* Tbuf = refine_ ? s.W3 : s.W2;  // where T = L21 lives: f32 keeps it (it IS the factor's block), f64 lets X21 overwrite it
    {
      // T = A21 * X11^T  -> Tbuf[2,1]      (TRSM of potrf as a product with the explicit inverse)
      GemmLaunch g{};
      g.nops = 1;
      GemmOp& op = g.op[0];
      op = base;
      op.A = s.W1; op.B = s.W2; op.C = Tbuf;
      op.a_kmajor = 0; op.b_kmajor = 0;
      op.ci0 = mid; op.mi = hi - mid; op.cj0 = lo; op.nj = mid - lo;
      op.k0 = lo; op.k1 = mid; op.klim = 1; op.maskB = 1;
      gemm(s, di, g, tm, PhaseTimer::GEMM);
      if (refine_ && !left_done) {
        // residual in place: A21 -= T * L11^T   (L11 = W3[lo:mid, lo:mid], lower), then the correction T += A21 * X11^T
        GemmLaunch gr{};
        gr.nops = 1;
        GemmOp& r = gr.op[0];
        r = base;
        r.A = s.W3; r.B = s.W3; r.C = s.W1;
        r.ci0 = mid; r.mi = hi - mid; r.cj0 = lo; r.nj = mid - lo;
        r.k0 = lo; r.k1 = mid; r.klim = 1; r.maskB = 1; r.alpha_neg = 1; r.beta_one = 1;
        gemm(s, di, gr, tm, PhaseTimer::GEMM);
        GemmLaunch gc{};
        gc.nops = 1;
        GemmOp& cc = gc.op[0];
        cc = op;
        cc.beta_one = 1;
        gemm(s, di, gc, tm, PhaseTimer::GEMM);
      }
    }
    {
      // A22 -= T T^T (lower)   and   U = T * X11 -> W1[2,1]   (independent: one launch, one static schedule).
      // Measured alternatives at n=4096: two launches 3.71 ms/evaluation, U forked onto a second stream inside the graph
      // 3.57 ms but 0.76 fit+predict/s in the 3-run bench (multi-branch graphs serialise badly); merged 3.38 ms, 1.10.
      GemmLaunch g{};
      g.nops = 2;
      GemmOp& syrk = g.op[0];
      syrk = base;
      syrk.A = Tbuf; syrk.B = Tbuf; syrk.C = s.W1;
      syrk.ci0 = mid; syrk.cj0 = mid; syrk.mi = hi - mid; syrk.nj = hi - mid; syrk.c_lower = 1;
      syrk.k0 = lo; syrk.k1 = mid; syrk.alpha_neg = 1; syrk.beta_one = 1;
      GemmOp& u = g.op[1];
      u = base;
      u.A = Tbuf; u.B = s.W2; u.C = s.W1;
      u.a_kmajor = 0; u.b_kmajor = 1;
      u.ci0 = mid; u.mi = hi - mid; u.cj0 = lo; u.nj = mid - lo;
      u.k0 = lo; u.k1 = mid; u.klim = 2; u.maskB = 1;
      gemm(s, di, g, tm, PhaseTimer::GEMM);
    }
    chol_inv_rec(s, di, mid, hi, tm);
    {
      // X21 = -X22 * U -> W2[2,1]
      GemmLaunch g{};
      g.nops = 1;
      GemmOp& op = g.op[0];
      op = base;
      op.A = s.W2; op.B = s.W1; op.C = s.W2;
      op.a_kmajor = 0; op.b_kmajor = 1;
      op.ci0 = mid; op.mi = hi - mid; op.cj0 = lo; op.nj = mid - lo;
      op.k0 = mid; op.k1 = hi; op.klim = 3; op.maskA = 1; op.alpha_neg = 1;
      gemm(s, di, g, tm, PhaseTimer::GEMM);
    }
  }

  // Diagnostics of a wait that exceeded its bound (a scheduling bug, never a data property): which task gave up, and
  // where its counters stood.
  void dag_report_timeout(Slot<T>& s) {
    if (!s.dag_ctrl) return;
    std::vector<int> ctrl(dag_ctrl_bytes / sizeof(int));
    if (hipMemcpyAsync(ctrl.data(), s.dag_ctrl, dag_ctrl_bytes, hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) return;
    const int row = ctrl[1] - 1;
    fprintf(stderr, "task queue timeout: queue head %d of %d, first task that gave up: %d\n", ctrl[0], dag_ntasks, row);
    const std::vector<DagTask>& host_tasks = dag_var[s.dag_variant].host_tasks;
    if (row >= 0 && row < (int)host_tasks.size()) {
      const DagTask& t = host_tasks[row];
      fprintf(stderr, "  kind %d flags %x row0 %d col0 %d k [%d, %d) waits:", t.kind, t.flags, t.row0, t.col0, t.kbeg, t.kend);
      for (int w = 0; w < t.nwait; ++w) fprintf(stderr, " c%d=%d/%d", t.wcnt[w], ctrl[DAG_CTRL_WORDS + t.wcnt[w]], t.wval[w]);
      fprintf(stderr, "\n");
    }
    (void)hipGetLastError();
  }

  // Top of the factorisation.  Above `big` (in 128-blocks) the binary recursion would issue a chain of mid-size,
  // poorly filled launches; instead the big blocks are swept right-looking (one TRSM and one SYRK per big block, each
  // covering everything below / behind it) and the off-diagonal blocks of X = L^-1 are formed afterwards level by level,
  // all nodes of a level in one launch (they are independent: TRTRI has no dependency along the diagonal).
  void chol_inv(Slot<T>& s, size_t di, int nb, PhaseTimer* tm) {
    if (dag_ && !adhoc_) {
      // the whole recursion in ONE persistent launch: workgroups pull diagonal-block and tile tasks from an ordered queue
      if (dry_) return;
      if (!s.ctrl_cleared) HIPCHECK(hipMemsetAsync(s.dag_ctrl, 0, dag_ctrl_bytes, s.stream));
      s.ctrl_cleared = false;
      DagLaunch g{};
      const DagVariant& var = dag_var[s.dag_variant];
      g.tasks = var.tasks[di]; g.ntasks = dag_ntasks; g.ctrl = s.dag_ctrl;
      g.W1 = s.W1; g.W2 = s.W2; g.ld = np; g.ldiag = s.ldiag; g.info = &s.dOut->info;
      g.W3 = s.W3;
      g.Kinv = dag_lauum_ ? s.dag_kinv : nullptr;
      g.trace = s.dag_trace;
      g.wait_ticks = dag_wait_ticks_;
      g.leaf_dbg = leaf_dbg_;
      if (tm) tm->begin(PhaseTimer::DAG, 0, g.Kinv ? dag_gflop : dag_gflop - dag_gflop_lauum);
      launch_dag<T>(g, var.nwg, s.stream);
      if (tm) tm->end();
      return;
    }
    // (round 1: a right-looking sweep over 512- / 1024-wide big blocks in front of the recursion -- 3.55 vs 3.43 ms per evaluation at
    // n = 4096, removed in round 5)
    chol_inv_rec(s, di, 0, nb, tm);
  }

  void small_eval(Slot<T>& s, size_t di, int target, int mode) {
    SmallEval g{};
    g.X = Xd[di]; g.y = yd[di]; g.n = n; g.d = d; g.P = s.dP;
    g.W2 = s.W2; g.ldiag = s.ldiag; g.Kinv = s.Kinv[target]; g.alpha = s.alpha[target]; g.out = s.dOut; g.mode = mode;
    // The kernel reads the parameters from, and writes its few scalar results to, the slot's pinned host blocks itself: the
    // captured graph of an evaluation is ONE node (no parameter copy, no reset kernel, no result copy -- each was ~2-5 us of
    // a 57 us evaluation).  HBEGP_SMALL_HOSTIO=0: through device memory and copy nodes, as the general path does.
    static const bool hostio = env_int("HBEGP_SMALL_HOSTIO", 1) != 0;
    if (hostio) { g.P = s.hP; g.hout = s.hOut; }
    else { g.hout = s.dOut; HIPCHECK(hipMemcpyAsync(s.dP, s.hP, sizeof(EvalParams), hipMemcpyHostToDevice, s.stream)); }
    launch_small_eval<T>(g, nu2, s.stream);
    if (!hostio) HIPCHECK(hipMemcpyAsync(s.hOut, s.dOut, sizeof(EvalOut), hipMemcpyDeviceToHost, s.stream));
  }

  void enqueue_eval(Slot<T>& s, size_t di, int target, bool want_grad, PhaseTimer* tm) {
    const int nb = np / NB;
    s.gemm_ord = 0;
    const int* info = &s.dOut->info;
    if (small_) {
      if (dry_) return;
      if (tm) tm->begin(PhaseTimer::LEAF);
      small_eval(s, di, target, 1 | 2 | (want_grad ? 4 : 0));
      if (tm) tm->end();
      CHECK_LAUNCHES();
      return;
    }
    const bool hostio = eval_published();
    if (!dry_) {
      if (hostio) {
        EvalPrologue pro;
        pro.dP = s.dP; pro.out = s.dOut;
        if (dag_ && !adhoc_) {
          pro.ctrl = s.dag_ctrl; pro.ctrl_words = (int)(dag_ctrl_bytes / sizeof(int));
          s.ctrl_cleared = true;
        }
        if (tm) tm->begin(PhaseTimer::KMAT);
        launch_kmat<T>(Xd[di], n, d, np, nu2, s.hP, s.W1, info, s.stream, &pro);
        if (tm) tm->end();
      } else {
        HIPCHECK(hipMemcpyAsync(s.dP, s.hP, sizeof(EvalParams), hipMemcpyHostToDevice, s.stream));
        launch_reset_out(s.dOut, s.stream);
        if (tm) tm->begin(PhaseTimer::KMAT);
        launch_kmat<T>(Xd[di], n, d, np, nu2, s.dP, s.W1, info, s.stream);
        if (tm) tm->end();
      }
    }
    s.dag_kinv = s.Kinv[target];
    chol_inv(s, di, nb, tm);
    if (!dry_) {
      if (tm) tm->begin(PhaseTimer::ALPHA);
      launch_alpha_lml<T>(s.W2, np, n, yd[di], s.ldiag, s.wbuf, s.part_t, s.alpha[target], s.dOut, info, s.stream, hostio ? s.tickets : nullptr);
      if (tm) tm->end();
    }
    if (!(dag_ && dag_lauum_ && !adhoc_)) {
      // K^-1 = X^T X (lower)  [LAUUM]   (task-queue path: tiles of the same queue, dag_plan.hpp build_lauum)
      GemmLaunch g{};
      g.nops = 1;
      GemmOp& op = g.op[0];
      op.lda = op.ldb = op.ldc = np;
      op.A = s.W2; op.B = s.W2; op.C = s.Kinv[target];
      op.a_kmajor = 1; op.b_kmajor = 1;
      op.ci0 = 0; op.cj0 = 0; op.mi = nb; op.nj = nb; op.c_lower = 1;
      op.k0 = 0; op.k1 = nb; op.klim = 4; op.maskA = 1; op.maskB = 1;
      gemm(s, di, g, tm, PhaseTimer::LAUUM);
    }
    if (dry_) return;
    bool fuse_grad = false;
    if (want_grad) {
      if (tm) tm->begin(PhaseTimer::GRAD);
      // hostio: the launch's last workgroup also finalises the gradient and publishes the evaluation -- where the launch is a
      // few hundred workgroups (latency-bound sizes).  Every workgroup drains its write-through partials before it takes its
      // ticket (~2 us at the end of its life): with the 2,080 workgroups of n = 4096 queueing for the ~60 CUs the other two
      // task-queue launches leave free that made the launch 138 -> 182 us; there the two tiny launches behind it are free.
      fuse_grad = hostio && (np / 64) * (np / 64 + 1) / 2 <= 512;
      launch_gradtrace<T>(Xd[di], n, d, np, nu2, s.dP, s.Kinv[target], s.alpha[target], s.part_g, s.dOut, info, s.stream,
                          fuse_grad ? s.tickets + 1 : nullptr, fuse_grad ? s.hOut : nullptr);
      if (tm) tm->end();
    }
    if (hostio) {
      if (!want_grad || !fuse_grad) launch_publish_out(s.dOut, s.hOut, s.dP, s.stream);
      s.published = true;
    }
    CHECK_LAUNCHES();
    if (!hostio) {
      HIPCHECK(hipMemcpyAsync(s.hOut, s.dOut, sizeof(EvalOut), hipMemcpyDeviceToHost, s.stream));
      s.published = false;
    }
  }

  // The host side of the end of an evaluation.  Published evaluations (HBEGP_HOSTIO): spin on the serial number the last kernel
  // stores into the pinned result block behind a system-scope fence; hipStreamSynchronize sleeps on an interrupt and wakes up
  // ~50 us late, which three optimiser runs pay 150 times each.  A kernel that faults never publishes: after two seconds the
  // thread falls back to hipStreamSynchronize, which reports the fault (or simply waits for a very long evaluation).
  // Long evaluations are slept through first: the thread remembers how long this slot's last evaluation took (they are all
  // alike) and, from 1.5 ms on, sleeps until an eighth of it (at least 300 us) before that: a fit at n = 4096 then keeps 1.4 cores busy
  // instead of 4 (hipStreamSynchronize spins as well: 4 cores with HBEGP_HOSTIO=0 too; tools/cpu_cost_probe.py).  A sleep that ran past the end shortens the next one.  HBEGP_SPIN_ONLY=1: no sleeping.
  void wait_eval(Slot<T>& s, std::chrono::steady_clock::time_point t_launch) {
    if (s.published) {
      using namespace std::chrono;
      static const bool spin_only = env_int("HBEGP_SPIN_ONLY", 0) != 0;
      const volatile unsigned long long* q = &s.hOut->seq;
      if (!spin_only && s.expect_ns > 1500000) {  // below ~1.5 ms a timer's wake-up jitter (50-100 us) costs more than it saves: n=512 fits 29.7 -> 21.6 /s with a 0.4 ms threshold
        const long long margin = std::max<long long>(300000, s.expect_ns / 8);  // concurrent runs stretch each other by a few per cent, unevenly
        std::this_thread::sleep_until(t_launch + nanoseconds(s.expect_ns - margin));
        if (*q == s.seq) {  // slept too long: the measurement below would include the oversleep
          std::atomic_thread_fence(std::memory_order_acquire);
          s.expect_ns = s.expect_ns * 9 / 10;
          return;
        }
      }
      const auto t0 = steady_clock::now();
      for (unsigned it = 1;; ++it) {
        if (*q == s.seq) {
          std::atomic_thread_fence(std::memory_order_acquire);
          s.expect_ns = duration_cast<nanoseconds>(steady_clock::now() - t_launch).count();
          return;
        }
        __builtin_ia32_pause();
        if ((it & 4095u) == 0 && steady_clock::now() - t0 > seconds(2)) {
          // never in normal operation: a faulted kernel, an evaluation longer than two seconds, or a mismatch between what was
          // captured and what run_eval expects (eval_published) -- say so once, the fallback below still returns the result
          static std::atomic<bool> said{false};
          if (!said.exchange(true))
            fprintf(stderr, "hbegp: an evaluation did not publish its serial number within 2 s; falling back to hipStreamSynchronize\n");
          break;
        }
      }
    }
    HIPCHECK(hipStreamSynchronize(s.stream));
  }

  // Kernel matrix + Cholesky/inverse-factor recursion only (no alpha, no K^-1): leaves X = L^-1 in the slot's W2.
  // Used when a model is built: X of the captured theta is recomputed (bitwise the same arithmetic as in the evaluation, from the
  // same parameters: a device-driven fit hands over the numbers its evaluation ran with, SmallFitResult::best_params).
  int factor_only(size_t di, int si) {
    Slot<T>& s = slots[di][si];
    HIPCHECK(hipSetDevice(s.dev));
    s.gemm_ord = 0;
    s.dag_variant = variant_now(di);
    HIPCHECK(hipMemcpyAsync(s.dP, s.hP, sizeof(EvalParams), hipMemcpyHostToDevice, s.stream));
    launch_reset_out(s.dOut, s.stream);
    s.dag_kinv = nullptr;  // the captured K^-1 stays as it is
    if (small_) {
      small_eval(s, di, 0, 0);  // factor + inverse factor only: alpha and K^-1 of the captured evaluation stay as they are
      CHECK_LAUNCHES();
      HIPCHECK(hipStreamSynchronize(s.stream));
      return s.hOut->info != 0 ? HBEGP_NOT_PD : HBEGP_OK;
    }
    launch_kmat<T>(Xd[di], n, d, np, nu2, s.dP, s.W1, &s.dOut->info, s.stream);
    chol_inv(s, di, np / NB, nullptr);
    CHECK_LAUNCHES();
    HIPCHECK(hipMemcpyAsync(s.hOut, s.dOut, sizeof(EvalOut), hipMemcpyDeviceToHost, s.stream));
    HIPCHECK(hipStreamSynchronize(s.stream));
    if (s.hOut->info < 0) {
      dag_report_timeout(s);
      throw HipError{hipErrorLaunchTimeOut, "factorisation task queue: a dependency wait exceeded its bound", __LINE__};
    }
    return s.hOut->info != 0 ? HBEGP_NOT_PD : HBEGP_OK;
  }

  // Incremental extend (the reference refactorises from scratch, fit.rs:33-68): the prior model was
  // built at the same theta on a prefix of these rows, so the leading q0 = floor(n_prior / 128) diagonal blocks of L,
  // X = L^-1 and the matching block of K^-1 are reused:
  //   [X11 0; X21 X22]:  T = A21 X11^T,  A22 -= T T^T,  U = T X11,  chol_inv(A22),  X21 = -X22 U          O(n^2 k)
  //   K^-1 = [K11^-1 + X21^T X21, .; X22^T X21, X22^T X22]                                                   O(n^2 k)
  // Results go to ping-pong buffer 0 of slot (di, si).  Returns HBEGP_EINVAL when nothing can be reused.
  int extend_from(size_t di, int si, const T* pX, const T* pXinv, const T* pKinv, const T* pldiag, int pn, int pnp) {
    Slot<T>& s = slots[di][si];
    HIPCHECK(hipSetDevice(s.dev));
    const int nb = np / NB, q0 = std::min(pn / NB, nb - 1);
    if (q0 < 1 || pn > n) return HBEGP_EINVAL;
    const int* info = &s.dOut->info;
    HIPCHECK(hipMemcpyAsync(s.dP, s.hP, sizeof(EvalParams), hipMemcpyHostToDevice, s.stream));
    launch_reset_out(s.dOut, s.stream);
    // same leading rows?  (only the kept blocks matter, compare the whole prior prefix anyway)
    launch_prefix_differs<T>(Xd[di], pX, (size_t)pn * d, &s.dOut->n_warn, s.stream);
    CHECK_LAUNCHES();
    HIPCHECK(hipMemcpyAsync(s.hOut, s.dOut, sizeof(EvalOut), hipMemcpyDeviceToHost, s.stream));
    HIPCHECK(hipStreamSynchronize(s.stream));
    if (s.hOut->n_warn != 0) {
      HIPCHECK(hipMemsetAsync(&s.dOut->n_warn, 0, sizeof(int), s.stream));
      return HBEGP_EINVAL;
    }
    const size_t w = (size_t)q0 * NB;
    HIPCHECK(hipMemcpy2DAsync(s.W2, sizeof(T) * np, pXinv, sizeof(T) * pnp, sizeof(T) * w, w, hipMemcpyDeviceToDevice, s.stream));
    HIPCHECK(hipMemcpy2DAsync(s.Kinv[0], sizeof(T) * np, pKinv, sizeof(T) * pnp, sizeof(T) * w, w, hipMemcpyDeviceToDevice, s.stream));
    HIPCHECK(hipMemcpyAsync(s.ldiag, pldiag, sizeof(T) * w, hipMemcpyDeviceToDevice, s.stream));
    launch_kmat<T>(Xd[di], n, d, np, nu2, s.dP, s.W1, info, s.stream);
    const bool was_adhoc = adhoc_;
    adhoc_ = true;
    try {
      chol_inv_split(s, di, 0, q0, nb, nullptr, true);
      launch_alpha_lml<T>(s.W2, np, n, yd[di], s.ldiag, s.wbuf, s.part_t, s.alpha[0], s.dOut, info, s.stream);
      if (pnp / NB > q0) {
        // the prior's K^-1 also holds X^T X contributions of its own trailing (partial) block, which is being replaced:
        // take them out of the kept block first
        GemmLaunch gf{};
        gf.nops = 1;
        GemmOp& fix = gf.op[0];
        fix.A = pXinv; fix.B = pXinv; fix.C = s.Kinv[0];
        fix.lda = fix.ldb = pnp; fix.ldc = np;
        fix.a_kmajor = 1; fix.b_kmajor = 1;
        fix.mi = fix.nj = q0; fix.c_lower = 1;
        fix.k0 = q0; fix.k1 = pnp / NB; fix.alpha_neg = 1; fix.beta_one = 1;
        gemm(s, di, gf, nullptr, PhaseTimer::LAUUM);
      }
      GemmLaunch g{};
      GemmOp base{};
      base.lda = base.ldb = base.ldc = np;
      base.A = s.W2; base.B = s.W2; base.C = s.Kinv[0];
      base.a_kmajor = 1; base.b_kmajor = 1;
      base.k0 = q0; base.k1 = nb; base.maskA = 1; base.maskB = 1;
      GemmOp& keep = g.op[g.nops++];   // kept block: += X21^T X21
      keep = base; keep.mi = keep.nj = q0; keep.c_lower = 1; keep.beta_one = 1;
      GemmOp& rect = g.op[g.nops++];   // new rows x kept columns: X22^T X21 (k >= i)
      rect = base; rect.ci0 = q0; rect.mi = nb - q0; rect.nj = q0; rect.klim = 4;
      GemmOp& tri = g.op[g.nops++];    // new rows x new columns (lower)
      tri = base; tri.ci0 = tri.cj0 = q0; tri.mi = tri.nj = nb - q0; tri.c_lower = 1; tri.klim = 4;
      gemm(s, di, g, nullptr, PhaseTimer::LAUUM);
    } catch (...) {
      adhoc_ = was_adhoc;
      throw;
    }
    adhoc_ = was_adhoc;
    CHECK_LAUNCHES();
    HIPCHECK(hipMemcpyAsync(s.hOut, s.dOut, sizeof(EvalOut), hipMemcpyDeviceToHost, s.stream));
    HIPCHECK(hipStreamSynchronize(s.stream));
    s.last_target = 0;
    if (s.hOut->info == 0 && !(s.hOut->done & 1)) throw HipError{hipErrorLaunchFailure, "extend: the evaluation kernels did not run", __LINE__};
    return s.hOut->info != 0 ? HBEGP_NOT_PD : HBEGP_OK;
  }

  // The task-queue variant for an evaluation that starts now on device di: sized for the slots that are inside an optimiser
  // run at this moment (a fit keeps the count; outside a fit every slot is assumed busy).
  int variant_now(size_t di) const {
    if (!dag_ || dag_var.size() < 2) return 1;
    const int nvar = dag_nvar_;
    const int busy = busy_slots_ ? busy_slots_[di].load(std::memory_order_relaxed) : 0;
    if (dag_ncrowd_ > 0) {
      // runs in flight on this GPU over ALL fits of the process: beyond what this fit alone accounts for, the device is crowded
      const int dev = ctx->devs[di];
      const int total = dev < MAX_DEVS ? g_dev_busy[dev].load(std::memory_order_relaxed) : 0;
      if (total > std::max(busy, nvar)) {
        int l = 0;
        while (l + 1 < dag_ncrowd_ && DAG_CROWD_BUSY[l] < total) ++l;
        return nvar + 1 + l;
      }
    }
    return busy <= 0 ? nvar : std::min(busy, nvar);
  }
  // Run one evaluation on (device index di, slot si) into ping-pong buffer `target`; blocks until the result is on the host.
  int run_eval(size_t di, int si, int target, bool want_grad, bool use_graph, double* lml, double* grad) {
    Slot<T>& s = slots[di][si];
    HIPCHECK(hipSetDevice(s.dev));
    static const bool graphs_on = env_int("HBEGP_NO_GRAPH", 0) == 0;
    s.dag_variant = variant_now(di);
    s.hP->seq = ++s.seq;
    const auto t_launch = std::chrono::steady_clock::now();
    if (use_graph && graphs_on) {
      hipGraphExec_t& ge = s.graph[s.dag_variant][target][want_grad ? 1 : 0];
      if (!ge) {
        hipGraph_t gr = nullptr;
        HIPCHECK(hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal));
        try {
          enqueue_eval(s, di, target, want_grad, nullptr);
        } catch (...) {
          (void)hipStreamEndCapture(s.stream, &gr);
          throw;
        }
        HIPCHECK(hipStreamEndCapture(s.stream, &gr));
        HIPCHECK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        HIPCHECK(hipGraphDestroy(gr));
      }
      HIPCHECK(hipGraphLaunch(ge, s.stream));
    } else {
      enqueue_eval(s, di, target, want_grad, nullptr);
    }
    s.published = eval_published();  // what enqueue_eval records when it is not replayed from a graph (ONE rule for both: eval_published)
    wait_eval(s, t_launch);
    s.last_target = target;
    const int p = d + 2;
    if (s.hOut->info < 0) {
      dag_report_timeout(s);
      throw HipError{hipErrorLaunchTimeOut, "factorisation task queue: a dependency wait exceeded its bound", __LINE__};
    }
    if (s.hOut->info != 0) {
      *lml = -std::numeric_limits<double>::infinity();
      if (grad) for (int j = 0; j < p; ++j) grad[j] = 0.0;  // fit.rs:105-112
      return HBEGP_NOT_PD;
    }
    // the evaluation starts by poisoning its outputs and clearing `done`: a kernel that was skipped cannot pass for a result
    if (!(s.hOut->done & 1) || (want_grad && !(s.hOut->done & 2)))
      throw HipError{hipErrorLaunchFailure, "evaluation: the lml/gradient kernels did not run", __LINE__};
    *lml = s.hOut->lml;
    if (grad) for (int j = 0; j < p; ++j) grad[j] = s.hOut->grad[j];
    // the outputs were poisoned with NaN before the launches: a block of the gradient reduction that never ran, or a genuinely
    // non-finite trace, must not reach the optimiser as a number
    bool finite = std::isfinite(*lml);
    if (grad) for (int j = 0; j < p; ++j) finite = finite && std::isfinite(grad[j]);
    if (!finite) {
      // Which of the two?  launch_reset_out poisons the outputs with ONE bit pattern (a quiet NaN with payload 0x5eed) that no
      // arithmetic produces: an output that still carries it was never written -- an engine bug, reported as such, not a
      // property of the data.  Any other non-finite value is a genuine one (NaN features, overflow).
      auto poisoned = [](double v) { unsigned long long b; memcpy(&b, &v, 8); return b == 0x7ff8000000005eedull; };
      bool never_written = poisoned(s.hOut->lml);
      if (grad) for (int j = 0; j < p; ++j) never_written = never_written || poisoned(s.hOut->grad[j]);
      if (never_written) throw HipError{hipErrorLaunchFailure, "evaluation: an output of the lml/gradient kernels was never written", __LINE__};
    }
    if (!finite) {  // handled like a failed factorisation (lml.rs:47-50 -> fit.rs:105-112): objective +inf, zero gradient, never captured
      *lml = -std::numeric_limits<double>::infinity();
      if (grad) for (int j = 0; j < p; ++j) grad[j] = 0.0;
      return HBEGP_NOT_PD;
    }
    return HBEGP_OK;
  }

  int eval(int dev, int slot, const double* theta, const double* lo, const double* hi, double* lml, double* grad) override {
    if (dev < 0 || dev >= (int)slots.size() || slot < 0 || slot >= n_slots) return fail(HBEGP_EINVAL, "bad device/slot index");
    Slot<T>& s = slots[dev][slot];
    theta_to_params(theta, lo, hi, d, s.hP);
    // never overwrite the captured best: write into the other buffer
    const int target = (s.best_idx < 0) ? 0 : 1 - s.best_idx;
    return run_eval((size_t)dev, slot, target, grad != nullptr, true, lml, grad);
  }

  int time_eval(int dev, int slot, const double* theta, int reps, double* phase_ms) override {
    if (dev < 0 || dev >= (int)slots.size() || slot < 0 || slot >= n_slots) return fail(HBEGP_EINVAL, "bad device/slot index");
    Slot<T>& s = slots[dev][slot];
    HIPCHECK(hipSetDevice(s.dev));
    theta_to_params(theta, nullptr, nullptr, d, s.hP);
    double lml;
    std::vector<double> grad(d + 2);
    int st = run_eval((size_t)dev, slot, 0, true, true, &lml, grad.data());  // warm-up + graph instantiation
    if (st != HBEGP_OK) return st;
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, s.stream));
    hipGraphExec_t timed_graph = s.graph[s.dag_variant][0][1];  // what the warm-up evaluation instantiated
    for (int r = 0; r < reps; ++r) {
      if (timed_graph) HIPCHECK(hipGraphLaunch(timed_graph, s.stream));
      else enqueue_eval(s, (size_t)dev, 0, true, nullptr);  // HBEGP_NO_GRAPH=1
    }
    HIPCHECK(hipEventRecord(e1, s.stream));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    if (phase_ms) {
      for (int i = 0; i < 24; ++i) phase_ms[i] = 0;
      phase_ms[6] = ms / reps;
      // eager pass with one event pair per launch
      const int treps = std::max(1, std::min(reps, 3));
      for (int r = 0; r < treps; ++r) {
        PhaseTimer tm;
        tm.s = s.stream;
        hipEvent_t t0, t1;
        HIPCHECK(hipEventCreate(&t0));
        HIPCHECK(hipEventCreate(&t1));
        HIPCHECK(hipEventRecord(t0, s.stream));
        enqueue_eval(s, (size_t)dev, 0, true, &tm);
        HIPCHECK(hipEventRecord(t1, s.stream));
        HIPCHECK(hipStreamSynchronize(s.stream));
        float tot = 0;
        HIPCHECK(hipEventElapsedTime(&tot, t0, t1));
        phase_ms[14] += tot / treps;
        for (auto& rec : tm.recs) {
          float dt = 0;
          HIPCHECK(hipEventElapsedTime(&dt, rec.a, rec.b));
          const double v = dt / treps;
          static const bool dump = env_int("HBEGP_TRACE_LAUNCHES", 0) != 0;
          if (dump && r == treps - 1)
            fprintf(stderr, "launch kind=%d tile=%d gflop=%.4f ms=%.4f tflops=%.2f\n", rec.kind, rec.tile, rec.gflop, dt,
                    dt > 0 ? rec.gflop / dt : 0.0);
          if (rec.kind == PhaseTimer::KMAT) phase_ms[0] += v;
          if (rec.kind == PhaseTimer::GEMM) { phase_ms[1] += v; phase_ms[7] += 1.0 / treps; }
          if (rec.kind == PhaseTimer::LEAF) { phase_ms[2] += v; phase_ms[15] += 1.0 / treps; }
          if (rec.kind == PhaseTimer::LAUUM) phase_ms[3] += v;
          if (rec.kind == PhaseTimer::ALPHA) phase_ms[4] += v;
          if (rec.kind == PhaseTimer::GRAD) phase_ms[5] += v;
          if (rec.kind == PhaseTimer::DAG) { phase_ms[19] += v; phase_ms[20] += rec.gflop / treps; }
          if (rec.kind == PhaseTimer::GEMM || rec.kind == PhaseTimer::LAUUM) {
            const int o = rec.tile == 128 ? 8 : (rec.tile == 64 ? 10 : 12);
            phase_ms[o] += v;
            phase_ms[o + 1] += rec.gflop / treps;
            phase_ms[rec.tile == 128 ? 16 : (rec.tile == 64 ? 17 : 18)] += 1.0 / treps;  // launches per evaluation
          }
          (void)hipEventDestroy(rec.a);
          (void)hipEventDestroy(rec.b);
        }
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
      }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (s.dag_trace && getenv("HBEGP_DAG_TRACE")) {
      // one more graph replay on a quiet device, then dump: idx kind row col depth nwait | pulled ready computed published (ticks of 10 ns) | xcc hwid
      if (timed_graph) HIPCHECK(hipGraphLaunch(timed_graph, s.stream));
      HIPCHECK(hipStreamSynchronize(s.stream));
      const std::vector<DagTask>& dag_host_tasks = dag_var[s.dag_variant].host_tasks;
      std::vector<unsigned long long> tr((size_t)5 * dag_ntasks);
      HIPCHECK(hipMemcpyAsync(tr.data(), s.dag_trace, sizeof(unsigned long long) * tr.size(), hipMemcpyDeviceToHost, s.stream));
      HIPCHECK(hipStreamSynchronize(s.stream));
      if (FILE* f = fopen(getenv("HBEGP_DAG_TRACE"), "w")) {
        for (int i = 0; i < dag_ntasks; ++i) {
          const DagTask& t = dag_host_tasks[i];
          fprintf(f, "%d %d %d %d %d %d %llu %llu %llu %llu %llu %llu\n", i, t.kind, t.row0, t.col0, t.kend - t.kbeg, t.nwait, tr[5 * i], tr[5 * i + 1],
                  tr[5 * i + 2], tr[5 * i + 3], tr[5 * i + 4] >> 32, tr[5 * i + 4] & 0xffffffffull);
        }
        fclose(f);
      }
    }
    return HBEGP_OK;
  }

  int time_concurrent(int dev, const double* theta, int reps, double* out) override {
    if (dev < 0 || dev >= (int)slots.size()) return fail(HBEGP_EINVAL, "bad device index");
    const size_t di = (size_t)dev;
    const int ns = n_slots;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    out[1] = ns;
    out[10] = dag_ ? dag_nwg : 0;
    std::string err;
    std::mutex err_mu;
    // pass 0: warm-up (graph instantiation), pass 1: graph replay, pass 2: eager with events
    std::vector<double> acc(12, 0.0);
    std::mutex acc_mu;
    for (int pass = 0; pass < 3; ++pass) {
      std::atomic<int> arrived{0};
      std::vector<double> wall(ns, 0.0);
      auto worker = [&](int si) {
        try {
          Slot<T>& s = slots[di][si];
          HIPCHECK(hipSetDevice(s.dev));
          theta_to_params(theta, nullptr, nullptr, d, s.hP);
          std::vector<double> grad(d + 2);
          double lml;
          arrived.fetch_add(1);
          while (arrived.load() < ns) std::this_thread::yield();  // start together
          const auto t0 = std::chrono::steady_clock::now();
          const int nrep = pass == 0 ? 1 : reps;
          for (int r = 0; r < nrep; ++r) {
            if (pass < 2) {
              const int st = run_eval(di, si, 0, true, true, &lml, grad.data());
              if (st != HBEGP_OK) throw std::runtime_error("evaluation failed (not positive definite) at the timing theta");
            } else {
              PhaseTimer tm;
              tm.s = s.stream;
              enqueue_eval(s, di, 0, true, &tm);
              HIPCHECK(hipStreamSynchronize(s.stream));
              std::vector<double> loc(12, 0.0);
              for (auto& rec : tm.recs) {
                float dt = 0;
                HIPCHECK(hipEventElapsedTime(&dt, rec.a, rec.b));
                if (rec.kind == PhaseTimer::KMAT) loc[3] += dt;
                if (rec.kind == PhaseTimer::DAG || rec.kind == PhaseTimer::GEMM || rec.kind == PhaseTimer::LEAF) { loc[4] += dt; loc[5] += rec.gflop; loc[11] += 1; }
                if (rec.kind == PhaseTimer::LAUUM) { loc[6] += dt; loc[7] += rec.gflop; }
                if (rec.kind == PhaseTimer::ALPHA) loc[8] += dt;
                if (rec.kind == PhaseTimer::GRAD) loc[9] += dt;
                (void)hipEventDestroy(rec.a);
                (void)hipEventDestroy(rec.b);
              }
              std::lock_guard<std::mutex> lk(acc_mu);
              for (int i = 0; i < 12; ++i) acc[i] += loc[i] / ((double)nrep * ns);
            }
          }
          wall[si] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / nrep;
        } catch (const HipError& he) {
          hip_fail(he);
          std::lock_guard<std::mutex> lk(err_mu);
          err = g_last_error;
          arrived.fetch_add(ns);  // never leave the others spinning at the start line
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> lk(err_mu);
          err = e.what();
          arrived.fetch_add(ns);
        }
      };
      std::vector<std::thread> threads;
      try {
        for (int si = 0; si < ns; ++si) threads.emplace_back(worker, si);
      } catch (...) {
        arrived.fetch_add(ns);
        for (auto& t : threads) t.join();
        throw;
      }
      for (auto& t : threads) t.join();
      if (!err.empty()) return fail(HBEGP_EHIP, "%s", err.c_str());
      const double wmax = *std::max_element(wall.begin(), wall.end());
      if (pass == 1) out[0] = wmax;
      if (pass == 2) out[2] = wmax;
    }
    for (int i = 3; i < 12; ++i)
      if (i != 10) out[i] = acc[i];
    return HBEGP_OK;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// model
struct hbegp_model {
  std::atomic<int> refs{1};
  int dev = 0, n = 0, d = 0, np = 0, nu2 = 5;
  bool is_f32 = false;
  double lml = 0;
  std::vector<double> theta;  // clamped, log space
  void *X = nullptr, *alpha = nullptr, *Kinv = nullptr;  // device
  void* Xinv = nullptr;  // L^-1 (lower), for the predictive variance as c + 1e-5 - |L^-1 k*|^2
  void* ldiag = nullptr; // diag(L), np entries (incremental extend needs the log-determinant of the kept part)
  size_t kinv_bytes = 0;
  EvalParams* dP = nullptr;
  EvalOut* dOut = nullptr;
  hipStream_t stream = nullptr;
  // predict scratch (grow-only)
  int cap_m = 0;
  void *Xs = nullptr, *Ks = nullptr, *Q = nullptr, *mean = nullptr, *var = nullptr;
  // scratch of the path for a handful of candidates (allocated on first use): device [Xs | Ks | out], partial sums, and
  // pinned host staging so that a single-point predict costs one H2D and one D2H
  void *sm_Xs = nullptr, *sm_Ks = nullptr, *sm_out = nullptr, *sm_hin = nullptr, *sm_hout = nullptr;
  double *sm_pmean = nullptr, *sm_w = nullptr;
  std::mutex mu;
  // every device array of the model comes from the block pool and goes back to it (no hipMalloc / hipFree on the caller's
  // path: they synchronise the whole device, i.e. every other host thread's fit); pinned blocks and the stream likewise
  struct Pooled { void* p; size_t bytes; };
  std::vector<Pooled> pooled;
  size_t sm_hin_bytes = 0, sm_hout_bytes = 0;
  void* palloc(size_t bytes) {
    bool fresh = false;
    bytes = std::max<size_t>(16, bytes);
    void* q = g_pool.get(dev, bytes, &fresh);
    pooled.push_back({q, bytes});
    return q;
  }
  void pfree(void* q) {
    for (size_t i = 0; i < pooled.size(); ++i)
      if (pooled[i].p == q) {
        g_pool.put(dev, q, pooled[i].bytes);
        pooled.erase(pooled.begin() + (long)i);
        return;
      }
  }
  ~hbegp_model() {
    (void)hipSetDevice(dev);
    if (stream) (void)hipStreamSynchronize(stream);
    g_pool.put(dev, Kinv, kinv_bytes); g_pool.put(dev, Xinv, kinv_bytes);
    for (const Pooled& q : pooled) g_pool.put(dev, q.p, q.bytes);
    g_host_pool.put(sm_hin, sm_hin_bytes); g_host_pool.put(sm_hout, sm_hout_bytes);
    g_stream_pool.put(dev, stream, STREAM_MODEL);  // synchronised above
  }
};

template <typename T>
static hbegp_model* make_model(Problem<T>& prob, size_t di, int si, const double* theta_clamped, double lml,
                               bool w2_current = false, const double* params_linear = nullptr) {
  Slot<T>& s = prob.slots[di][si];
  HIPCHECK(hipSetDevice(s.dev));
  std::unique_ptr<hbegp_model> m(new hbegp_model());
  m->dev = s.dev; m->n = prob.n; m->d = prob.d; m->np = prob.np; m->nu2 = prob.nu2; m->is_f32 = prob.is_f32; m->lml = lml;
  m->theta.assign(theta_clamped, theta_clamped + prob.d + 2);
  const size_t nn = (size_t)prob.np * prob.np;
  // (a model that took over its fit's first slot stream instead -- one stream fewer per fit in flight -- was tried in round 5:
  // nothing gained for fits side by side, and the slot streams then change hands from fit to fit, so that two slots of one fit
  // end up on one hardware queue now and then: bench 1.777 -> 1.739, solo fits at n = 1024 16.4 -> 6.7 per s in some processes)
  m->stream = g_stream_pool.get(m->dev, STREAM_MODEL);
  m->X = m->palloc(sizeof(T) * (size_t)prob.n * prob.d);
  m->alpha = m->palloc(sizeof(T) * prob.np);
  { bool fr; m->Kinv = g_pool.get(m->dev, sizeof(T) * nn, &fr); m->Xinv = g_pool.get(m->dev, sizeof(T) * nn, &fr); m->kinv_bytes = sizeof(T) * nn; }
  m->dP = static_cast<EvalParams*>(m->palloc(sizeof(EvalParams)));
  m->dOut = static_cast<EvalOut*>(m->palloc(sizeof(EvalOut)));
  const int b = s.best_idx < 0 ? s.last_target : s.best_idx;
  // X = L^-1 at the model's theta (the evaluation slots only keep K^-1 and alpha of the captured evaluation)
  // (w2_current: the slot's last evaluation WAS at this theta -- extend -- so W2 and ldiag already hold them)
  auto fill_params = [&](EvalParams* P) {
    if (params_linear) {  // a device-driven fit: the very numbers the captured evaluation ran with
      P->noise = params_linear[0];
      P->amp = params_linear[1];
      for (int k = 0; k < prob.d; ++k) P->ell[k] = params_linear[2 + k];
    } else {
      theta_to_params(theta_clamped, nullptr, nullptr, prob.d, P);
    }
  };
  // (x_captured: a device-driven small fit kept the captured evaluation's own factor: buffer b of the ping-pong pair)
  const T* Xsrc = (s.x_captured && b == 1) ? s.Xalt : s.W2;
  const T* ldsrc = (s.x_captured && b == 1) ? s.ldalt : s.ldiag;
  if (!w2_current && !s.x_captured) {
    fill_params(s.hP);
    if (prob.factor_only(di, si) != HBEGP_OK) throw HipError{hipErrorUnknown, "factorisation at the captured theta failed", __LINE__};
  }
  m->ldiag = m->palloc(sizeof(T) * prob.np);
  HIPCHECK(hipMemcpyAsync(m->ldiag, ldsrc, sizeof(T) * prob.np, hipMemcpyDeviceToDevice, m->stream));
  HIPCHECK(hipMemcpyAsync(m->Xinv, Xsrc, sizeof(T) * nn, hipMemcpyDeviceToDevice, m->stream));
  HIPCHECK(hipMemcpyAsync(m->X, prob.Xd[di], sizeof(T) * (size_t)prob.n * prob.d, hipMemcpyDeviceToDevice, m->stream));
  HIPCHECK(hipMemcpyAsync(m->alpha, s.alpha[b], sizeof(T) * prob.np, hipMemcpyDeviceToDevice, m->stream));
  HIPCHECK(hipMemcpyAsync(m->Kinv, s.Kinv[b], sizeof(T) * nn, hipMemcpyDeviceToDevice, m->stream));
  launch_symmetrize<T>(static_cast<T*>(m->Kinv), prob.np, m->stream);  // invc_into() returns the full matrix (fit.rs:60,168)
  CHECK_LAUNCHES();
  EvalParams P;
  memset(&P, 0, sizeof(P));
  fill_params(&P);
  HIPCHECK(hipMemcpyAsync(m->dP, &P, sizeof(P), hipMemcpyHostToDevice, m->stream));
  HIPCHECK(hipStreamSynchronize(m->stream));
  return m.release();
}

template <typename T>
static int model_predict(hbegp_model* m, const T* Xs, int cnt, T* mean, T* var, int* n_warn) {
  std::lock_guard<std::mutex> lock(m->mu);
  HIPCHECK(hipSetDevice(m->dev));
  static const bool small_on = env_int("HBEGP_PRED_SMALL", 1) != 0;
  static const bool kinv_form_small = env_int("HBEGP_PREDVAR_KINV", 0) != 0;
  static const int small_max = std::min(PRED_SMALL_MAX, std::max(0, env_int("HBEGP_PRED_SMALL_MAX", 8)));
  if (small_on && cnt <= small_max && !kinv_form_small) {
    // a handful of candidates (the caller's scalar predict_* loops): read L^-1 once instead of a padded 128-row tile GEMM
    const size_t out_bytes = sizeof(T) * 2 * PRED_SMALL_MAX + 16;
    if (!m->sm_Xs) {
      m->sm_Xs = m->palloc(sizeof(T) * PRED_SMALL_MAX * m->d);
      m->sm_Ks = m->palloc(sizeof(T) * (size_t)PRED_SMALL_MAX * m->np);
      m->sm_out = m->palloc(out_bytes);
      m->sm_pmean = static_cast<double*>(m->palloc(sizeof(double) * (size_t)((m->np + 255) / 256) * PRED_SMALL_MAX));
      m->sm_w = static_cast<double*>(m->palloc(sizeof(double) * (size_t)m->n * PRED_SMALL_MAX));
      m->sm_hin_bytes = sizeof(T) * PRED_SMALL_MAX * m->d;
      m->sm_hout_bytes = out_bytes;
      m->sm_hin = g_host_pool.get(m->sm_hin_bytes);
      m->sm_hout = g_host_pool.get(m->sm_hout_bytes);
    }
    hipStream_t s = m->stream;
    memcpy(m->sm_hin, Xs, sizeof(T) * (size_t)cnt * m->d);
    HIPCHECK(hipMemcpyAsync(m->sm_Xs, m->sm_hin, sizeof(T) * (size_t)cnt * m->d, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemsetAsync(m->sm_out, 0, out_bytes, s));
    T* dmean = static_cast<T*>(m->sm_out);
    T* dvar = dmean + PRED_SMALL_MAX;
    int* dwarn = reinterpret_cast<int*>(dvar + PRED_SMALL_MAX);
    launch_predict_small<T>(static_cast<T*>(m->sm_Xs), cnt, static_cast<T*>(m->X), m->n, m->d, m->np, m->nu2, m->dP, static_cast<T*>(m->alpha),
                            static_cast<T*>(m->Xinv), static_cast<T*>(m->sm_Ks), m->sm_pmean, m->sm_w, var ? 1 : 0, dmean, dvar, dwarn, s);
    CHECK_LAUNCHES();
    HIPCHECK(hipMemcpyAsync(m->sm_hout, m->sm_out, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    const T* hmean = static_cast<const T*>(m->sm_hout);
    memcpy(mean, hmean, sizeof(T) * cnt);
    if (var) memcpy(var, hmean + PRED_SMALL_MAX, sizeof(T) * cnt);
    if (n_warn) *n_warn = var ? *reinterpret_cast<const int*>(hmean + 2 * PRED_SMALL_MAX) : 0;
    return HBEGP_OK;
  }
  const int mp = round_up(std::max(cnt, 1), NB);
  if (mp > m->cap_m) {
    HIPCHECK(hipStreamSynchronize(m->stream));  // nothing of an earlier predict is still using the smaller arrays
    m->pfree(m->Xs); m->pfree(m->Ks); m->pfree(m->Q); m->pfree(m->mean); m->pfree(m->var);
    m->Xs = m->Ks = m->Q = m->mean = m->var = nullptr;
    m->cap_m = 0;
    m->Xs = m->palloc(sizeof(T) * (size_t)mp * m->d);
    m->Ks = m->palloc(sizeof(T) * (size_t)mp * m->np);
    m->Q = m->palloc(sizeof(T) * (size_t)mp * m->np);
    m->mean = m->palloc(sizeof(T) * mp);
    m->var = m->palloc(sizeof(T) * mp);
    m->cap_m = mp;
  }
  hipStream_t s = m->stream;
  HIPCHECK(hipMemcpyAsync(m->Xs, Xs, sizeof(T) * (size_t)cnt * m->d, hipMemcpyHostToDevice, s));
  HIPCHECK(hipMemsetAsync(m->dOut, 0, sizeof(EvalOut), s));
  launch_kstar<T>(static_cast<T*>(m->Xs), cnt, mp, static_cast<T*>(m->X), m->n, m->d, m->np, m->nu2, m->dP,
                  static_cast<T*>(m->Ks), s);
  launch_pred_mean<T>(static_cast<T*>(m->Ks), cnt, m->np, static_cast<T*>(m->alpha), static_cast<T*>(m->mean), s);
  if (var) {
    static const bool kinv_form = env_int("HBEGP_PREDVAR_KINV", 0) != 0;
    GemmLaunch g{};
    g.nops = 1;
    g.info = &m->dOut->info;
    GemmOp& op = g.op[0];
    op.lda = m->np; op.ldb = m->np; op.ldc = m->np;
    op.ci0 = 0; op.cj0 = 0; op.mi = mp / NB; op.nj = m->np / NB;
    op.k0 = 0; op.k1 = m->np / NB;
    if (kinv_form) {
      // the reference's literal formula (predict.rs:30-37): Q = Kstar * Kinv, var = c + 1e-5 - rowsum(Q o Kstar)
      op.A = m->Ks; op.B = m->Kinv; op.C = m->Q;
      launch_gemm<T>(g, pick_tile(op.mi * op.nj), s);
      launch_pred_var<T>(static_cast<T*>(m->Ks), static_cast<T*>(m->Q), cnt, m->np, m->dP, static_cast<T*>(m->var), m->dOut, s);
    } else {
      // same quantity as k*^T K^-1 k* = |L^-1 k*|^2: Q = Kstar * X^T (X = L^-1 lower: k <= j, half the flops), then
      // var = c + 1e-5 - rowsum(Q o Q).  A sum of squares has no cancellation inside the quadratic form, so the
      // result is at least as close to the exact value as the K^-1 form.
      op.A = m->Ks; op.B = m->Xinv; op.C = m->Q;
      op.klim = 1; op.maskB = 1;
      launch_gemm<T>(g, pick_tile(op.mi * op.nj), s);
      launch_pred_var<T>(static_cast<T*>(m->Q), static_cast<T*>(m->Q), cnt, m->np, m->dP, static_cast<T*>(m->var), m->dOut, s);
    }
  }
  CHECK_LAUNCHES();
  HIPCHECK(hipMemcpyAsync(mean, m->mean, sizeof(T) * cnt, hipMemcpyDeviceToHost, s));
  if (var) HIPCHECK(hipMemcpyAsync(var, m->var, sizeof(T) * cnt, hipMemcpyDeviceToHost, s));
  EvalOut out;
  HIPCHECK(hipMemcpyAsync(&out, m->dOut, sizeof(EvalOut), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  if (n_warn) *n_warn = var ? out.n_warn : 0;
  return HBEGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fit (fit.rs:71-176 + gradmin.rs:7-60)
// The caller's options struct may be older (shorter) or newer (longer) than this library's: copy what both know, the rest
// stays zero / NULL.  Nothing is read or written beyond min(struct_size, sizeof).
static int read_fit_options(const hbegp_fit_options* in, hbegp_fit_options* out) {
  *out = hbegp_fit_options{};
  out->struct_size = sizeof(hbegp_fit_options);
  if (!in) return HBEGP_OK;
  if (in->struct_size < offsetof(hbegp_fit_options, maxeval) + sizeof(int) || in->struct_size > ((size_t)1 << 16))
    return fail(HBEGP_EINVAL, "hbegp_fit_options.struct_size = %zu: set it to sizeof(hbegp_fit_options) (HBEGP_FIT_OPTIONS_INIT)",
                in->struct_size);
  memcpy(out, in, std::min(in->struct_size, sizeof(*out)));
  return HBEGP_OK;
}

template <typename T>
static int do_fit(hbegp_ctx* ctx, const T* X, const T* y, int n, int d, double nu, const double* theta0, const double* lo,
                  const double* hi, const double* starts, int n_restarts, const hbegp_fit_options* opt_in,
                  double* theta_best, double* lml_best, hbegp_model** model_out) {
  hbegp_fit_options opt{};
  if (int e = read_fit_options(opt_in, &opt)) return e;
  if (opt.maxeval <= 0) opt.maxeval = 150;
  const int p = d + 2;
  const int nruns = 1 + std::max(0, n_restarts);
  const int ndev = (int)ctx->devs.size();
  const int max_conc = std::max(1, env_int("HBEGP_MAX_CONCURRENT", 3));  // measured (C3, 8 runs): 1: 3.29, 2: 2.36, 3: 2.09, 4: 2.33, 8: 2.10 ms per evaluation
  // workers: device di gets min(runs on that device, max_conc) slots; run r -> device r % ndev
  std::vector<int> runs_on(ndev, 0);
  for (int r = 0; r < nruns; ++r) runs_on[r % ndev]++;
  int n_slots = 1;
  for (int di = 0; di < ndev; ++di) n_slots = std::max(n_slots, std::min(runs_on[di], max_conc));
  // Up to 128 rows an optimiser run is ONE persistent launch on ONE compute unit (small_fit_kernel: evaluation + L-BFGS step +
  // capture on the device): every run gets a slot of its own and all of them run side by side
  constexpr int SMALL_FIT_MAX_RUNS = 64;  // per device
  bool small_fit = round_up(n, NB) == NB && d <= SMALL_EVAL_MAXD && env_int("HBEGP_SMALL", 1) != 0 && env_int("HBEGP_SMALL_FIT", 1) != 0 &&
                   !(sizeof(T) == 4 && env_int("HBEGP_F32_REFINE", 0) != 0);
  for (int di = 0; di < ndev; ++di) small_fit = small_fit && runs_on[di] <= SMALL_FIT_MAX_RUNS;
  if (small_fit)
    for (int di = 0; di < ndev; ++di) n_slots = std::max(n_slots, runs_on[di]);
  static const int timing = env_int("HBEGP_TIMING", 0);
  const auto tf0 = std::chrono::steady_clock::now();
  // like_fit: the evaluation path (launches / task queue) is a function of n alone, also for a fit with one slot per device
  // (n_restarts = 0, or no more runs than devices): `extend` at the fitted theta then repeats the fit's own evaluation bit for bit
  SmallArrival arrival;  // other threads' small fits wait (briefly) for this one's runs before they launch: SmallBatcher
  if (small_fit) arrival.announce();
  // The host-side phases of a small fit (set-up; model + release: ~35 runtime calls together) run one thread at a time: sixteen
  // native threads that enter them at the same instant queue on the runtime's locks for 2-3 ms each, one after the other they take
  // 0.1-0.2 ms (measured, n = 128, fits/s at 4 / 8 / 16 native threads: 312 / 507 / 865 -> 351 / 658 / 1,157; Python threads, which
  // the interpreter lock staggers already: 1,198 -> 1,104 at 16).  Only while at most HBEGP_SMALL_HOST_SERIAL (16; 0 = never) threads
  // are inside small fits: beyond that the turns themselves are the queue (24 threads: 1,031 without / 862 with; 32 threads on 16
  // cores: 744 / 238), and a bounded wait for the turn (1-10 ms) keeps none of the gain (16 threads: 740-770).
  static const int host_serial_max = env_int("HBEGP_SMALL_HOST_SERIAL", 16);
  struct SmallActive {
    bool on = false;
    int enter() { on = true; return g_small_active.fetch_add(1) + 1; }
    ~SmallActive() { if (on) g_small_active.fetch_sub(1); }
  } small_active;
  const bool host_serial = small_fit && small_active.enter() <= host_serial_max;
  std::unique_lock<std::mutex> host_lk(g_small_host_mu, std::defer_lock);
  if (host_serial) host_lk.lock();
  const auto th0 = std::chrono::steady_clock::now();  // (HBEGP_TIMING: how long the turns are)
  double held_ms = 0;
  Problem<T> prob(ctx, X, y, n, d, nu, n_slots, false, true);
  const auto tf1 = std::chrono::steady_clock::now();
  if (!(small_fit && prob.small_)) arrival.arrived();
  if (!(small_fit && prob.small_) && host_lk.owns_lock()) host_lk.unlock();

  std::vector<double> lnlo(p), lnhi(p);
  for (int i = 0; i < p; ++i) {
    lnlo[i] = std::log(lo[i]);
    lnhi[i] = std::log(hi[i]);
  }
  std::mutex trace_mu;
  int trace_n = 0;
  std::atomic<int> n_evals{0}, n_not_pd{0};
  std::string err;
  std::mutex err_mu;

  if (small_fit && prob.small_) {
    struct RunWs {
      int di = 0, si = 0, run = 0;
      LbfgsState* st = nullptr;
      SmallFitResult* res = nullptr;
      double *x0 = nullptr, *tr_theta = nullptr, *tr_lml = nullptr, *tr_grad = nullptr;
    };
    const int cap = opt.trace_cap > 0 ? opt.maxeval : 0;  // per run; the merged trace is cut at opt.trace_cap below
    std::vector<RunWs> ws(nruns);
    // per run, in pinned memory: the result and the word the kernel sets when the run is over (SmallFit::hres / hdone)
    struct alignas(64) HostRun {
      SmallFitResult res;
      unsigned long long done;
    };
    const size_t hruns_bytes = (sizeof(HostRun) * (size_t)nruns + 4095) / 4096 * 4096;
    HostRun* hruns = static_cast<HostRun*>(g_host_pool.get(hruns_bytes));
    std::vector<std::shared_ptr<SmallBatch>> batches(ndev);
    bool runs_over = false;
    auto free_all = [&]() {  // (the trace arrays belong to the problem's pooled allocations, released with it)
      for (auto& b : batches) small_batch_leave(b, runs_over);
      batches.clear();
      g_host_pool.put(hruns, hruns_bytes);
    };
    try {
      std::vector<int> next_slot(ndev, 0);
      // the runs of a device are the workgroups of ONE launch on the device's first slot stream (one stream per fit and device:
      // fits side by side on one GPU then do not queue behind each other's persistent kernels, see small_fit_kernel)
      std::vector<std::vector<SmallFit>> fits_on(ndev);
      std::vector<double*> boxes_dev(ndev, nullptr);
      std::vector<std::vector<double>> boxes_host(ndev);  // the source of an asynchronous copy: alive until the stream has been waited for below
      for (int r = 0; r < nruns; ++r) {
        RunWs& w = ws[r];
        w.di = r % ndev; w.si = next_slot[w.di]++; w.run = r;
        HIPCHECK(hipSetDevice(ctx->devs[w.di]));
        Slot<T>& s = prob.slots[w.di][w.si];
        hipStream_t st = prob.slots[w.di][0].stream;
        // the run's workspace: the slot's W1 (the kernel matrix never leaves the LDS on this path, so the buffer is free) --
        // a hipMalloc / hipFree pair per array cost more than the run's arithmetic
        static_assert(sizeof(LbfgsState) + sizeof(SmallFitResult) + 3 * MAXP * sizeof(double) + 64 <= (size_t)NB * NB * sizeof(float),
                      "the run's workspace fits the slot's W1");
        char* base = reinterpret_cast<char*>(s.W1);
        w.st = reinterpret_cast<LbfgsState*>(base);
        w.res = reinterpret_cast<SmallFitResult*>(base + (sizeof(LbfgsState) + 15) / 16 * 16);
        // start point and box of every run of a device: one array, one copy (a copy per run was a runtime call per run inside the turn)
        if (!boxes_dev[w.di]) {
          boxes_dev[w.di] = prob.template palloc<double>(s.dev, (size_t)runs_on[w.di] * 3 * p);
          boxes_host[w.di].assign((size_t)runs_on[w.di] * 3 * p, 0.0);
        }
        w.x0 = boxes_dev[w.di] + (size_t)w.si * 3 * p;
        if (cap > 0) {
          w.tr_theta = prob.template palloc<double>(s.dev, (size_t)cap * p);
          w.tr_lml = prob.template palloc<double>(s.dev, cap);
          w.tr_grad = prob.template palloc<double>(s.dev, (size_t)cap * p);
        }
        const double* start = r == 0 ? theta0 : starts + (size_t)(r - 1) * p;
        double* box = boxes_host[w.di].data() + (size_t)w.si * 3 * p;  // start point, lower and upper bounds
        memcpy(box, start, sizeof(double) * p);
        memcpy(box + p, lo, sizeof(double) * p);
        memcpy(box + 2 * p, hi, sizeof(double) * p);
        SmallFit f{};
        f.ev.X = prob.Xd[w.di]; f.ev.y = prob.yd[w.di]; f.ev.n = n; f.ev.d = d; f.ev.P = s.dP;
        f.ev.W2 = s.W2; f.ev.ldiag = s.ldiag; f.ev.out = s.dOut; f.ev.hout = s.dOut;
        f.Kinv[0] = s.Kinv[0]; f.Kinv[1] = s.Kinv[1]; f.alpha[0] = s.alpha[0]; f.alpha[1] = s.alpha[1];
        // L^-1 and diag(L) ping-pong too: the captured evaluation's factor survives the run, the model takes it as it is (no
        // factorisation at the captured theta behind the fit: one small launch and a wait less inside the model's turn)
        static const bool x_pingpong = env_int("HBEGP_SMALL_X_PINGPONG", 1) != 0;
        if (x_pingpong) {
          if (!s.Xalt) {
            s.Xalt = prob.template palloc<T>(s.dev, (size_t)NB * NB);
            s.ldalt = prob.template palloc<T>(s.dev, NB);
            HIPCHECK(hipMemsetAsync(s.Xalt, 0, sizeof(T) * NB * NB, st));  // the strict upper triangle must be zero, like W2's
          }
          f.Xinv[0] = s.W2; f.Xinv[1] = s.Xalt; f.ldiag[0] = s.ldiag; f.ldiag[1] = s.ldalt;
          s.x_captured = true;
        }
        f.st = w.st; f.x0 = w.x0; f.lo = w.x0 + p; f.hi = w.x0 + 2 * p;
        LbfgsOptions lo_opt;
        f.maxeval = opt.maxeval; f.memory = opt.lbfgs_memory > 0 ? opt.lbfgs_memory : lo_opt.memory; f.fixed_work = opt.fixed_work != 0;
        f.pgtol = lo_opt.pgtol; f.ftol = lo_opt.ftol;
        f.res = w.res; f.trace_theta = w.tr_theta; f.trace_lml = w.tr_lml; f.trace_grad = w.tr_grad; f.trace_cap = cap;
        hruns[r].done = 0;
        f.hres = &hruns[r].res; f.hdone = &hruns[r].done;
        fits_on[w.di].push_back(f);
      }
      for (int di = 0; di < ndev; ++di) {
        if (fits_on[di].empty()) continue;
        HIPCHECK(hipSetDevice(ctx->devs[di]));
        HIPCHECK(hipMemcpyAsync(boxes_dev[di], boxes_host[di].data(), sizeof(double) * boxes_host[di].size(), hipMemcpyHostToDevice, prob.slots[di][0].stream));
        // the grid runs on the batch's stream: what this fit queued on its own (features, observations, start points) is there first
        HIPCHECK(hipStreamSynchronize(prob.slots[di][0].stream));
      }
      held_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
      if (host_lk.owns_lock()) host_lk.unlock();
      for (int di = 0; di < ndev; ++di) {
        if (fits_on[di].empty()) continue;
        HIPCHECK(hipSetDevice(ctx->devs[di]));
        batches[di] = small_batch_submit<T>(ctx->devs[di], prob.nu2, fits_on[di], arrival);
      }
      // every run's word, in run order.  The stream is asked now and then: a grid that is over without the word (a fault) must not
      // leave this thread spinning.
      auto wait_run = [&](int r, int di) {
        using namespace std::chrono;
        const volatile unsigned long long* q = &hruns[r].done;
        auto last_query = steady_clock::now();
        for (unsigned it = 1;; ++it) {
          if (*q == 1ull) break;
          if (it < 256) {
            __builtin_ia32_pause();
          } else {
            std::this_thread::sleep_for(microseconds(40));
            if (steady_clock::now() - last_query > milliseconds(20)) {
              last_query = steady_clock::now();
              const hipError_t e = hipStreamQuery(batches[di]->stream);
              if (e == hipSuccess) {
                if (*q == 1ull) break;
                throw HipError{hipErrorLaunchFailure, "small fit: the launch is over and a run never reported", __LINE__};
              }
              if (e != hipErrorNotReady) throw HipError{e, "hipStreamQuery (small-fit launch)", __LINE__};
            }
          }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
      };
      for (int r = 0; r < nruns; ++r) wait_run(r, ws[r].di);
      runs_over = true;
      int total_evals = 0, total_not_pd = 0, trace_n = 0;
      for (int r = 0; r < nruns; ++r) {
        RunWs& w = ws[r];
        HIPCHECK(hipSetDevice(ctx->devs[w.di]));
        Slot<T>& s = prob.slots[w.di][w.si];
        hipStream_t st = prob.slots[w.di][0].stream;
        const SmallFitResult hr = hruns[r].res;
        total_evals += hr.n_evals;
        total_not_pd += hr.n_not_pd;
        s.best_idx = hr.best_idx;
        s.best_lml = hr.best_lml;
        s.best_run = r;
        s.best_eval = hr.best_eval;
        s.best_theta.assign(hr.best_theta, hr.best_theta + p);
        s.best_params.assign(hr.best_params, hr.best_params + p);
        s.last_target = hr.best_idx < 0 ? 0 : hr.best_idx;
        if (cap > 0 && trace_n < opt.trace_cap) {  // the trace, run by run
          const int take = std::min(std::min(hr.n_evals, cap), opt.trace_cap - trace_n);
          if (opt.trace_theta) HIPCHECK(hipMemcpyAsync(opt.trace_theta + (size_t)trace_n * p, w.tr_theta, sizeof(double) * (size_t)take * p, hipMemcpyDeviceToHost, st));
          if (opt.trace_lml) HIPCHECK(hipMemcpyAsync(opt.trace_lml + trace_n, w.tr_lml, sizeof(double) * take, hipMemcpyDeviceToHost, st));
          if (opt.trace_grad) HIPCHECK(hipMemcpyAsync(opt.trace_grad + (size_t)trace_n * p, w.tr_grad, sizeof(double) * (size_t)take * p, hipMemcpyDeviceToHost, st));
          HIPCHECK(hipStreamSynchronize(st));
          if (opt.trace_run) for (int e = 0; e < take; ++e) opt.trace_run[trace_n + e] = r;
          trace_n += take;
        }
      }
      if (opt.trace_count) *opt.trace_count = trace_n;
      if (opt.n_evals) *opt.n_evals = total_evals;
      if (opt.n_not_pd) *opt.n_not_pd = total_not_pd;
      n_evals.store(total_evals);
    } catch (...) {
      for (int di = 0; di < ndev && di < (int)prob.slots.size(); ++di) {  // let the launched kernels finish before their workspaces go
        if (prob.slots[di].empty()) continue;
        (void)hipSetDevice(ctx->devs[di]);
        (void)hipStreamSynchronize(prob.slots[di][0].stream);
      }
      free_all();
      throw;
    }
    free_all();
  } else {
  // how many slots of each device are inside an optimiser run: the task-queue launches are sized for that (Problem::DagVariant)
  if (prob.busy_slots_)
    for (int di = 0; di < ndev; ++di) prob.busy_slots_[di].store(std::min(runs_on[di], max_conc));
  auto worker = [&](int di, int si) {
    struct Leave {
      std::atomic<int>* busy;
      std::atomic<int>* dev_busy;
      ~Leave() {
        if (busy) busy->fetch_sub(1);
        if (dev_busy) dev_busy->fetch_sub(1);
      }
    } leave{prob.busy_slots_ ? &prob.busy_slots_[di] : nullptr, ctx->devs[di] < MAX_DEVS ? &g_dev_busy[ctx->devs[di]] : nullptr};
    if (leave.dev_busy) leave.dev_busy->fetch_add(1);
    try {
      HIPCHECK(hipSetDevice(ctx->devs[di]));
      Slot<T>& s = prob.slots[di][si];
      const int nslots_here = std::min(runs_on[di], max_conc);
      // runs assigned to this device: r = di, di+ndev, ...; this worker takes every nslots_here-th of them
      int local = 0;
      for (int r = di; r < nruns; r += ndev, ++local) {
        if (local % nslots_here != si) continue;
        std::vector<double> x(p);
        if (r == 0) for (int i = 0; i < p; ++i) x[i] = theta0[i];
        else for (int i = 0; i < p; ++i) x[i] = starts[(size_t)(r - 1) * p + i];
        int eval_idx = 0;
        Objective obj = [&](const double* th, double* grad) -> double {
          theta_to_params(th, lo, hi, d, s.hP);
          const int target = (s.best_idx < 0) ? 0 : 1 - s.best_idx;
          double lml;
          const int st = prob.run_eval((size_t)di, si, target, true, true, &lml, grad);
          const int my_eval = eval_idx++;
          n_evals.fetch_add(1);
          if (st != HBEGP_OK) n_not_pd.fetch_add(1);
          if (opt.trace_cap > 0) {
            std::lock_guard<std::mutex> lk(trace_mu);
            if (trace_n < opt.trace_cap) {
              const int tI = trace_n++;
              if (opt.trace_theta) memcpy(opt.trace_theta + (size_t)tI * p, th, sizeof(double) * p);
              if (opt.trace_lml) opt.trace_lml[tI] = lml;
              if (opt.trace_grad) memcpy(opt.trace_grad + (size_t)tI * p, grad, sizeof(double) * p);
              if (opt.trace_run) opt.trace_run[tI] = r;
            }
          }
          if (st != HBEGP_OK) return std::numeric_limits<double>::infinity();  // fit.rs:105-112
          // capture (fit.rs:116-125): strictly greater lml wins; ties keep the earlier (run, eval)
          if (s.best_idx < 0 || lml > s.best_lml) {
            s.best_idx = target;
            s.best_lml = lml;
            s.best_run = r;
            s.best_eval = my_eval;
            s.best_theta.assign(th, th + p);
          }
          for (int j = 0; j < p; ++j) grad[j] = -grad[j];  // fit.rs:128-133
          return -lml;
        };
        LbfgsOptions lo_opt;
        lo_opt.maxeval = opt.maxeval;
        lo_opt.fixed_work = opt.fixed_work != 0;
        if (opt.lbfgs_memory > 0) lo_opt.memory = opt.lbfgs_memory;
        lbfgsb_minimize(obj, x.data(), lnlo.data(), lnhi.data(), p, lo_opt);
      }
    } catch (const HipError& he) {
      hip_fail(he);
      std::lock_guard<std::mutex> lk(err_mu);
      err = g_last_error;
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(err_mu);
      err = std::string("worker thread: ") + e.what();
    }
  };

  std::vector<std::thread> threads;
  try {
    for (int di = 0; di < ndev; ++di)
      for (int si = 0; si < std::min(runs_on[di], max_conc); ++si) threads.emplace_back(worker, di, si);
  } catch (...) {
    // thread creation failed part-way (std::system_error): the workers already started own device state -- let them
    // finish before the error leaves this frame (destroying a joinable std::thread terminates the process)
    for (auto& t : threads) t.join();
    throw;
  }
  for (auto& t : threads) t.join();
  // the workers append to the trace as their evaluations complete; hand it back in (run, eval) order
  if (opt.trace_cap > 0 && trace_n > 1 && opt.trace_run) {
    std::vector<int> order(trace_n);
    for (int i = 0; i < trace_n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return opt.trace_run[a] < opt.trace_run[b]; });
    auto permute = [&](auto* arr, int width) {
      if (!arr) return;
      std::vector<typename std::remove_pointer<decltype(arr)>::type> tmp((size_t)trace_n * width);
      for (int i = 0; i < trace_n; ++i) memcpy(&tmp[(size_t)i * width], arr + (size_t)order[i] * width, sizeof(tmp[0]) * width);
      memcpy(arr, tmp.data(), sizeof(tmp[0]) * tmp.size());
    };
    permute(opt.trace_theta, p);
    permute(opt.trace_lml, 1);
    permute(opt.trace_grad, p);
    permute(opt.trace_run, 1);
  }
  if (opt.trace_count) *opt.trace_count = trace_n;
  if (opt.n_evals) *opt.n_evals = n_evals.load();
  if (opt.n_not_pd) *opt.n_not_pd = n_not_pd.load();
  if (!err.empty()) return fail(HBEGP_EHIP, "%s", err.c_str());
  }  // host-driven optimiser runs

  // global arg-max over the per-slot captures, ties -> lowest (run, eval)
  int bdi = -1, bsi = -1;
  for (int di = 0; di < ndev; ++di)
    for (int si = 0; si < (int)prob.slots[di].size(); ++si) {
      const Slot<T>& s = prob.slots[di][si];
      if (s.best_idx < 0) continue;
      bool better = bdi < 0;
      if (!better) {
        const Slot<T>& b = prob.slots[bdi][bsi];
        better = s.best_lml > b.best_lml ||
                 (s.best_lml == b.best_lml && (s.best_run < b.best_run || (s.best_run == b.best_run && s.best_eval < b.best_eval)));
      }
      if (better) { bdi = di; bsi = si; }
    }
  if (bdi < 0) return fail(HBEGP_ALL_FAILED, "every evaluation of the fit failed (kernel matrix not positive definite)");
  Slot<T>& best = prob.slots[bdi][bsi];
  // clamp the captured parameters (fit.rs:155-164)
  std::vector<double> th(p);
  for (int i = 0; i < p; ++i) {
    double v = std::exp(best.best_theta[i]);
    if (v < lo[i]) v = lo[i];
    if (hi[i] < v) v = hi[i];
    th[i] = std::log(v);
  }
  if (theta_best) memcpy(theta_best, th.data(), sizeof(double) * p);
  if (lml_best) *lml_best = best.best_lml;
  const auto tf2 = std::chrono::steady_clock::now();
  if (host_serial && prob.small_ && !host_lk.owns_lock()) host_lk.lock();
  const auto th1 = std::chrono::steady_clock::now();
  if (model_out) *model_out = make_model<T>(prob, (size_t)bdi, bsi, th.data(), best.best_lml, false, best.best_params.empty() ? nullptr : best.best_params.data());
  if (host_lk.owns_lock()) {
    prob.release();  // (again, harmlessly, when the problem goes out of scope)
    held_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th1).count();
    host_lk.unlock();
    if (timing) fprintf(stderr, "fit: host-side turns held for %.3f ms (set-up + model / release)\n", held_ms);
  }
  if (timing) {
    const auto tf3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "fit: problem %.2f ms, optimiser runs %.2f ms (%d evaluations), model %.2f ms\n", ms(tf0, tf1), ms(tf1, tf2), n_evals.load(), ms(tf2, tf3));
  }
  return HBEGP_OK;
}

template <typename T>
static int do_extend(hbegp_ctx* ctx, const T* X, const T* y, int n, int d, double nu, const double* theta, const double* lo,
                     const double* hi, hbegp_model** model_out) {
  hbegp_ctx one;
  one.devs = {ctx->devs[0]};
  // One order of operations for one theta: the evaluation runs the way a fit's evaluations do at this size -- through the task
  // queue from 6 blocks on (its K^-1 split gives the undivided tiles' bits, DAGF_CINIT), as ad-hoc launches below that (a tile's
  // arithmetic does not depend on how the launch is scheduled).
  const int dag_env = env_int("HBEGP_DAG", -1);
  const bool queue_like_fit = (dag_env < 0 ? round_up(n, NB) / NB >= env_int("HBEGP_DAG_MIN_BLOCKS", DAG_MIN_BLOCKS_FIT) : dag_env != 0) && round_up(n, NB) / NB >= 2;
  Problem<T> prob(&one, X, y, n, d, nu, 1, !queue_like_fit && env_int("HBEGP_EXTEND_SCHED", 0) == 0, true);
  const int p = d + 2;
  Slot<T>& s = prob.slots[0][0];
  theta_to_params(theta, lo, hi, d, s.hP);
  double lml;
  const int st = prob.run_eval(0, 0, 0, false, false, &lml, nullptr);
  if (st != HBEGP_OK) return fail(HBEGP_NOT_PD, "Kernel matrix must be invertible.");  // fit.rs:55
  s.best_idx = 0;
  std::vector<double> th(theta, theta + p);
  if (lo && hi)
    for (int i = 0; i < p; ++i) {
      double v = std::exp(theta[i]);
      if (v < lo[i]) v = lo[i];
      if (hi[i] < v) v = hi[i];
      th[i] = std::log(v);
    }
  if (model_out) *model_out = make_model<T>(prob, 0, 0, th.data(), lml, true);
  return HBEGP_OK;
}

// extend with a prior model fitted on a prefix of the rows (minimize.rs:629-644 appends the validation samples to the
// data the last model was built from): same theta, incremental factorisation; falls back to the full path when the
// prefix does not match or is shorter than one 128-block.
template <typename T>
static int do_extend_from(hbegp_ctx* ctx, hbegp_model* prior, const T* X, const T* y, int n, hbegp_model** model_out,
                          int* incremental) {
  if (incremental) *incremental = 0;
  const int d = prior->d, p = d + 2;
  const double nu = prior->nu2 == 0 ? std::numeric_limits<double>::infinity() : prior->nu2 / 2.0;
  if (prior->dev != ctx->devs[0] || n < prior->n || prior->n < NB || !prior->ldiag)
    return do_extend<T>(ctx, X, y, n, d, nu, prior->theta.data(), nullptr, nullptr, model_out);
  hbegp_ctx one;
  one.devs = {ctx->devs[0]};
  static const int timing = env_int("HBEGP_TIMING", 0);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  Problem<T> prob(&one, X, y, n, d, nu, 1, true);
  const auto t1 = now();
  Slot<T>& s = prob.slots[0][0];
  theta_to_params(prior->theta.data(), nullptr, nullptr, d, s.hP);
  int st;
  {
    std::lock_guard<std::mutex> lock(prior->mu);
    st = prob.extend_from(0, 0, static_cast<const T*>(prior->X), static_cast<const T*>(prior->Xinv),
                          static_cast<const T*>(prior->Kinv), static_cast<const T*>(prior->ldiag), prior->n, prior->np);
  }
  if (st == HBEGP_EINVAL) return do_extend<T>(ctx, X, y, n, d, nu, prior->theta.data(), nullptr, nullptr, model_out);
  if (st != HBEGP_OK) return fail(HBEGP_NOT_PD, "Kernel matrix must be invertible.");  // fit.rs:55
  s.best_idx = 0;
  if (incremental) *incremental = 1;
  (void)p;
  const auto t2 = now();
  if (model_out) *model_out = make_model<T>(prob, 0, 0, prior->theta.data(), s.hOut->lml, true);
  const auto t3 = now();
  if (timing) fprintf(stderr, "extend_from: problem %.3f ms, factor %.3f ms, model %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
  return HBEGP_OK;
}

static int check_args(hbegp_ctx* ctx, const void* X, const void* y, int n, int d, double nu) {
  if (!ctx) return fail(HBEGP_EINVAL, "ctx is NULL");
  if (!X || !y) return fail(HBEGP_EINVAL, "X/y is NULL");
  if (n < 1) return fail(HBEGP_EINVAL, "n must be >= 1 (got %d)", n);
  if (d < 1 || d > MAXD) return fail(HBEGP_EINVAL, "d must be in 1..%d (got %d)", MAXD, d);
  if (!(nu == 0.5 || nu == 1.5 || nu == 2.5 || (std::isinf(nu) && nu > 0)))  // infinity: squared exponential (extension)
    return fail(HBEGP_EINVAL, "Matern kernel with arbitrary values for nu is unimplemented (got %g)", nu);  // matern_kernel.rs:79
  return HBEGP_OK;
}

#define GUARD_BEGIN try {
#define GUARD_END                                   \
  }                                                 \
  catch (const HipError& he) { return hip_fail(he); } \
  catch (const std::bad_alloc&) { return fail(HBEGP_ENOMEM, "host allocation failed"); } \
  catch (const std::exception& e) { return fail(HBEGP_EHIP, "internal error: %s", e.what()); } \
  catch (...) { return fail(HBEGP_EHIP, "internal error: unknown exception"); }

// ---------------------------------------------------------------------------------------------------------------
extern "C" {

int hbegp_version(void) { return HBEGP_VERSION; }

int hbegp_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < cnt; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

const char* hbegp_last_error(void) { return g_last_error.c_str(); }

int hbegp_ctx_create(int n_devices, const int* device_ids, hbegp_ctx** out) {
  if (!out || n_devices < 1) return fail(HBEGP_EINVAL, "n_devices must be >= 1");
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1)
    return fail(HBEGP_ENODEV, "no HIP device available: libhbegp has no CPU fallback");
  GUARD_BEGIN
  std::unique_ptr<hbegp_ctx> ctx(new hbegp_ctx());
  for (int i = 0; i < n_devices; ++i) {
    const int id = device_ids ? device_ids[i] : i;
    if (id < 0 || id >= cnt) return fail(HBEGP_ENODEV, "device %d not present (%d visible)", id, cnt);
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(HBEGP_ENODEV, "device %d is %s; this library only carries gfx950 code", id, prop.gcnArchName);
    HIPCHECK(hipSetDevice(id));
    if (env_int("HBEGP_SPIN", 0)) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    init_kernels();
    ctx->devs.push_back(id);
  }
  *out = ctx.release();
  g_live_ctx.fetch_add(1);
  return HBEGP_OK;
  GUARD_END
}
void hbegp_ctx_destroy(hbegp_ctx* ctx) {
  if (!ctx) return;
  delete ctx;
  if (g_live_ctx.fetch_sub(1) == 1) {
    g_pool.trim();
    g_host_pool.trim();
    g_stream_pool.trim();
  }
}

int hbegp_problem_create_f64(hbegp_ctx* ctx, const double* X, const double* y, int n, int d, double nu, int n_slots,
                             hbegp_problem** out) {
  if (int e = check_args(ctx, X, y, n, d, nu)) return e;
  if (!out || n_slots < 1) return fail(HBEGP_EINVAL, "bad out/n_slots");
  GUARD_BEGIN
  std::unique_ptr<hbegp_problem> p(new hbegp_problem());
  p->impl.reset(new Problem<double>(ctx, X, y, n, d, nu, n_slots));
  *out = p.release();
  return HBEGP_OK;
  GUARD_END
}
int hbegp_problem_create_f32(hbegp_ctx* ctx, const float* X, const float* y, int n, int d, double nu, int n_slots,
                             hbegp_problem** out) {
  if (int e = check_args(ctx, X, y, n, d, nu)) return e;
  if (!out || n_slots < 1) return fail(HBEGP_EINVAL, "bad out/n_slots");
  GUARD_BEGIN
  std::unique_ptr<hbegp_problem> p(new hbegp_problem());
  p->impl.reset(new Problem<float>(ctx, X, y, n, d, nu, n_slots));
  *out = p.release();
  return HBEGP_OK;
  GUARD_END
}
void hbegp_problem_destroy(hbegp_problem* prob) { delete prob; }

int hbegp_problem_eval(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo, const double* hi,
                       double* lml, double* grad) {
  if (!prob || !theta || !lml) return fail(HBEGP_EINVAL, "NULL argument");
  GUARD_BEGIN
  return prob->impl->eval(dev, slot, theta, lo, hi, lml, grad);
  GUARD_END
}

int hbegp_problem_time_concurrent(hbegp_problem* prob, int dev, const double* theta, int reps, double* out) {
  if (!prob || !theta || !out || reps < 1) return fail(HBEGP_EINVAL, "bad argument");
  GUARD_BEGIN
  return prob->impl->time_concurrent(dev, theta, reps, out);
  GUARD_END
}

int hbegp_problem_time_eval(hbegp_problem* prob, int dev, int slot, const double* theta, int reps, double* phase_ms) {
  if (!prob || !theta || reps < 1) return fail(HBEGP_EINVAL, "bad argument");
  GUARD_BEGIN
  return prob->impl->time_eval(dev, slot, theta, reps, phase_ms);
  GUARD_END
}

}  // extern "C"
template <typename T>
static int problem_get(hbegp_problem* prob, int dev, int slot, T* alpha, T* kinv, T* ldiag) {
  if (!prob) return fail(HBEGP_EINVAL, "NULL problem");
  auto* p = dynamic_cast<Problem<T>*>(prob->impl.get());
  if (!p) return fail(HBEGP_EINVAL, "element type mismatch");
  if (dev < 0 || dev >= (int)p->slots.size() || slot < 0 || slot >= p->n_slots) return fail(HBEGP_EINVAL, "bad device/slot index");
  GUARD_BEGIN
  Slot<T>& s = p->slots[dev][slot];
  HIPCHECK(hipSetDevice(s.dev));
  const int b = s.last_target, n = p->n, np = p->np;
  // (copies on the slot's own stream: a null-stream copy fails while another host thread captures a graph)
  if (alpha) HIPCHECK(hipMemcpyAsync(alpha, s.alpha[b], sizeof(T) * n, hipMemcpyDeviceToHost, s.stream));
  if (ldiag) HIPCHECK(hipMemcpyAsync(ldiag, s.ldiag, sizeof(T) * n, hipMemcpyDeviceToHost, s.stream));
  HIPCHECK(hipStreamSynchronize(s.stream));
  if (kinv) {
    std::vector<T> tmp((size_t)np * np);
    HIPCHECK(hipMemcpyAsync(tmp.data(), s.Kinv[b], sizeof(T) * tmp.size(), hipMemcpyDeviceToHost, s.stream));
    HIPCHECK(hipStreamSynchronize(s.stream));
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) kinv[(size_t)i * n + j] = kinv[(size_t)j * n + i] = tmp[(size_t)i * np + j];
  }
  return HBEGP_OK;
  GUARD_END
}
template <typename T>
static int problem_debug_get(hbegp_problem* prob, int dev, int slot, int which, T* out) {
  if (!prob || !out) return fail(HBEGP_EINVAL, "NULL argument");
  auto* p = dynamic_cast<Problem<T>*>(prob->impl.get());
  if (!p) return fail(HBEGP_EINVAL, "element type mismatch");
  if (dev < 0 || dev >= (int)p->slots.size() || slot < 0 || slot >= p->n_slots || which < 1 || which > 4)
    return fail(HBEGP_EINVAL, "bad device/slot/which");
  GUARD_BEGIN
  Slot<T>& s = p->slots[dev][slot];
  HIPCHECK(hipSetDevice(s.dev));
  HIPCHECK(hipStreamSynchronize(s.stream));
  const T* src = which == 1 ? s.W1 : (which == 2 ? s.W2 : (which == 3 ? s.W3 : s.Kinv[s.last_target]));
  if (!src) return fail(HBEGP_EINVAL, "this problem has no such work matrix");
  HIPCHECK(hipMemcpyAsync(out, src, sizeof(T) * (size_t)p->np * p->np, hipMemcpyDeviceToHost, s.stream));
  HIPCHECK(hipStreamSynchronize(s.stream));
  return HBEGP_OK;
  GUARD_END
}
extern "C" {
int hbegp_problem_debug_get_f64(hbegp_problem* prob, int dev, int slot, int which, double* out) {
  return problem_debug_get<double>(prob, dev, slot, which, out);
}
int hbegp_problem_debug_get_f32(hbegp_problem* prob, int dev, int slot, int which, float* out) {
  return problem_debug_get<float>(prob, dev, slot, which, out);
}
int hbegp_problem_get_f64(hbegp_problem* prob, int dev, int slot, double* alpha, double* kinv, double* ldiag) {
  return problem_get<double>(prob, dev, slot, alpha, kinv, ldiag);
}
int hbegp_problem_get_f32(hbegp_problem* prob, int dev, int slot, float* alpha, float* kinv, float* ldiag) {
  return problem_get<float>(prob, dev, slot, alpha, kinv, ldiag);
}

}  // extern "C"
template <typename T>
static int problem_kmat(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo, const double* hi, T* K) {
  if (!prob || !theta || !K) return fail(HBEGP_EINVAL, "NULL argument");
  auto* p = dynamic_cast<Problem<T>*>(prob->impl.get());
  if (!p) return fail(HBEGP_EINVAL, "element type mismatch");
  if (dev < 0 || dev >= (int)p->slots.size() || slot < 0 || slot >= p->n_slots) return fail(HBEGP_EINVAL, "bad device/slot index");
  GUARD_BEGIN
  Slot<T>& s = p->slots[dev][slot];
  HIPCHECK(hipSetDevice(s.dev));
  theta_to_params(theta, lo, hi, p->d, s.hP);
  HIPCHECK(hipMemcpyAsync(s.dP, s.hP, sizeof(EvalParams), hipMemcpyHostToDevice, s.stream));
  launch_reset_out(s.dOut, s.stream);
  launch_kmat<T>(p->Xd[dev], p->n, p->d, p->np, p->nu2, s.dP, s.W1, &s.dOut->info, s.stream);
  CHECK_LAUNCHES();
  const int n = p->n, np = p->np;
  std::vector<T> tmp((size_t)np * np);
  HIPCHECK(hipMemcpyAsync(tmp.data(), s.W1, sizeof(T) * tmp.size(), hipMemcpyDeviceToHost, s.stream));
  HIPCHECK(hipStreamSynchronize(s.stream));
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) K[(size_t)i * n + j] = K[(size_t)j * n + i] = tmp[(size_t)i * np + j];
  return HBEGP_OK;
  GUARD_END
}
extern "C" {
int hbegp_problem_kmat_f64(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo, const double* hi,
                           double* K) {
  return problem_kmat<double>(prob, dev, slot, theta, lo, hi, K);
}
int hbegp_problem_kmat_f32(hbegp_problem* prob, int dev, int slot, const double* theta, const double* lo, const double* hi,
                           float* K) {
  return problem_kmat<float>(prob, dev, slot, theta, lo, hi, K);
}

int hbegp_fit_f64(hbegp_ctx* ctx, const double* X, const double* y, int n, int d, double nu, const double* theta0,
                  const double* lo, const double* hi, const double* starts, int n_restarts, const hbegp_fit_options* opt,
                  double* theta_best, double* lml_best, hbegp_model** model) {
  {
    hbegp_fit_options probe;
    if (int e = read_fit_options(opt, &probe)) return e;  // before anything else: a mis-sized struct is a build problem
  }
  if (int e = check_args(ctx, X, y, n, d, nu)) return e;
  if (!theta0 || !lo || !hi || (n_restarts > 0 && !starts)) return fail(HBEGP_EINVAL, "theta0/lo/hi/starts is NULL");
  GUARD_BEGIN
  return do_fit<double>(ctx, X, y, n, d, nu, theta0, lo, hi, starts, n_restarts, opt, theta_best, lml_best, model);
  GUARD_END
}
int hbegp_fit_f32(hbegp_ctx* ctx, const float* X, const float* y, int n, int d, double nu, const double* theta0,
                  const double* lo, const double* hi, const double* starts, int n_restarts, const hbegp_fit_options* opt,
                  double* theta_best, double* lml_best, hbegp_model** model) {
  {
    hbegp_fit_options probe;
    if (int e = read_fit_options(opt, &probe)) return e;  // before anything else: a mis-sized struct is a build problem
  }
  if (int e = check_args(ctx, X, y, n, d, nu)) return e;
  if (!theta0 || !lo || !hi || (n_restarts > 0 && !starts)) return fail(HBEGP_EINVAL, "theta0/lo/hi/starts is NULL");
  GUARD_BEGIN
  return do_fit<float>(ctx, X, y, n, d, nu, theta0, lo, hi, starts, n_restarts, opt, theta_best, lml_best, model);
  GUARD_END
}

int hbegp_extend_f64(hbegp_ctx* ctx, const double* X, const double* y, int n, int d, double nu, const double* theta,
                     const double* lo, const double* hi, hbegp_model** model) {
  if (int e = check_args(ctx, X, y, n, d, nu)) return e;
  if (!theta) return fail(HBEGP_EINVAL, "theta is NULL");
  GUARD_BEGIN
  return do_extend<double>(ctx, X, y, n, d, nu, theta, lo, hi, model);
  GUARD_END
}
int hbegp_extend_f32(hbegp_ctx* ctx, const float* X, const float* y, int n, int d, double nu, const double* theta,
                     const double* lo, const double* hi, hbegp_model** model) {
  if (int e = check_args(ctx, X, y, n, d, nu)) return e;
  if (!theta) return fail(HBEGP_EINVAL, "theta is NULL");
  GUARD_BEGIN
  return do_extend<float>(ctx, X, y, n, d, nu, theta, lo, hi, model);
  GUARD_END
}

int hbegp_extend_from_f64(hbegp_ctx* ctx, hbegp_model* prior, const double* X, const double* y, int n, hbegp_model** model,
                          int* incremental) {
  if (!ctx || !prior || !X || !y) return fail(HBEGP_EINVAL, "ctx/prior/X/y is NULL");
  if (prior->is_f32) return fail(HBEGP_EINVAL, "prior model holds f32 data");
  if (n < 1) return fail(HBEGP_EINVAL, "n must be >= 1 (got %d)", n);
  GUARD_BEGIN
  return do_extend_from<double>(ctx, prior, X, y, n, model, incremental);
  GUARD_END
}
int hbegp_extend_from_f32(hbegp_ctx* ctx, hbegp_model* prior, const float* X, const float* y, int n, hbegp_model** model,
                          int* incremental) {
  if (!ctx || !prior || !X || !y) return fail(HBEGP_EINVAL, "ctx/prior/X/y is NULL");
  if (!prior->is_f32) return fail(HBEGP_EINVAL, "prior model holds f64 data");
  if (n < 1) return fail(HBEGP_EINVAL, "n must be >= 1 (got %d)", n);
  GUARD_BEGIN
  return do_extend_from<float>(ctx, prior, X, y, n, model, incremental);
  GUARD_END
}

int hbegp_predict_f64(hbegp_model* model, const double* Xs, int m, double* mean, double* var, int* n_warn) {
  if (!model || !Xs || !mean || m < 0) return fail(HBEGP_EINVAL, "bad argument");
  if (model->is_f32) return fail(HBEGP_EINVAL, "model holds f32 data");
  if (m == 0) { if (n_warn) *n_warn = 0; return HBEGP_OK; }
  GUARD_BEGIN
  return model_predict<double>(model, Xs, m, mean, var, n_warn);
  GUARD_END
}
int hbegp_predict_f32(hbegp_model* model, const float* Xs, int m, float* mean, float* var, int* n_warn) {
  if (!model || !Xs || !mean || m < 0) return fail(HBEGP_EINVAL, "bad argument");
  if (!model->is_f32) return fail(HBEGP_EINVAL, "model holds f64 data");
  if (m == 0) { if (n_warn) *n_warn = 0; return HBEGP_OK; }
  GUARD_BEGIN
  return model_predict<float>(model, Xs, m, mean, var, n_warn);
  GUARD_END
}

int hbegp_model_info(const hbegp_model* model, int* n, int* d, int* is_f32, double* nu, double* lml) {
  if (!model) return fail(HBEGP_EINVAL, "NULL model");
  if (n) *n = model->n;
  if (d) *d = model->d;
  if (is_f32) *is_f32 = model->is_f32 ? 1 : 0;
  if (nu) *nu = model->nu2 == 0 ? std::numeric_limits<double>::infinity() : model->nu2 / 2.0;
  if (lml) *lml = model->lml;
  return HBEGP_OK;
}

}  // extern "C"
template <typename T>
static int model_get(hbegp_model* m, double* theta, T* alpha, T* kinv) {
  if (!m) return fail(HBEGP_EINVAL, "NULL model");
  if (m->is_f32 != (sizeof(T) == 4)) return fail(HBEGP_EINVAL, "element type mismatch");
  GUARD_BEGIN
  std::lock_guard<std::mutex> lock(m->mu);
  HIPCHECK(hipSetDevice(m->dev));
  if (theta) memcpy(theta, m->theta.data(), sizeof(double) * m->theta.size());
  if (alpha) HIPCHECK(hipMemcpyAsync(alpha, m->alpha, sizeof(T) * m->n, hipMemcpyDeviceToHost, m->stream));
  if (kinv) HIPCHECK(hipMemcpy2DAsync(kinv, sizeof(T) * m->n, m->Kinv, sizeof(T) * m->np, sizeof(T) * m->n, m->n, hipMemcpyDeviceToHost, m->stream));
  HIPCHECK(hipStreamSynchronize(m->stream));
  return HBEGP_OK;
  GUARD_END
}
extern "C" {
int hbegp_model_get_f64(hbegp_model* model, double* theta, double* alpha, double* kinv) {
  return model_get<double>(model, theta, alpha, kinv);
}
int hbegp_model_get_f32(hbegp_model* model, double* theta, float* alpha, float* kinv) {
  return model_get<float>(model, theta, alpha, kinv);
}
void hbegp_model_retain(hbegp_model* model) {
  if (model) model->refs.fetch_add(1);
}
void hbegp_model_release(hbegp_model* model) {
  if (model && model->refs.fetch_sub(1) == 1) delete model;
}

int hbegp_debug_lbfgs_replay(int n, const double* x0, const double* lo, const double* hi, int maxeval, int memory, int fixed_work,
                             int count, const double* f, const double* g, double* requested, int* n_requested) {
  if (n < 1 || n > LBFGS_MAXN || count < 0 || !x0 || !lo || !hi || !requested || !n_requested || (count > 0 && (!f || !g)))
    return fail(HBEGP_EINVAL, "bad argument");
  GUARD_BEGIN
  LbfgsOptions o;
  std::unique_ptr<LbfgsState> st(new LbfgsState);
  lbfgs_begin(*st, x0, lo, hi, n, maxeval, memory > 0 ? memory : o.memory, o.pgtol, o.ftol, fixed_work != 0);
  int produced = 0;
  memcpy(requested, lbfgs_request(*st), sizeof(double) * n);
  produced = 1;
  for (int i = 0; i < count; ++i) {
    if (!lbfgs_advance(*st, f[i], g + (size_t)i * n)) break;
    if (i + 1 < count) {
      memcpy(requested + (size_t)(i + 1) * n, lbfgs_request(*st), sizeof(double) * n);
      produced = i + 2;
    }
  }
  *n_requested = produced;
  return HBEGP_OK;
  GUARD_END
}

int hbegp_debug_dag_plan(int nblocks, int bk, int small_h, int nwg, int fine, int* ntasks, int* ncounters, int* nleaf,
                         double* gflop, double* crit_us, double* sim_us, char* err, int errlen) {
  if (nblocks < 1 || (bk != 16 && bk != 32) || small_h < 0 || nwg < 0) return fail(HBEGP_EINVAL, "bad argument");
  GUARD_BEGIN
  DagBuilder builder(bk, small_h, nwg, (fine & 1) != 0);
  builder.set_rl_progressive((fine & 16) != 0, -1, -1, false);
  builder.set_big128(getenv("HBEGP_DAG_BIG128") && atoi(getenv("HBEGP_DAG_BIG128")) != 0, getenv("HBEGP_DAG_BIG128") && atoi(getenv("HBEGP_DAG_BIG128")) >= 2);
  // bit 1: unused (rounds 2-4: kernel-matrix tiles and alpha / lml reductions as tasks too); bit 2: the K^-1 = X^T X tiles behind
  // the recursion; bit 3: the right-looking plan; bit 4: its row-progressive inverse and K^-1
  if (fine & 2) return fail(HBEGP_EINVAL, "fine bit 1 (kernel-matrix / reduction tasks in the queue) no longer exists");
  DagPlan plan = builder.build(0, nblocks, (fine & 4) != 0, (fine & 8) != 0);
  // fault injection for the validator's own test: HBEGP_DAG_TEST_FAULT = "drop:<i>" (task i loses its first wait) or
  // "move:<i>:<j>" (task i is moved to queue position j)
  if (const char* fault = getenv("HBEGP_DAG_TEST_FAULT")) {
    int a = 0, b = 0;
    if (sscanf(fault, "drop:%d", &a) == 1 && a >= 0 && a < (int)plan.tasks.size() && plan.tasks[a].nwait > 0) {
      DagTask& t = plan.tasks[a];
      for (int w = 1; w < t.nwait; ++w) { t.wcnt[w - 1] = t.wcnt[w]; t.wval[w - 1] = t.wval[w]; }
      t.nwait--;
    } else if (sscanf(fault, "move:%d:%d", &a, &b) == 2 && a >= 0 && b >= 0 && a < (int)plan.tasks.size() && b < (int)plan.tasks.size()) {
      const DagTask t = plan.tasks[a];
      plan.tasks.erase(plan.tasks.begin() + a);
      plan.tasks.insert(plan.tasks.begin() + b, t);
    }
  }
  const std::string why = plan.tasks.empty() ? std::string("too many counters") : dag_plan_validate(plan, nblocks);
  if (ntasks) *ntasks = (int)plan.tasks.size();
  if (ncounters) *ncounters = (int)plan.totals.size();
  if (nleaf) *nleaf = plan.n_leaf;
  if (gflop) *gflop = plan.gflop;
  if (crit_us) *crit_us = plan.crit_us;
  if (sim_us) *sim_us = plan.sim_us;
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", why.c_str());
  return why.empty() ? HBEGP_OK : fail(HBEGP_EINVAL, "%s", why.c_str());
  GUARD_END
}

double hbegp_minimize_by_gradient(hbegp_objective_fn f, void* user, double* x, const double* lo, const double* hi, int n,
                                  int maxeval) {
  if (!f || !x || !lo || !hi || n < 1) {
    fail(HBEGP_EINVAL, "hbegp_minimize_by_gradient: NULL argument or n < 1");
    return std::numeric_limits<double>::quiet_NaN();
  }
  try {
    LbfgsOptions opt;
    opt.maxeval = maxeval > 0 ? maxeval : 150;
    Objective obj = [&](const double* xx, double* g) { return f(xx, g, user); };
    return lbfgsb_minimize(obj, x, lo, hi, n, opt).f;
  } catch (const std::exception& e) {
    fail(HBEGP_EHIP, "hbegp_minimize_by_gradient: %s", e.what());
  } catch (...) {
    fail(HBEGP_EHIP, "hbegp_minimize_by_gradient: unknown exception");
  }
  return std::numeric_limits<double>::quiet_NaN();
}

}  // extern "C"
