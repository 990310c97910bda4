// dag_plan.hpp — host side of the device-scheduled factorisation (dag_kernel.inc.hpp): builds the ordered task queue
// and its dependency counters for the Cholesky + inverse-factor recursion on a range of 128-blocks, and checks it.
// Plain C++ (no HIP): also compiled into the CPU-only tests through hbegp_debug_dag_plan().
//
// Recursion (same as Problem::chol_inv_rec / chol_inv_split, lml.rs:47 + the inverse of the factor lml.rs:62 needs):
//   node N = [lo, hi), mid:   left subtree L;  T = A21 X11^T (W1,W2 -> W2[2,1]);  A22 -= T T^T (W2 -> W1, lower);
//                             U = T X11 (W2 -> W1[2,1]);  right subtree R;  X21 = -X22 U (W2,W1 -> W2[2,1])
// Dependencies are per 128-row block, so that the latency-bound chain of diagonal blocks runs ahead of (and beside) the
// bulk tiles -- the launch-per-product path can only run them one after the other.  Every wait is for the FULL count of a
// counter ("B waits on c" = "every task that bumps c happened before B"), which is what dag_plan_validate() relies on.
//   counters   leaf[k] | Trow[N][i] | Srow[N][i], Sall[N] | Uall[N] | Xrow[N][i], Xall[N]
//   rowfinal(S, j)  "row block j of X = L^-1 is final inside subtree S": Xrow[M][j] of the highest node M in S with j in
//                   its right half, or leaf[j] when j is S's first block
//   gate(i)         "row block i of the Schur complement is up to date": Srow[P][i] of the nearest ancestor P whose right
//                   half holds i (none for rows no ancestor updates)
//   leaf(k)        waits gate(k)
//   T(N)(i,j)      waits rowfinal(L, j) [X11 row j], gate(i) [A21 row i]                      bumps Trow[N][i]
//   SYRK(N)(i,j)   waits Trow[N][i], Trow[N][j]                                               bumps Srow[N][i], Sall[N]
//   U(N)(i,j)      waits Trow[N][i]  (whole rows of T imply all of X11; U overwrites A21(i,j), read by T's row i)
//                                                                                            bumps Uall[N]
//   X21(N)(i,j)    waits rowfinal(R, i) [X22 row i], Uall[N], Sall[N] (it overwrites T(i,j), read by SYRK and U)
//                                                                                            bumps Xrow[N][i], Xall[N]
// Queue order: a list schedule simulated on the host for the number of workgroups the launch will have (critical-path
// priority, estimated task times); workgroups pull in that order, so tasks tend to be pulled when their inputs are ready.
// Any topological order is deadlock-free for any number of resident workgroups; this one is also fast.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <string>
#include <vector>

#include "engine.hpp"

namespace hbegp {

// Where a node [lo, hi) of the recursion is split.  Default: in the middle, as the launch path does (same arithmetic, same
// bits).  HBEGP_SPLIT_NUM/HBEGP_SPLIT_DEN (experiment, task queue only): nodes wider than HBEGP_SPLIT_MIN blocks put num/den
// of their blocks into the LEFT part -- a smaller left part shortens the chain-bound head of the factorisation (nothing but
// the diagonal chain can run until the left part is done) at the price of a longer chain later.  Simulated for n=4096, 85
// workgroups: 1/2 -> 2932 us, 3/8 -> 2932, 1/3 -> 2887, 1/4 -> 3015, 1/8 -> 3045: nothing to gain, the default stays.
inline int dag_split_point(int lo, int hi) {
  static const int num = getenv("HBEGP_SPLIT_NUM") ? atoi(getenv("HBEGP_SPLIT_NUM")) : 1;
  static const int den = getenv("HBEGP_SPLIT_DEN") ? atoi(getenv("HBEGP_SPLIT_DEN")) : 2;
  static const int minw = getenv("HBEGP_SPLIT_MIN") ? atoi(getenv("HBEGP_SPLIT_MIN")) : 8;
  const int w = hi - lo;
  if (w <= minw || den <= 0 || num <= 0 || num >= den || 2 * num == den) return lo + w / 2;  // default: exactly the launch path's split
  int left = (w * num) / den;
  // keep the parts multiples of 4 blocks where possible (small nodes stay balanced binary trees)
  left = std::max(4, left / 4 * 4);
  return lo + std::min(left, w - 1);
}

struct DagPlan {
  std::vector<DagTask> tasks;
  std::vector<int> totals;  // per counter: number of tasks that bump it
  double gflop = 0;         // algorithmic flops of the tile products (2*128^3 per pair of 128-blocks, half on triangles)
  double gflop_lauum = 0;   // of which: the K^-1 = X^T X tiles (lauum = true)
  int n_lauum = 0;
  int n_leaf = 0;
  double sim_us = 0;        // makespan of the simulated schedule (estimate)
  double crit_us = 0;       // critical path of the graph under the same estimates
};

struct DagGate {
  int cnt = -1, val = 0;  // cnt < 0: nothing to wait for
};

// estimated task times in microseconds (MI355X, one workgroup of 512 threads per CU; measured round 2)
struct DagCosts {
  // from task traces (tools/dag_trace.py, n=4096 f64): diagonal block 28.3 us; tile = 1.5 us + 3.9 (64x64) / 7.9 (128x64) us
  // per 128 elements of contraction (81-84 % of one CU's fp64 MFMA peak); counter bump -> dependent task running ~4 us
  double leaf = 28.5 + 4.0;
  double overhead = 1.5 + 4.0;
  double per128_big = 7.9;
  double per128_small = 3.9;
  double per128_big128 = 15.0;  // 128x128 tile
  double per128_chain = 1.5;  // 32x64 one-shot tile: one dependent MFMA chain per wave
};

class DagBuilder {
 public:
  // bk: contraction elements per pipeline stage (16 for f64, 32 for f32): ranges are whole stages
  // small_h: nodes whose halves are at most this many 128-blocks wide use 64x64 tiles (latency-bound products)
  // nwg: workgroups the queue is ordered for (0: keep the recursion's order)
  // crit_rows: in the big nodes, this many block rows next to the diagonal chain (the first rows of T and of the Schur
  //            update, the last rows of X21) also use 64x64 tiles: they sit on the critical path, where a tile's time counts
  DagBuilder(int bk, int small_h, int nwg = 0, bool fine = true, int crit_rows = 1)
      : bk_(bk), small_h_(small_h), nwg_(nwg), fine_(fine), crit_rows_(crit_rows) {}

  // full = true: the kernel-matrix tiles in front of the recursion and the alpha / lml reductions behind it are tasks of the
  // same queue (whole matrix only: blo = 0, bhi = np / 128)
  // right-looking plan: widest column range of a grouped update (measured at n = 4096, fit+predict/s: 4 -> 1.50, 8 -> 1.53,
  // 16 -> 1.53, 32 -> 1.57) and how many block columns ahead of the chain are updated column by column (1 -> 1.57, 2 -> 1.53,
  // 3 -> 1.51; one evaluation alone: 2.21 / 2.19 ms)
  void set_rl(int group, int near, bool lauum_split = true, bool chain32 = true) {
    rl_group_ = std::max(1, group); rl_near_ = std::max(1, near); rl_lauum_split_ = lauum_split; rl_chain32_ = chain32;
  }
  // right-looking plan, round 4: the inverse of the factor and K^-1 follow the diagonal chain row by row (rl_progressive)
  // instead of by divide and conquer behind it
  void set_rl_progressive(bool on, int unear = -1, int knear = -1, bool small_tiles = false) {
    rl_prog_ = on; prog_unear_ = unear; prog_knear_ = knear; prog_small_ = small_tiles;
  }
  // lauum = true: the tiles of K^-1 = X^T X (lml.rs:62) follow the recursion in the same queue (whole matrix only)
  // rl = true: right-looking tile Cholesky + recursive inverse of the factor (build_rl) instead of the recursion that
  // carries the inverse (whole matrix only; the factor L lives in W3)
  DagPlan build(int blo, int bhi, bool lauum = false, bool rl = false) {
    plan_ = DagPlan();
    top_lo_ = blo; top_hi_ = bhi;
    if (rl && blo == 0) {
      lauum_split_ = rl_lauum_split_;
      build_rl(bhi, lauum);
    } else {
      const Sub root = rec(blo, bhi, nullptr);
      if (lauum && blo == 0) build_lauum(bhi, root);
    }
    if (plan_.totals.size() >= 0xffff) plan_.tasks.clear();  // counter ids are 16-bit: the caller falls back
    for (int tot : plan_.totals)
      if (tot > 0xffff) plan_.tasks.clear();
    if (nwg_ > 0 && !plan_.tasks.empty()) order();
    return plan_;
  }

 private:
  int bk_, small_h_, nwg_;
  bool fine_;
  int crit_rows_ = 1;
  int rl_group_ = 32, rl_near_ = 1;
  bool rl_lauum_split_ = true;
  bool big128_ = false;      // 128x128 tiles for the bulk products (never for a continued sum)
  bool big128_acc_ = false;  // ... also with beta = 1 (the old values fetched in the epilogue)
  bool rl_prog_ = false;     // right-looking plan: row-progressive inverse and K^-1 (rl_progressive)
  int prog_unear_ = -1, prog_knear_ = -1;  // single rows at the end of the U / K^-1 range lists (-1: rl_near_)
  bool prog_small_ = false;  // 64x64 tiles for the rows of X and the last U updates before a row (measured: no gain, more tasks)
  bool rl_chain32_ = true;   // right-looking plan: the two products between consecutive diagonal blocks as 32x64 one-shot tiles
  DagPlan plan_;
  DagCosts cost_;

  int new_counter() {
    plan_.totals.push_back(0);
    return (int)plan_.totals.size() - 1;
  }
  void push(DagTask t, const std::vector<DagGate>& waits, int sig0, int sig1, double cost_us, int sig2 = -1) {
    t.nwait = 0;
    for (const DagGate& g : waits) {
      if (g.cnt < 0) continue;
      bool dup = false;
      for (int w = 0; w < t.nwait; ++w) dup = dup || t.wcnt[w] == g.cnt;
      if (dup) continue;
      t.wcnt[t.nwait] = (uint16_t)g.cnt;
      t.wval[t.nwait] = (uint16_t)g.val;
      ++t.nwait;
    }
    t.sig[0] = sig0 < 0 ? DAG_NOSIG : (uint16_t)sig0;
    t.sig[1] = sig1 < 0 ? DAG_NOSIG : (uint16_t)sig1;
    t.sig[2] = sig2 < 0 ? DAG_NOSIG : (uint16_t)sig2;
    if (sig2 >= 0) plan_.totals[sig2]++;
    if (sig0 >= 0) plan_.totals[sig0]++;
    if (sig1 >= 0) plan_.totals[sig1]++;
    t.cost = (uint16_t)std::min(65535.0, cost_us * 10.0);
    plan_.tasks.push_back(t);
  }

  struct Op {
    uint16_t flags;
    int r0, r1, c0, c1;  // output block rows / cols (128-blocks)
    bool lower;          // only blocks with row >= col (diagonal blocks: lower 64-tiles only)
    int k0, k1;          // contraction block range
    int klim;            // 0 none | 1: k <= tj | 2: k >= tj | 3: k <= ti   (64-element units, as gemm_kernel's 64-tile)
    bool tri_a, tri_b;   // operand is triangular (flop accounting only)
    int crit;            // 0 none | 1: the first crit_rows_ block rows are on the critical path | 2: the last ones
  };
  struct Tile { int kind, bi, bj, row0, col0, ka, kb; };

  // the tiles of one product, block row by block row (deepest contraction first inside a row)
  // chain: the product sits between two consecutive diagonal blocks (right-looking plan): 32x64 one-shot tiles (DAG_GEMM_32x64)
  std::vector<Tile> tiles_of(const Op& op, bool small, bool chain = false) {
    std::vector<Tile> tiles;
    for (int bi = op.r0; bi < op.r1; ++bi) {
      const size_t row_begin = tiles.size();
      for (int bj = op.c0; bj < op.c1; ++bj) {
        if (op.lower && bj > bi) continue;
        const bool diag = op.lower && bi == bj;
        for (int hj = 0; hj < 2; ++hj) {
          const int tj = 2 * bj + hj;  // 64-column unit
          auto krange = [&](int ti_last, int ti_first, int* ka, int* kb) {
            *ka = 2 * op.k0; *kb = 2 * op.k1;
            if (op.klim == 1) *kb = std::min(*kb, tj + 1);
            if (op.klim == 2) *ka = std::max(*ka, tj);
            if (op.klim == 3) *kb = std::min(*kb, ti_last + 1);
            if (op.klim == 4) *ka = std::max(*ka, ti_first);
          };
          const bool crit_row = (op.crit == 1 && bi < op.r0 + crit_rows_) || (op.crit == 2 && bi >= op.r1 - crit_rows_);
          if (small || diag || crit_row) {
            for (int hi = 0; hi < 2; ++hi) {
              const int ti = 2 * bi + hi;
              if (diag && ti < tj) continue;  // strictly upper 64-tile of a symmetric result
              int ka, kb;
              krange(ti, ti, &ka, &kb);
              if (chain && !(op.flags & (DAGF_AKM | DAGF_BKM | DAGF_CINIT | DAGF_CKINV))) {
                for (int h32 = 0; h32 < 2; ++h32) tiles.push_back({DAG_GEMM_32x64, bi, bj, ti * 64 + 32 * h32, tj * 64, ka * 64, kb * 64});
              } else {
                tiles.push_back({DAG_GEMM_64x64, bi, bj, ti * 64, tj * 64, ka * 64, kb * 64});
              }
            }
          } else if (big128_ && !(op.flags & DAGF_CINIT) && (big128_acc_ || !(op.flags & DAGF_ACC))) {
            // one 128x128 tile for both column halves: the union of their contraction ranges (the extra range of one half meets
            // the zeros of a triangular operand, as for the two row halves)
            if (hj == 0) {
              int ka0, kb0, ka1, kb1;
              krange(2 * bi + 1, 2 * bi, &ka0, &kb0);
              const int tj_keep = tj;
              (void)tj_keep;
              // second half: the same formulas with tj + 1
              ka1 = 2 * op.k0; kb1 = 2 * op.k1;
              if (op.klim == 1) kb1 = std::min(kb1, tj + 2);
              if (op.klim == 2) ka1 = std::max(ka1, tj + 1);
              if (op.klim == 3) kb1 = std::min(kb1, 2 * bi + 2);
              if (op.klim == 4) ka1 = std::max(ka1, 2 * bi);
              tiles.push_back({DAG_GEMM_128x128, bi, bj, bi * 128, tj * 64, std::min(ka0, ka1) * 64, std::max(kb0, kb1) * 64});
            }
          } else {
            int ka, kb;
            krange(2 * bi + 1, 2 * bi, &ka, &kb);  // klim 3: the tile's lower 64 rows decide, klim 4: its upper 64 rows; the extra range of the other half meets zeros
            tiles.push_back({DAG_GEMM_128x64, bi, bj, bi * 128, tj * 64, ka * 64, kb * 64});
          }
        }
        // flops, as Problem::op_gflop counts them
        int ka = op.k0, kb = op.k1;
        if (op.klim == 1) kb = std::min(kb, bj + 1);
        if (op.klim == 2) ka = std::max(ka, bj);
        if (op.klim == 3) kb = std::min(kb, bi + 1);
        if (op.klim == 4) ka = std::max(ka, bi);
        for (int k = ka; k < kb; ++k) {
          double w = 1.0;
          if ((op.tri_a && k == bi) || (op.tri_b && k == bj)) w = 0.5;
          if (op.lower && bi == bj) w = std::min(w, 0.5);
          plan_.gflop += w * 2.0 * 128.0 * 128.0 * 128.0 * 1e-9;
        }
      }
      std::stable_sort(tiles.begin() + row_begin, tiles.end(), [](const Tile& a, const Tile& b) {
        auto wt = [](int kind) { return kind == DAG_GEMM_128x128 ? 4 : (kind == DAG_GEMM_128x64 ? 2 : 1); };
        const long wa = (long)(a.kb - a.ka) * wt(a.kind), wb = (long)(b.kb - b.ka) * wt(b.kind);
        return wa > wb;
      });
    }
    return tiles;
  }
  DagTask make(const Op& op, const Tile& tl, double* cost_us) {
    DagTask t{};
    t.kind = (uint16_t)tl.kind;
    t.flags = op.flags;
    t.row0 = tl.row0; t.col0 = tl.col0;
    t.kbeg = tl.ka / bk_ * bk_;
    t.kend = (tl.kb + bk_ - 1) / bk_ * bk_;
    *cost_us = cost_.overhead + (t.kend - t.kbeg) / 128.0 * (tl.kind == DAG_GEMM_128x128 ? cost_.per128_big128 : (tl.kind == DAG_GEMM_128x64 ? cost_.per128_big : (tl.kind == DAG_GEMM_32x64 ? cost_.per128_chain : cost_.per128_small)));
    return t;
  }
  static int count_row(const std::vector<Tile>& tiles, int bi) {
    int c = 0;
    for (const Tile& t : tiles) c += t.bi == bi;
    return c;
  }

  int top_lo_ = 0, top_hi_ = 0, top_mid_ = -1;
  DagGate top_right_all_;  // the right half of the top node is final (its X22)
  DagGate top_left_all_;   // the left half of the top node is final (its X11)
  bool lauum_split_ = false;  // right-looking plan: the K^-1 tiles of the top-left quadrant in two parts (DAGF_CINIT: the same bits as one part)
  struct Sub {
    int lo = 0, hi = 0;
    std::vector<DagGate> rowfin;  // [j - lo]: row block j of X is final inside this subtree
    DagGate all;                  // the whole subtree is final
  };
  typedef std::vector<DagGate> RowGates;  // indexed by absolute block row; cnt < 0 where nothing has to be waited for

  Sub rec(int lo, int hi, const RowGates* gate) {
    Sub out;
    out.lo = lo; out.hi = hi;
    auto gate_of = [&](int i) { return (gate && i < (int)gate->size()) ? (*gate)[i] : DagGate(); };
    if (hi - lo == 1) {
      DagTask t{};
      t.kind = DAG_LEAF;
      t.row0 = lo;
      const int c = new_counter();
      push(t, {gate_of(lo)}, c, -1, cost_.leaf);
      plan_.n_leaf++;
      out.rowfin.assign(1, DagGate{c, 1});
      out.all = DagGate{c, 1};
      return out;
    }
    const int mid = dag_split_point(lo, hi);
    const bool small = std::max(mid - lo, hi - mid) <= small_h_;
    const Sub left = rec(lo, mid, gate);
    double cu = 0;
    // ---- T = A21 * X11^T -> W2[2,1]
    Op t{};
    t.flags = DAGF_BBUF | DAGF_CBUF;  // A = W1, B = W2, C = W2
    t.r0 = mid; t.r1 = hi; t.c0 = lo; t.c1 = mid; t.k0 = lo; t.k1 = mid; t.klim = 1; t.tri_b = true; t.crit = 1;
    const std::vector<Tile> tt = tiles_of(t, small);
    RowGates trow(hi);
    for (int i = mid; i < hi; ++i) trow[i] = DagGate{new_counter(), count_row(tt, i)};
    for (const Tile& tl : tt) {
      const DagTask tk = make(t, tl, &cu);
      push(tk, {fine_ ? left.rowfin[tl.bj - lo] : left.all, gate_of(tl.bi)}, trow[tl.bi].cnt, -1, cu);
    }
    // ---- A22 -= T T^T (lower) -> W1: its rows gate the right subtree
    Op s{};
    s.flags = DAGF_ABUF | DAGF_BBUF | DAGF_NEG | DAGF_ACC;  // A = B = W2, C = W1
    s.r0 = mid; s.r1 = hi; s.c0 = mid; s.c1 = hi; s.lower = true; s.k0 = lo; s.k1 = mid; s.crit = 1;
    const std::vector<Tile> st = tiles_of(s, small);
    RowGates srow(hi);
    for (int i = mid; i < hi; ++i) srow[i] = DagGate{new_counter(), count_row(st, i)};
    const DagGate sall{new_counter(), (int)st.size()};
    for (const Tile& tl : st) {
      const DagTask tk = make(s, tl, &cu);
      push(tk, {trow[tl.bi], trow[tl.bj]}, srow[tl.bi].cnt, sall.cnt, cu);
    }
    // ---- U = T * X11 -> W1[2,1]
    Op u{};
    u.flags = DAGF_ABUF | DAGF_BBUF | DAGF_BKM;  // A = W2 (T), B = W2 (X11, contraction along rows), C = W1
    u.r0 = mid; u.r1 = hi; u.c0 = lo; u.c1 = mid; u.k0 = lo; u.k1 = mid; u.klim = 2; u.tri_b = true;
    const std::vector<Tile> ut = tiles_of(u, small);
    const DagGate uall{new_counter(), (int)ut.size()};
    for (const Tile& tl : ut) {
      const DagTask tk = make(u, tl, &cu);
      push(tk, {trow[tl.bi]}, uall.cnt, -1, cu);
    }
    // ---- right subtree, gated row by row by this node's SYRK (coarse mode: by all of it)
    RowGates rgate(hi);
    for (int i = mid; i < hi; ++i) rgate[i] = fine_ ? srow[i] : sall;
    const Sub right = rec(mid, hi, &rgate);
    if (lo == top_lo_ && hi == top_hi_) { top_mid_ = mid; top_right_all_ = right.all; }
    // ---- X21 = -X22 * U -> W2[2,1]
    Op x{};
    x.flags = DAGF_ABUF | DAGF_BKM | DAGF_CBUF | DAGF_NEG;  // A = W2 (X22), B = W1 (U, contraction along rows), C = W2
    x.r0 = mid; x.r1 = hi; x.c0 = lo; x.c1 = mid; x.k0 = mid; x.k1 = hi; x.klim = 3; x.tri_a = true; x.crit = 2;
    const std::vector<Tile> xt = tiles_of(x, small);
    RowGates xrow(hi);
    for (int i = mid; i < hi; ++i) xrow[i] = DagGate{new_counter(), count_row(xt, i)};
    const DagGate xall{new_counter(), (int)xt.size()};
    for (const Tile& tl : xt) {
      const DagTask tk = make(x, tl, &cu);
      push(tk, {fine_ ? right.rowfin[tl.bi - mid] : right.all, uall, sall}, xrow[tl.bi].cnt, xall.cnt, cu);
    }
    out.rowfin.resize(hi - lo);
    for (int j = lo; j < mid; ++j) out.rowfin[j - lo] = left.rowfin[j - lo];
    for (int j = mid; j < hi; ++j) out.rowfin[j - lo] = xrow[j];
    out.all = xall;
    return out;
  }

  // K^-1 = X^T X (lower): tile (i, j) = sum over k >= i of X(k, i)^T X(k, j); X = L^-1 lower triangular in W2, zeros above
  // the diagonal in memory.  A tile needs every row of X from its own block row down: the tiles of the bottom-right quarter
  // wait for the right half of the top node only (they start while the top node's X21 is still being formed), all others
  // for the whole inverse factor.  Same per-element accumulation order as the LAUUM launch of gemm_kernel (k ascending from
  // the tile's first row; rows below it start on zeros).
  void build_lauum(int nb, const Sub& root) {
    Op l{};
    l.flags = DAGF_ABUF | DAGF_BBUF | DAGF_AKM | DAGF_BKM | DAGF_CKINV;  // A = B = W2 (contraction along rows), C = K^-1
    l.lower = true; l.klim = 4; l.tri_a = true; l.tri_b = true;
    const double g0 = plan_.gflop;
    double cu = 0;
    const bool split = lauum_split_ && top_mid_ > 0 && bk_ == 16;  // f64 only: an f32 partial sum would be rounded on its way through memory
    for (int bi = 0; bi < nb; ++bi)
      for (int bj = 0; bj <= bi; ++bj) {
        Op one = l;
        one.r0 = bi; one.r1 = bi + 1; one.c0 = bj; one.c1 = bj + 1;
        if (split && bi < top_mid_) {
          // top-left quadrant: the rows of X in the left half are final long before the top node's X21 is -- their part of
          // the sum first (waits for the left half only), the X21 part on top of it (beta = 1).  On the evaluation's tail
          // the deepest K^-1 tile is then half as deep (n/2 instead of n).
          Op first = one;
          first.k0 = 0; first.k1 = top_mid_;
          const std::vector<Tile> t1 = tiles_of(first, false);
          const int c1 = new_counter();
          for (const Tile& tl : t1) {
            const DagTask tk = make(first, tl, &cu);
            push(tk, {top_left_all_}, c1, -1, cu);
            plan_.n_lauum++;
          }
          Op second = one;
          second.flags |= DAGF_ACC | DAGF_CINIT;  // continues the first part's accumulation: the same bits as one undivided tile
          second.klim = 0; second.tri_a = second.tri_b = false;
          second.k0 = top_mid_; second.k1 = nb;
          const std::vector<Tile> t2 = tiles_of(second, false);
          for (const Tile& tl : t2) {
            const DagTask tk = make(second, tl, &cu);
            push(tk, {root.all, DagGate{c1, (int)t1.size()}}, -1, -1, cu);
            plan_.n_lauum++;
          }
        } else {
          one.k0 = 0; one.k1 = nb;
          const std::vector<Tile> lt = tiles_of(one, false);
          const bool quarter = top_mid_ >= 0 && bj >= top_mid_;
          for (const Tile& tl : lt) {
            const DagTask tk = make(one, tl, &cu);
            push(tk, {quarter ? top_right_all_ : root.all}, -1, -1, cu);
            plan_.n_lauum++;
          }
        }
      }
    plan_.gflop_lauum = plan_.gflop - g0;
  }

  // ---------------------------------------------------------------------------------------------------------------
  // Right-looking plan.  The recursion above computes T = A21 X11^T with the explicit inverse of the whole left half, so the
  // first block row of every T -- depth (half width) x 128, and only ready once the left half's inverse is complete -- sits
  // on the chain between two diagonal blocks: 1.34 of the 2.39 ms of one evaluation at n = 4096.  Here the factor itself is
  // formed tile by tile, classically:
  //   LEAF(k)          X_kk = L_kk^-1 from the updated diagonal tile (W1 -> W2 diagonal block, as before)
  //   TRSM(i,k)        L(i,k) = A(i,k) X_kk^T                      W1, W2 -> W3           (depth 128)
  //   UPD(i,j,[k0,k1)) A(i,j) -= L(i,k0..k1) L(j,k0..k1)^T          W3 -> W1, i >= j > k   (one column for the tiles needed
  //                                                                                       soon, ranges of 4 / 8 / 16 columns far from the chain)
  // so the chain between two diagonal blocks is TRSM(k+1,k) and UPD(k+1,k+1,k): two 128-deep tiles.  The inverse of the
  // factor follows the factor by the same divide and conquer as before, off the chain:
  //   U(N) = L21 X11 (W3, W2 -> W1[2,1]),  X21(N) = -X22 U (W2, W1 -> W2[2,1])
  // and the K^-1 tiles (build_lauum) follow the inverse.  X = L^-1 ends in W2 exactly as in the recursion (all its consumers
  // are unchanged); L stays intact in W3.  Arithmetic: mathematically the same factor, but a different order of operations
  // than the launch path's recursion -- parity with it is 1e-12-close, not bitwise.
  // Dependencies (every wait for a full count):
  //   leafc[k] | lc[i][k]: tile L(i,k) written (and, through the chain along the row, every L(i,k') with k' < k)
  //   lrow[N][i]: row i of L21(N) written
  //   upd(i,j,s): the s-th update of tile (i,j) applied (a chain per tile: updates run in ascending k)
  //   Uall[N], Xrow[N][i], Xall[N] as in the recursion
  struct RlNode { int lo, mid, hi; std::vector<int> lrow; };  // lrow[i - mid]: counter id
  // How the updates of the tiles of block column j are grouped: a list of column ranges [k0, k1) in ascending order.  Far
  // from the diagonal chain the ranges are wide (deep, efficient tiles: 16, 8 or 4 columns, aligned), the last columns
  // before j are applied one by one as soon as each is ready.  A range of G columns is ready when its last column is, and
  // takes G x 8 us: it has to end early enough not to delay column j (NEAR columns for G = 4, +1 for 8, +3 for 16).
  static std::vector<std::pair<int, int>> rl_groups(int j, int maxgroup, int near) {
    std::vector<std::pair<int, int>> out;
    int a = 0;
    while (a < j) {
      int g = 1;
      for (int cand : {32, 16, 8, 4, 2}) {
        const int extra = cand == 32 ? 6 : (cand == 16 ? 3 : (cand == 8 ? 1 : 0));
        if (cand <= maxgroup && a % cand == 0 && a + cand <= j - near - extra) { g = cand; break; }
      }
      out.push_back({a, a + g});
      a += g;
    }
    return out;
  }
  void build_rl(int nb, bool lauum) {
    // widest grouped update; tiles within NEAR block columns (and NEAR + 1 block rows) of the current column get 64x64 tasks
    const int GROUP = rl_group_, NEAR = rl_near_;
    // node tree of the inverse's recursion (same split points as rec())
    std::vector<RlNode> nodes;
    std::vector<std::vector<int>> node_of(nb, std::vector<int>(nb, -1));  // node_of[i][k]: node with k in its left, i in its right half
    struct Frame { int lo, hi; };
    std::vector<Frame> stack{{0, nb}};
    while (!stack.empty()) {
      const Frame f = stack.back();
      stack.pop_back();
      if (f.hi - f.lo < 2) continue;
      const int mid = dag_split_point(f.lo, f.hi);
      RlNode nd{f.lo, mid, f.hi, {}};
      for (int i = mid; i < f.hi; ++i) nd.lrow.push_back(new_counter());
      const int id = (int)nodes.size();
      nodes.push_back(nd);
      for (int i = mid; i < f.hi; ++i)
        for (int k = f.lo; k < mid; ++k) node_of[i][k] = id;
      stack.push_back({f.lo, mid});
      stack.push_back({mid, f.hi});
    }
    std::vector<DagGate> leafc(nb);
    std::vector<std::vector<DagGate>> lc(nb, std::vector<DagGate>(nb)), last_upd(nb, std::vector<DagGate>(nb));
    std::vector<std::vector<std::pair<int, int>>> groups(nb);
    for (int j = 0; j < nb; ++j) groups[j] = rl_groups(j, GROUP, NEAR);
    double cu = 0;
    auto single_tile = [&](Op op, int i, int j) { op.r0 = i; op.r1 = i + 1; op.c0 = j; op.c1 = j + 1; return op; };
    for (int k = 0; k < nb; ++k) {
      // ---- diagonal block k
      {
        DagTask t{};
        t.kind = DAG_LEAF;
        t.row0 = k;
        const int c = new_counter();
        push(t, {last_upd[k][k]}, c, -1, cost_.leaf);
        plan_.n_leaf++;
        leafc[k] = DagGate{c, 1};
      }
      // ---- L(i,k) = A(i,k) X_kk^T -> W3.  Each also waits for its left neighbour L(i,k-1) (long done in practice): "the
      // last tile of a column range is written" then implies the whole range, so a grouped update needs two waits, not 2 G.
      for (int i = k + 1; i < nb; ++i) {
        Op t{};
        t.flags = DAGF_BBUF | DAGF_C3;  // A = W1, B = W2 (X_kk), C = W3
        t.k0 = k; t.k1 = k + 1; t.klim = 1; t.tri_b = true;
        const std::vector<Tile> tt = tiles_of(single_tile(t, i, k), i - k <= NEAR, rl_chain32_ && i == k + 1);
        const int c = new_counter();
        lc[i][k] = DagGate{c, (int)tt.size()};
        const int nd = node_of[i][k];
        for (const Tile& tl : tt) {
          const DagTask tk = make(t, tl, &cu);
          push(tk, {leafc[k], last_upd[i][k], k > 0 ? lc[i][k - 1] : DagGate()}, c, nodes[nd].lrow[i - nodes[nd].mid], cu);
        }
      }
      // ---- updates whose column range ends with column k
      for (int j = k + 1; j < nb; ++j) {
        int k0 = -1;
        for (const auto& gr : groups[j])
          if (gr.second == k + 1) k0 = gr.first;
        if (k0 < 0) continue;  // column k is inside a wider range of this block column: applied with the range's last column
        const bool single = k0 == k;
        for (int i = j; i < nb; ++i) {
          Op u{};
          u.flags = DAGF_A3 | DAGF_B3 | DAGF_NEG | DAGF_ACC;  // A = B = W3 (L), C = W1
          u.lower = (i == j);
          u.k0 = k0; u.k1 = k + 1;
          const std::vector<Tile> ut = tiles_of(single_tile(u, i, j), single && j - k <= NEAR && i - k <= NEAR + 1,
                                                rl_chain32_ && single && i == k + 1 && j == k + 1);
          const int c = new_counter();
          for (const Tile& tl : ut) {
            const DagTask tk = make(u, tl, &cu);
            push(tk, {lc[i][k], lc[j][k], last_upd[i][j]}, c, -1, cu);
          }
          last_upd[i][j] = DagGate{c, (int)ut.size()};
        }
      }
    }
    if (rl_prog_) {
      rl_progressive(nb, lauum, leafc, lc);
      return;
    }
    // ---- the inverse of the factor, by the recursion's divide and conquer
    const Sub root = rl_inverse(0, nb, nodes, leafc);
    if (lauum) build_lauum(nb, root);
  }

  // Round 4.  With the divide-and-conquer inverse nothing of the bottom block rows of X = L^-1 exists before the LAST diagonal
  // block is done (X21 = -X22 L21 X11 needs all of X22), and every tile of K^-1 = X^T X needs those rows: at n = 4096 a third of
  // an evaluation's flops (all of K^-1, the top node's X21) could only start when the chain of diagonal blocks had ended -- 460 us
  // of a 1.88 ms launch with every CU busy, after 1.4 ms in which the chain kept 30-130 of the 256 CUs busy.  Here both follow
  // the chain row by row (block forward substitution), so that behind the last diagonal block only its own row is left:
  //   U(i,j)   = sum_{k=j}^{i-1} L(i,k) X(k,j)      accumulated in W1(i,j) (dead since L(i,j) was formed) over ranges of k: a range
  //                                                  [a,b) is applied once row b-1 of X is final -- wide ranges far from row i
  //                                                  (deep, efficient tiles), single rows just before it (rl_groups' rule)
  //   X(i,j)   = -X_ii U(i,j)                        as soon as diagonal block i is done; row i of X is then final
  //   Kinv(i,j) += sum_{k in [a,b)} X(k,i)^T X(k,j)  for every range [a,b) of final rows, i, j < b
  // f64: a continued sum starts its accumulators from the stored values (DAGF_CINIT), so every element is ONE k-ascending chain of
  // MFMA accumulations whatever the ranges are; f32: the partial sums go through memory (rounded to f32 once per range).
  // Same flops as the divide and conquer (sum_{i>j} (i - j) block products for the inverse).  Dependencies: leafc / lc as above;
  //   rowfin[k]: row k of X final (implies every row above it: U(k,.) needed them);  uprev[i]: the last update of U(i,.) so far;
  //   kprev: the previous range's K^-1 tiles.
  void rl_progressive(int nb, bool lauum, const std::vector<DagGate>& leafc, const std::vector<std::vector<DagGate>>& lc) {
    const int GROUP = rl_group_, NEAR = rl_near_;
    const uint16_t cont = bk_ == 16 ? (uint16_t)(DAGF_ACC | DAGF_CINIT) : (uint16_t)DAGF_ACC;
    std::vector<DagGate> rowfin(nb), uprev(nb);
    std::vector<std::vector<std::pair<int, int>>> ugroups(nb);
    const int UNEAR = prog_unear_ >= 0 ? prog_unear_ : NEAR, KNEAR = prog_knear_ >= 0 ? prog_knear_ : NEAR;
    for (int i = 1; i < nb; ++i) ugroups[i] = rl_groups(i, GROUP, UNEAR);
    const std::vector<std::pair<int, int>> kgroups = rl_groups(nb, GROUP, KNEAR);
    DagGate kprev;
    const double g0 = plan_.gflop;
    double g_lauum = 0, cu = 0;
    for (int k = 0; k < nb; ++k) {
      // ---- row k of X
      if (k == 0) {
        rowfin[0] = leafc[0];
      } else {
        Op x{};
        x.flags = DAGF_ABUF | DAGF_BKM | DAGF_CBUF | DAGF_NEG;  // A = W2 (X_kk), B = W1 (U, contraction along rows), C = W2
        x.r0 = k; x.r1 = k + 1; x.c0 = 0; x.c1 = k; x.k0 = k; x.k1 = k + 1; x.klim = 3; x.tri_a = true;
        const std::vector<Tile> xt = tiles_of(x, prog_small_);
        const int c = new_counter();
        for (const Tile& tl : xt) {
          const DagTask tk = make(x, tl, &cu);
          push(tk, {leafc[k], uprev[k]}, c, -1, cu);
        }
        rowfin[k] = DagGate{c, (int)xt.size()};
      }
      // ---- U(i, .) += L(i, a..k) X(a..k, .) for every row i whose list of ranges has one that ends with row k
      for (int i = k + 1; i < nb; ++i) {
        int a = -1;
        for (const auto& gr : ugroups[i])
          if (gr.second == k + 1) a = gr.first;
        if (a < 0) continue;
        const bool near = prog_small_ && (k + 1 - a) == 1 && i - k <= NEAR + 1;  // the last rows before row i: latency counts
        const int c = new_counter();
        int cnt = 0;
        const std::vector<DagGate> waits = {lc[i][k], rowfin[k], uprev[i]};
        if (a > 0) {  // columns left of the range: earlier ranges have started their sums
          Op u{};
          u.flags = (uint16_t)(DAGF_A3 | DAGF_BBUF | DAGF_BKM | cont);  // A = W3 (L), B = W2 (X, contraction along rows), C = W1
          u.r0 = i; u.r1 = i + 1; u.c0 = 0; u.c1 = a; u.k0 = a; u.k1 = k + 1;
          for (const Tile& tl : tiles_of(u, near)) {
            const DagTask tk = make(u, tl, &cu);
            push(tk, waits, c, -1, cu);
            ++cnt;
          }
        }
        {  // the range's own columns: first contribution (W1 holds the dead A(i,j) there), X(k,j) = 0 above the diagonal
          Op u{};
          u.flags = DAGF_A3 | DAGF_BBUF | DAGF_BKM;
          u.r0 = i; u.r1 = i + 1; u.c0 = a; u.c1 = k + 1; u.k0 = a; u.k1 = k + 1; u.klim = 2; u.tri_b = true;
          for (const Tile& tl : tiles_of(u, near)) {
            const DagTask tk = make(u, tl, &cu);
            push(tk, waits, c, -1, cu);
            ++cnt;
          }
        }
        uprev[i] = DagGate{c, cnt};
      }
      // ---- K^-1 += X(a..k, .)^T X(a..k, .) for the range of rows that ends with row k
      if (lauum) {
        int a = -1;
        for (const auto& gr : kgroups)
          if (gr.second == k + 1) a = gr.first;
        if (a >= 0) {
          const double gl0 = plan_.gflop;
          const int c = new_counter();
          int cnt = 0;
          const std::vector<DagGate> waits = {rowfin[k], kprev};
          if (a > 0) {
            Op l{};
            l.flags = (uint16_t)(DAGF_ABUF | DAGF_BBUF | DAGF_AKM | DAGF_BKM | DAGF_CKINV | cont);
            l.r0 = 0; l.r1 = a; l.c0 = 0; l.c1 = a; l.lower = true; l.k0 = a; l.k1 = k + 1;
            for (const Tile& tl : tiles_of(l, false)) {
              const DagTask tk = make(l, tl, &cu);
              push(tk, waits, c, -1, cu);
              ++cnt;
              plan_.n_lauum++;
            }
          }
          {
            Op l{};
            l.flags = DAGF_ABUF | DAGF_BBUF | DAGF_AKM | DAGF_BKM | DAGF_CKINV;
            l.r0 = a; l.r1 = k + 1; l.c0 = 0; l.c1 = k + 1; l.lower = true; l.k0 = a; l.k1 = k + 1; l.klim = 4; l.tri_a = true; l.tri_b = true;
            for (const Tile& tl : tiles_of(l, false)) {
              const DagTask tk = make(l, tl, &cu);
              push(tk, waits, c, -1, cu);
              ++cnt;
              plan_.n_lauum++;
            }
          }
          kprev = DagGate{c, cnt};
          g_lauum += plan_.gflop - gl0;
        }
      }
    }
    (void)g0;
    plan_.gflop_lauum = g_lauum;
  }
  Sub rl_inverse(int lo, int hi, const std::vector<RlNode>& nodes, const std::vector<DagGate>& leafc) {
    Sub out;
    out.lo = lo; out.hi = hi;
    if (hi - lo == 1) {
      out.rowfin.assign(1, leafc[lo]);
      out.all = leafc[lo];
      return out;
    }
    const int mid = dag_split_point(lo, hi);
    const RlNode* nd = nullptr;
    for (const RlNode& cand : nodes)
      if (cand.lo == lo && cand.hi == hi) nd = &cand;
    const bool small = std::max(mid - lo, hi - mid) <= small_h_;
    const Sub left = rl_inverse(lo, mid, nodes, leafc);
    const Sub right = rl_inverse(mid, hi, nodes, leafc);
    if (lo == top_lo_ && hi == top_hi_) { top_mid_ = mid; top_right_all_ = right.all; top_left_all_ = left.all; }
    double cu = 0;
    // U = L21 X11 -> W1[2,1]   (A = W3, B = W2 with the contraction along rows)
    Op u{};
    u.flags = DAGF_A3 | DAGF_BBUF | DAGF_BKM;
    u.r0 = mid; u.r1 = hi; u.c0 = lo; u.c1 = mid; u.k0 = lo; u.k1 = mid; u.klim = 2; u.tri_b = true;
    const std::vector<Tile> ut = tiles_of(u, small);
    const DagGate uall{new_counter(), (int)ut.size()};
    for (const Tile& tl : ut) {
      const DagTask tk = make(u, tl, &cu);
      DagGate lrow{nd->lrow[tl.bi - mid], plan_.totals[nd->lrow[tl.bi - mid]]};
      push(tk, {lrow, left.all}, uall.cnt, -1, cu);
    }
    // X21 = -X22 U -> W2[2,1]
    Op x{};
    x.flags = DAGF_ABUF | DAGF_BKM | DAGF_CBUF | DAGF_NEG;
    x.r0 = mid; x.r1 = hi; x.c0 = lo; x.c1 = mid; x.k0 = mid; x.k1 = hi; x.klim = 3; x.tri_a = true; x.crit = 2;
    const std::vector<Tile> xt = tiles_of(x, small);
    RowGates xrow(hi);
    for (int i = mid; i < hi; ++i) xrow[i] = DagGate{new_counter(), count_row(xt, i)};
    const DagGate xall{new_counter(), (int)xt.size()};
    for (const Tile& tl : xt) {
      const DagTask tk = make(x, tl, &cu);
      push(tk, {fine_ ? right.rowfin[tl.bi - mid] : right.all, uall}, xrow[tl.bi].cnt, xall.cnt, cu);
    }
    out.rowfin.resize(hi - lo);
    for (int j = lo; j < mid; ++j) out.rowfin[j - lo] = left.rowfin[j - lo];
    for (int j = mid; j < hi; ++j) out.rowfin[j - lo] = xrow[j];
    out.all = xall;
    return out;
  }

  // Reorder the queue: simulate a list schedule with nwg_ workers, priority = longest remaining path.
  void order() {
    const int nt = (int)plan_.tasks.size(), nc = (int)plan_.totals.size();
    std::vector<std::vector<int>> waiters(nc);
    for (int i = 0; i < nt; ++i)
      for (int w = 0; w < plan_.tasks[i].nwait; ++w) waiters[plan_.tasks[i].wcnt[w]].push_back(i);
    // bottom levels: the emission order is topological, so one reverse sweep does it
    std::vector<double> bl(nt, 0.0), blc(nc, 0.0);
    for (int i = nt - 1; i >= 0; --i) {
      const DagTask& t = plan_.tasks[i];
      double tail = 0;
      for (int q = 0; q < DAG_MAXSIG; ++q)
        if (t.sig[q] != DAG_NOSIG) tail = std::max(tail, blc[t.sig[q]]);
      bl[i] = t.cost * 0.1 + tail;
      for (int w = 0; w < t.nwait; ++w) blc[t.wcnt[w]] = std::max(blc[t.wcnt[w]], bl[i]);
    }
    plan_.crit_us = *std::max_element(bl.begin(), bl.end());
    if (getenv("HBEGP_DAG_DUMP")) {  // the critical path, task by task (diagnostics)
      int cur = (int)(std::max_element(bl.begin(), bl.end()) - bl.begin());
      double acc_leaf = 0, acc_small = 0, acc_big = 0;
      for (;;) {
        const DagTask& t = plan_.tasks[cur];
        fprintf(stderr, "crit: kind=%d row=%d col=%d depth=%d cost=%.1f remaining=%.1f flags=%x\n", t.kind, t.row0, t.col0, t.kend - t.kbeg,
                t.cost * 0.1, bl[cur], t.flags);
        (t.kind == DAG_LEAF ? acc_leaf : (t.kind == DAG_GEMM_64x64 ? acc_small : acc_big)) += t.cost * 0.1;
        int nxt = -1;
        for (int q = 0; q < DAG_MAXSIG; ++q)
          if (t.sig[q] != DAG_NOSIG)
            for (int wtr : waiters[t.sig[q]])
              if (nxt < 0 || bl[wtr] > bl[nxt]) nxt = wtr;
        if (nxt < 0) break;
        cur = nxt;
      }
      fprintf(stderr, "crit totals: leaf %.0f us, 64x64 tiles %.0f us, 128x64 tiles %.0f us\n", acc_leaf, acc_small, acc_big);
    }
    std::vector<int> count(nc, 0), missing(nt, 0);
    typedef std::pair<double, int> Pri;  // (bottom level, -index): highest first, earlier emission breaks ties
    std::priority_queue<Pri> ready;
    for (int i = 0; i < nt; ++i) {
      missing[i] = plan_.tasks[i].nwait;
      if (missing[i] == 0) ready.push({bl[i], -i});
    }
    typedef std::pair<double, int> Ev;  // (finish time, task)
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> running;
    std::vector<DagTask> out;
    out.reserve(nt);
    std::vector<int> out_idx;      // original index of the task at each queue position
    std::vector<double> start(nt, 0.0);
    out_idx.reserve(nt);
    int idle = nwg_;
    double now = 0;
    while ((int)out.size() < nt) {
      while (idle > 0 && !ready.empty()) {
        const int i = -ready.top().second;
        ready.pop();
        out.push_back(plan_.tasks[i]);
        out_idx.push_back(i);
        start[i] = now;
        running.push({now + plan_.tasks[i].cost * 0.1, i});
        --idle;
      }
      if (running.empty()) break;  // cannot happen for a sound graph
      const Ev e = running.top();
      running.pop();
      now = e.first;
      ++idle;
      const DagTask& t = plan_.tasks[e.second];
      for (int q = 0; q < DAG_MAXSIG; ++q) {
        if (t.sig[q] == DAG_NOSIG) continue;
        const int c = t.sig[q];
        if (++count[c] == plan_.totals[c])
          for (int wtr : waiters[c])
            if (--missing[wtr] == 0) ready.push({bl[wtr], -wtr});
      }
    }
    while (!running.empty()) { now = running.top().first; running.pop(); }
    if ((int)out.size() == nt) {
      plan_.tasks.swap(out);
      plan_.sim_us = now;
    }
  }

 public:
  void set_big128(bool on, bool with_acc = false) { big128_ = on; big128_acc_ = on && with_acc; }
};

// ---------------------------------------------------------------------------------------------------------------
// Checks of a plan, all on the host:
//  1. the queue order is a topological order: executing the tasks one by one in queue order never waits (this is what
//     makes the kernel deadlock-free for any number of resident workgroups);
//  2. every wait is for the full count of its counter;
//  3. no data race at 64x64-tile granularity: for every tile a task reads, the last writer happened before it; for
//     every tile it writes, the last writer and every reader since happened before it ("happened before" = reachable
//     through full-count waits).
// Returns an empty string when the plan is sound.
inline std::string dag_plan_validate(const DagPlan& plan, int nblocks_total) {
  const int nc = (int)plan.totals.size(), nt = (int)plan.tasks.size();
  const int words = (nc + 63) / 64;
  char buf[256];
  std::vector<int> count(nc, 0);
  // known[task] = set of counters whose full count happened before the task STARTS; done_known[c] = union over the
  // tasks bumping c of (known[task] + nothing): what a waiter on c learns
  std::vector<uint64_t> counter_known((size_t)nc * words, 0), cur(words);
  const int nt64 = nblocks_total * 4;  // table dimension: 64x64 tiles need 2 per block, the 32-row cells of w need 4
  struct Cell { int writer = -1; std::vector<int> readers; };
  // buffers 0/1: W1/W2 in 64x64 tiles; 2: w (32-row cells); 3: chunk partials (chunk, 128-column cell); 4: diag(L) (128-row
  // cells); 5: per-block sums.  All share one table of nt64 x nt64 cells (the 1-D ones use column 0).
  constexpr int NBUF = 8;  // 6: K^-1 (64x64 tiles), 7: W3 (the factor L of the right-looking plan)
  std::vector<Cell> cells[NBUF];
  for (int b = 0; b < NBUF; ++b) cells[b].resize((size_t)nt64 * nt64);
  std::vector<std::vector<uint64_t>> known(nt);
  auto hb = [&](int a, int b) -> bool {  // task a happened before task b starts
    for (int q = 0; q < DAG_MAXSIG; ++q) {
      const int c = plan.tasks[a].sig[q];
      if (c != DAG_NOSIG && ((known[b][c / 64] >> (c % 64)) & 1u) != 0) return true;
    }
    return false;
  };
  for (int i = 0; i < nt; ++i) {
    const DagTask& t = plan.tasks[i];
    // A tile that STARTS its accumulation from the stored values (DAGF_CINIT) is only defined the way build_lauum emits it:
    // beta = 1, alpha = +1 and a non-empty contraction range (an empty one would store zeros over the first part's sums on the
    // 128x64 path, dag_gemm_tile); f32 problems never split (an f32 partial sum would be rounded on its way through memory):
    // build_lauum splits at the f64 stage depth only.
    if (t.kind == DAG_GEMM_128x128 && (t.flags & DAGF_CINIT)) {
      snprintf(buf, sizeof buf, "task %d: the 128x128 tile cannot continue a sum (flags %x)", i, t.flags);
      return buf;
    }
    if (t.kind == DAG_GEMM_32x64 && (t.flags & (DAGF_AKM | DAGF_BKM | DAGF_CINIT | DAGF_CKINV))) {
      snprintf(buf, sizeof buf, "task %d: the 32x64 one-shot tile reads both operands [outer][k] and writes W1/W2/W3 (flags %x)", i, t.flags);
      return buf;
    }
    if ((t.kind == DAG_GEMM_128x64 || t.kind == DAG_GEMM_64x64) && (t.flags & DAGF_CINIT)) {
      if (!(t.flags & DAGF_ACC) || (t.flags & DAGF_NEG) || t.kend <= t.kbeg) {
        snprintf(buf, sizeof buf, "task %d: DAGF_CINIT needs DAGF_ACC, no DAGF_NEG and a non-empty contraction range (flags %x, k [%d, %d))", i,
                 t.flags, t.kbeg, t.kend);
        return buf;
      }
    }
    std::fill(cur.begin(), cur.end(), 0);
    for (int w = 0; w < t.nwait; ++w) {
      const int c = t.wcnt[w];
      if (c >= nc) return "wait on an unknown counter";
      if (t.wval[w] != plan.totals[c]) {
        snprintf(buf, sizeof buf, "task %d waits for %d of counter %d (total %d): partial waits are not allowed", i, t.wval[w], c, plan.totals[c]);
        return buf;
      }
      if (count[c] < t.wval[w]) {
        snprintf(buf, sizeof buf, "task %d waits on counter %d = %d but only %d earlier tasks bump it: queue order is not topological", i,
                 c, t.wval[w], count[c]);
        return buf;
      }
      cur[c / 64] |= 1ull << (c % 64);
      for (int q = 0; q < words; ++q) cur[q] |= counter_known[(size_t)c * words + q];
    }
    known[i] = cur;
    // footprint
    struct Acc { int buf, r, c; bool write; };
    std::vector<Acc> accs;
    auto rect = [&](int bufi, int r0, int r1, int c0, int c1, bool write) {  // element ranges; cells of 32 rows x 64 columns
      for (int r = r0 / 32; r < (r1 + 31) / 32; ++r)
        for (int c = c0 / 64; c < (c1 + 63) / 64; ++c) accs.push_back({bufi, r, c, write});
    };
    auto cell = [&](int bufi, int r, int c, bool write) { accs.push_back({bufi, r, c, write}); };
    if (t.kind == DAG_LEAF) {
      const int b = t.row0 * 128;
      rect(0, b, b + 128, b, b + 128, false);
      rect(1, b, b + 128, b, b + 128, true);
      cell(4, t.row0, 0, true);
    } else {
      const int ta = (t.kind == DAG_GEMM_128x64 || t.kind == DAG_GEMM_128x128) ? 128 : (t.kind == DAG_GEMM_32x64 ? 32 : 64), tb = t.kind == DAG_GEMM_128x128 ? 128 : 64;
      const int ab = (t.flags & DAGF_A3) ? 7 : ((t.flags & DAGF_ABUF) ? 1 : 0), bb = (t.flags & DAGF_B3) ? 7 : ((t.flags & DAGF_BBUF) ? 1 : 0);
      const int cb = (t.flags & DAGF_CKINV) ? 6 : ((t.flags & DAGF_C3) ? 7 : ((t.flags & DAGF_CBUF) ? 1 : 0));
      if (t.kend <= t.kbeg) return "empty contraction range";
      if (t.flags & DAGF_AKM) rect(ab, t.kbeg, t.kend, t.row0, t.row0 + ta, false);
      else rect(ab, t.row0, t.row0 + ta, t.kbeg, t.kend, false);
      if (t.flags & DAGF_BKM) rect(bb, t.kbeg, t.kend, t.col0, t.col0 + tb, false);
      else rect(bb, t.col0, t.col0 + tb, t.kbeg, t.kend, false);
      if (t.flags & DAGF_ACC) rect(cb, t.row0, t.row0 + ta, t.col0, t.col0 + tb, false);
      rect(cb, t.row0, t.row0 + ta, t.col0, t.col0 + tb, true);
    }
    for (const Acc& a : accs) {
      if (a.r >= nt64 || a.c >= nt64 || a.r < 0 || a.c < 0) return "tile outside the matrix";
      Cell& cell = cells[a.buf][(size_t)a.r * nt64 + a.c];
      if (a.buf >= 2 && a.buf != 6 && !a.write && cell.writer < 0) {
        snprintf(buf, sizeof buf, "task %d reads cell (%d,%d) of buffer %d that no earlier task has written", i, a.r, a.c, a.buf);
        return buf;
      }
      if (cell.writer >= 0 && cell.writer != i && !hb(cell.writer, i)) {
        snprintf(buf, sizeof buf, "task %d %s tile (%d,%d) of W%d written by task %d without waiting for it", i,
                 a.write ? "overwrites" : "reads", a.r, a.c, a.buf + 1, cell.writer);
        return buf;
      }
      if (a.write)
        for (int rd : cell.readers)
          if (rd != i && !hb(rd, i)) {
            snprintf(buf, sizeof buf, "task %d overwrites tile (%d,%d) of W%d that task %d reads, without waiting for it", i, a.r, a.c,
                     a.buf + 1, rd);
            return buf;
          }
    }
    for (const Acc& a : accs) {
      Cell& cell = cells[a.buf][(size_t)a.r * nt64 + a.c];
      if (a.write) { cell.writer = i; cell.readers.clear(); }
      else if (cell.readers.empty() || cell.readers.back() != i) cell.readers.push_back(i);
    }
    for (int sq = 0; sq < DAG_MAXSIG; ++sq) {
      if (t.sig[sq] == DAG_NOSIG) continue;
      const int c = t.sig[sq];
      if (c >= nc) return "bump of an unknown counter";
      count[c]++;
      for (int q = 0; q < words; ++q) counter_known[(size_t)c * words + q] |= cur[q];
    }
  }
  for (int c = 0; c < nc; ++c)
    if (count[c] != plan.totals[c]) return "counter total mismatch";
  return "";
}

}  // namespace hbegp
