// dag_plan.hpp — host side of the device-scheduled factorisation (dag_kernel.inc.hpp): builds the ordered task queue
// and its dependency counters for the Cholesky + inverse-factor recursion on a range of 128-blocks, and checks it.
// Plain C++ (no HIP): also compiled into the CPU-only tests through hbegp_debug_dag_plan().
//
// Recursion (same as Problem::chol_inv_rec / chol_inv_split, lml.rs:47 + the inverse of the factor lml.rs:62 needs):
//   node [lo, hi), mid:   left subtree;  T = A21 X11^T (W1,W2 -> W2[2,1]);  A22 -= T T^T (W2 -> W1, lower);
//                         U = T X11 (W2 -> W1[2,1]);  right subtree;  X21 = -X22 U (W2,W1 -> W2[2,1])
// Counters: one per product and one per diagonal block.  Every wait is for the FULL count of a counter, so "B waits on
// c" means "every task that bumps c happened before B" -- which is what dag_plan_validate() relies on.
//   T      waits: left subtree complete            (its X11, and through it every earlier update of A21)
//   SYRK,U wait : all T tiles                      (U overwrites A21, which every T tile of its row has read)
//   right subtree's diagonal blocks wait: all SYRK tiles of this node (their gate; inner nodes inherit it through
//                                         the chain leaf -> T -> ...)
//   X21    waits: right subtree complete, all U tiles   (it overwrites T: SYRK and U have read it)
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "engine.hpp"

namespace hbegp {

struct DagPlan {
  std::vector<DagTask> tasks;
  std::vector<int> totals;  // per counter: number of tasks that bump it
  double gflop = 0;         // algorithmic flops of the tile products (2*128^3 per pair of 128-blocks, half on triangles)
  int n_leaf = 0;
};

struct DagGate {
  int cnt = -1, val = 0;  // cnt < 0: nothing to wait for
};

class DagBuilder {
 public:
  // bk: contraction elements per pipeline stage (16 for f64, 32 for f32): ranges are whole stages
  // small_h: nodes whose halves are at most this many 128-blocks wide use 64x64 tiles (latency-bound products)
  DagBuilder(int bk, int small_h) : bk_(bk), small_h_(small_h) {}

  DagPlan build(int blo, int bhi) {
    plan_ = DagPlan();
    rec(blo, bhi, DagGate());
    return plan_;
  }

 private:
  int bk_, small_h_;
  DagPlan plan_;

  int new_counter() {
    plan_.totals.push_back(0);
    return (int)plan_.totals.size() - 1;
  }
  void push(DagTask t, const std::vector<DagGate>& waits, int sig) {
    t.nwait = 0;
    for (const DagGate& g : waits)
      if (g.cnt >= 0) {
        t.wcnt[t.nwait] = (uint16_t)g.cnt;
        t.wval[t.nwait] = g.val;
        ++t.nwait;
      }
    t.sig = sig < 0 ? DAG_NOSIG : (uint16_t)sig;
    if (sig >= 0) plan_.totals[sig]++;
    plan_.tasks.push_back(t);
  }

  struct Op {
    uint16_t flags;
    int r0, r1, c0, c1;  // output block rows / cols (128-blocks)
    bool lower;          // only blocks with row >= col (diagonal blocks: lower 64-tiles only)
    int k0, k1;          // contraction block range
    int klim;            // 0 none | 1: k <= tj | 2: k >= tj | 3: k <= ti   (64-element units, as gemm_kernel's 64-tile)
    bool tri_a, tri_b;   // operand is triangular (flop accounting only)
  };

  // emit the tiles of one product; returns the number of tasks.  order: deepest contraction first.
  int emit(const Op& op, bool small, const std::vector<DagGate>& waits, int sig) {
    struct Tile { int kind, row0, col0, ka, kb; };
    std::vector<Tile> tiles;
    for (int bi = op.r0; bi < op.r1; ++bi)
      for (int bj = op.c0; bj < op.c1; ++bj) {
        if (op.lower && bj > bi) continue;
        const bool diag = op.lower && bi == bj;
        for (int hj = 0; hj < 2; ++hj) {
          const int tj = 2 * bj + hj;  // 64-column unit
          auto krange = [&](int ti_last, int* ka, int* kb) {
            *ka = 2 * op.k0; *kb = 2 * op.k1;
            if (op.klim == 1) *kb = std::min(*kb, tj + 1);
            if (op.klim == 2) *ka = std::max(*ka, tj);
            if (op.klim == 3) *kb = std::min(*kb, ti_last + 1);
          };
          if (small || diag) {
            for (int hi = 0; hi < 2; ++hi) {
              const int ti = 2 * bi + hi;
              if (diag && ti < tj) continue;  // strictly upper 64-tile of a symmetric result
              int ka, kb;
              krange(ti, &ka, &kb);
              tiles.push_back({DAG_GEMM_64x64, ti * 64, tj * 64, ka * 64, kb * 64});
            }
          } else {
            int ka, kb;
            krange(2 * bi + 1, &ka, &kb);  // the tile's lower 64 rows decide; the extra range of the upper rows meets zeros
            tiles.push_back({DAG_GEMM_128x64, bi * 128, tj * 64, ka * 64, kb * 64});
          }
        }
        // flops, as Problem::op_gflop counts them
        {
          int ka = op.k0, kb = op.k1;
          if (op.klim == 1) kb = std::min(kb, bj + 1);
          if (op.klim == 2) ka = std::max(ka, bj);
          if (op.klim == 3) kb = std::min(kb, bi + 1);
          for (int k = ka; k < kb; ++k) {
            double w = 1.0;
            if ((op.tri_a && k == bi) || (op.tri_b && k == bj)) w = 0.5;
            if (op.lower && bi == bj) w = std::min(w, 0.5);
            plan_.gflop += w * 2.0 * 128.0 * 128.0 * 128.0 * 1e-9;
          }
        }
      }
    std::stable_sort(tiles.begin(), tiles.end(), [](const Tile& a, const Tile& b) {
      const long wa = (long)(a.kb - a.ka) * (a.kind == DAG_GEMM_128x64 ? 2 : 1), wb = (long)(b.kb - b.ka) * (b.kind == DAG_GEMM_128x64 ? 2 : 1);
      return wa > wb;
    });
    for (const Tile& tl : tiles) {
      DagTask t{};
      t.kind = (uint16_t)tl.kind;
      t.flags = op.flags;
      t.row0 = tl.row0; t.col0 = tl.col0;
      t.kbeg = tl.ka / bk_ * bk_;
      t.kend = (tl.kb + bk_ - 1) / bk_ * bk_;
      push(t, waits, sig);
    }
    return (int)tiles.size();
  }

  // returns the gate "X of [lo, hi) is final"
  DagGate rec(int lo, int hi, DagGate gate) {
    if (hi - lo == 1) {
      DagTask t{};
      t.kind = DAG_LEAF;
      t.row0 = lo;
      const int c = new_counter();
      push(t, {gate}, c);
      plan_.n_leaf++;
      return DagGate{c, 1};
    }
    const int mid = lo + (hi - lo) / 2;
    const bool small = std::max(mid - lo, hi - mid) <= small_h_;
    const DagGate left = rec(lo, mid, gate);
    // T = A21 * X11^T -> W2[2,1]
    const int cT = new_counter();
    Op t{};
    t.flags = DAGF_BBUF | DAGF_CBUF;  // A = W1, B = W2, C = W2
    t.r0 = mid; t.r1 = hi; t.c0 = lo; t.c1 = mid; t.k0 = lo; t.k1 = mid; t.klim = 1; t.tri_b = true;
    const int nT = emit(t, small, {left}, cT);
    const DagGate gT{cT, nT};
    // A22 -= T T^T (lower) -> W1: gates the right subtree, so it is queued before U
    const int cS = new_counter();
    Op s{};
    s.flags = DAGF_ABUF | DAGF_BBUF | DAGF_NEG | DAGF_ACC;  // A = B = W2, C = W1
    s.r0 = mid; s.r1 = hi; s.c0 = mid; s.c1 = hi; s.lower = true; s.k0 = lo; s.k1 = mid;
    const int nS = emit(s, small, {gT}, cS);
    // U = T * X11 -> W1[2,1]
    const int cU = new_counter();
    Op u{};
    u.flags = DAGF_ABUF | DAGF_BBUF | DAGF_BKM;  // A = W2 (T), B = W2 (X11, contraction along rows), C = W1
    u.r0 = mid; u.r1 = hi; u.c0 = lo; u.c1 = mid; u.k0 = lo; u.k1 = mid; u.klim = 2; u.tri_b = true;
    const int nU = emit(u, small, {gT}, cU);
    const DagGate right = rec(mid, hi, DagGate{cS, nS});
    // X21 = -X22 * U -> W2[2,1]
    const int cX = new_counter();
    Op x{};
    x.flags = DAGF_ABUF | DAGF_BKM | DAGF_CBUF | DAGF_NEG;  // A = W2 (X22), B = W1 (U, contraction along rows), C = W2
    x.r0 = mid; x.r1 = hi; x.c0 = lo; x.c1 = mid; x.k0 = mid; x.k1 = hi; x.klim = 3; x.tri_a = true;
    const int nX = emit(x, small, {right, DagGate{cU, nU}}, cX);
    return DagGate{cX, nX};
  }
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
  const int nt64 = nblocks_total * 2;
  struct Cell { int writer = -1; std::vector<int> readers; };
  std::vector<Cell> cells[2];
  cells[0].resize((size_t)nt64 * nt64);
  cells[1].resize((size_t)nt64 * nt64);
  std::vector<std::vector<uint64_t>> known(nt);
  auto hb = [&](int a, int b) -> bool {  // task a happened before task b starts
    const int c = plan.tasks[a].sig;
    if (c == DAG_NOSIG) return false;
    return ((known[b][c / 64] >> (c % 64)) & 1u) != 0;
  };
  for (int i = 0; i < nt; ++i) {
    const DagTask& t = plan.tasks[i];
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
    auto rect = [&](int bufi, int r0, int r1, int c0, int c1, bool write) {  // element ranges
      for (int r = r0 / 64; r < (r1 + 63) / 64; ++r)
        for (int c = c0 / 64; c < (c1 + 63) / 64; ++c) accs.push_back({bufi, r, c, write});
    };
    if (t.kind == DAG_LEAF) {
      const int b = t.row0 * 128;
      rect(0, b, b + 128, b, b + 128, false);
      rect(1, b, b + 128, b, b + 128, true);
    } else {
      const int ta = t.kind == DAG_GEMM_128x64 ? 128 : 64, tb = 64;
      const int ab = (t.flags & DAGF_ABUF) ? 1 : 0, bb = (t.flags & DAGF_BBUF) ? 1 : 0, cb = (t.flags & DAGF_CBUF) ? 1 : 0;
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
    if (t.sig != DAG_NOSIG) {
      const int c = t.sig;
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
