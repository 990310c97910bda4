// kernels.hip — hand-written gfx950 (CDNA4) kernels of the GP engine and their launchers.
//
//  gemm_kernel   : MFMA tile GEMM with triangular contraction ranges; carries Cholesky TRSM/SYRK, the
//                  triangular inverse (TRTRI as GEMMs), LAUUM (K^-1 = X^T X) and the predictive GEMM.
//  leaf_kernel   : 128x128 diagonal block: Cholesky + inverse of the factor, LDS resident, 16x16 diagonal
//                  sub-blocks factored in one wavefront with lane broadcasts, the rest on MFMA.
//  kmat_kernel   : K = c*Matern(X,X) + s2*I (lower tiles), matern_kernel.rs:37-81 + constant/product kernels + lml.rs:44.
//  gradtrace     : 1/2 tr((aa^T - K^-1) dK/dtheta_j) for all j in one pass over K^-1 (lml.rs:62-70, matern_kernel.rs:83-135)
//                  without the n x n x (d+1) tensor the reference materialises.
//  trmv/finalize : alpha = X^T (X y), y^T alpha, sum log L_ii (lml.rs:54-59).
//  kstar/pred_*  : predict.rs:7-52.
//
// Element type T is double or float; MFMA = v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32
// (A frag: lane l holds A[l&15][l>>4]; B frag: B[l>>4][l&15]; C/D: col = l&15, row = (l>>4)+4r for f64,
//  4*(l>>4)+r for f32).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "engine.hpp"
#include "fastmath.hpp"
#include "lbfgs_step.hpp"

namespace hbegp {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Cfg;
template <> struct Cfg<double> {
  static constexpr int VEC = 2;   // elements per 16-byte chunk
  static constexpr int BK = 16;   // contraction depth per LDS stage (128 bytes)
  using vec_t = d2;
  using acc_t = d4;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ void lds_store(double* dst, vec_t v) { *reinterpret_cast<vec_t*>(dst) = v; }
};
template <> struct Cfg<float> {
  static constexpr int VEC = 4;
  static constexpr int BK = 32;
  using vec_t = f4;
  using acc_t = f4;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) * 4 + r; }
  static __device__ __forceinline__ void lds_store(float* dst, vec_t v) {  // rows are only 8-byte aligned
    f2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    *reinterpret_cast<f2*>(dst) = lo;
    *reinterpret_cast<f2*>(dst + 2) = hi;
  }
};

__device__ __forceinline__ double readlane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// =================================================================================================================
// Tile GEMM
// =================================================================================================================
#ifndef HBEGP_F32_CHUNK
#define HBEGP_F32_CHUNK 32  /* round 4, 64 (make variant DEFS=-DHBEGP_F32_CHUNK=64): C5 f32 fits 8.63 -> 9.70 /s, M f32 2.78 -> 2.79, but alpha at cond(K) = 7e4 leaves the plain 1e-4 bar (8e-5 -> 1.03e-4, tests/test_gpu_fullsize.py): stays 32 */
#endif
constexpr int F32_CHUNK = HBEGP_F32_CHUNK;  // contraction elements per f32 accumulation chunk (see gemm_kernel); boundaries at absolute multiples

template <typename T, int TILE, int KM = 0>
struct GemmGeom {
  using C = Cfg<T>;
  // contraction depth per LDS stage: 128 bytes of k for the big tiles, 512 bytes for the 32-tiles (small, latency-bound
  // launches: fewer, fatter stages)
#ifndef HBEGP_KMUL32
#define HBEGP_KMUL32 2  /* measured in the 3-stream bench: 1 -> 1.104, 2 -> 1.127, 4 -> 1.054 fit+predict/s */
#endif
#ifndef HBEGP_KMUL64
#define HBEGP_KMUL64 1
#endif
  static constexpr int BKE = (KM > 0 ? KM : (TILE == 32 ? HBEGP_KMUL32 : (TILE == 64 ? HBEGP_KMUL64 : 1))) * C::BK;
  static constexpr int SK = BKE + 2;          // LDS row stride, operand stored [outer][k]
  static constexpr int SM = TILE + 16;        // LDS row stride, operand stored [k][outer]
  static constexpr int LDSE = (TILE * SK > BKE * SM) ? TILE * SK : BKE * SM;  // elements per operand buffer
  static constexpr int NCH = TILE * BKE / C::VEC / 256;  // 16-byte chunks per thread per operand per stage
  static constexpr int TM = TILE / 32;        // 16x16 MFMA blocks per wave per dimension
};

__device__ __forceinline__ int tri_row(int idx) {
  // largest r with r(r+1)/2 <= idx
  int r = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
  while ((r + 1) * (r + 2) / 2 <= idx) ++r;
  while (r * (r + 1) / 2 > idx) --r;
  return r;
}

#ifndef HBEGP_XBAR
#define HBEGP_XBAR 1  /* 64-tile: 64.7 -> 66.9 TFLOP/s at 3 workgroups per CU, 64.0 -> 66.5 at 2, 55.0 -> 61.5 at 1; 32-tile: 55.0 -> 52.9 */
#endif
#ifndef HBEGP_T128_TWOSETS
#define HBEGP_T128_TWOSETS 0  /* two register sets spill at 128 accumulator registers: 52.9 vs 59.7 TFLOP/s */
#endif
template <typename T, int TILE, int KM = 0>
__global__ void __launch_bounds__(256, 2) gemm_kernel(GemmLaunch g) {
  using C = Cfg<T>;
  using G = GemmGeom<T, TILE, KM>;
  using vec_t = typename C::vec_t;
  using acc_t = typename C::acc_t;
  constexpr int VEC = C::VEC, BK = G::BKE, SK = G::SK, SM = G::SM, NCH = G::NCH, TM = G::TM;

  // The not-positive-definite flag is only needed before anything is written: load it now, test it in the epilogue, so
  // its latency overlaps the operand loads (work on garbage is harmless, stores are not).
  const int bad = *g.info;

  // Work list of this workgroup: either one tile decoded from blockIdx (plain launches) or a host-built static
  // schedule (longest-processing-time assignment of tiles to a fixed number of resident workgroups, see hbegp.cpp).
  int it = 0, it_end = 1;
  if (g.sched_off) {
    it = g.sched_off[blockIdx.x];
    it_end = g.sched_off[blockIdx.x + 1];
  }
  // Kernel-argument reads are scalar loads with a few hundred cycles of latency each.  Everything the decoding needs is
  // fetched in two batches -- the per-op tile counts at fixed offsets, then the whole descriptor of the chosen op by
  // value -- instead of one dependent load per field (measured: 3000 cycles from kernel entry to the first operand load).
  const int nops = g.nops;
  const int nt0 = g.op[0].ntiles, nt1 = g.op[1].ntiles, nt2 = g.op[2].ntiles;
  for (; it < it_end; ++it) {
  int oi = 0, li = 0, lj = 0, bid = blockIdx.x;
  if (g.sched_off) {
    const unsigned item = g.sched_items[it];
    oi = item >> 30;
    li = (item >> 16) & 0x3fff;
    lj = item & 0xffff;
  } else {
    if (nops > 1 && bid >= nt0) {
      bid -= nt0; oi = 1;
      if (nops > 2 && bid >= nt1) {
        bid -= nt1; oi = 2;
        if (nops > 3 && bid >= nt2) { bid -= nt2; oi = 3; }
      }
    }
  }
  const GemmOp op = g.op[oi];
  if (!g.sched_off) {
    if (op.reverse) bid = op.ntiles - 1 - bid;  // heaviest tiles first
    if (op.c_lower) {
      li = tri_row(bid);
      lj = bid - li * (li + 1) / 2;
    } else {
      li = bid / op.nj;
      lj = bid - li * op.nj;
    }
  }
  const int ti = op.ci0 + li, tj = op.cj0 + lj;

  int ka = op.k0, kb = op.k1;
  switch (op.klim) {
    case 1: kb = min(kb, tj + 1); break;
    case 2: ka = max(ka, tj); break;
    case 3: kb = min(kb, ti + 1); break;
    case 4: ka = max(ka, ti); break;
    default: break;
  }
  // contraction range in elements, rounded outward to whole stages.  The extension stays inside the same 128-block:
  // there a triangular operand is zero in memory (its strict upper triangle, see load_stage) and the other operand is finite.
  const int kbeg = (ka * TILE) / BK * BK, kend = (kb * TILE + BK - 1) / BK * BK;
  const int nstages = (kend - kbeg) / BK;

  extern __shared__ __align__(16) char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);  // [A buf0 | A buf1 | B buf0 | B buf1], LDSE elements each

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int akm = op.a_kmajor, bkm = op.b_kmajor;
  const int lda = op.lda, ldb = op.ldb;

  // Per-thread chunk 0 of a stage slab: storage (row, col) relative to the slab origin; chunk q = chunk 0 + q row passes.
  constexpr int CPR = TILE / VEC;        // chunks per slab row, operand stored [k][outer]
  constexpr int CPK = BK / VEC;          // chunks per slab row, operand stored [outer][k]
  constexpr int RPP_K = 256 / CPK;       // slab rows covered per pass, operand stored [outer][k]
  constexpr int RPP_M = 256 / CPR;       // slab rows covered per pass, operand stored [k][outer]
  const int a_r0 = akm ? t / CPR : t / CPK, a_c0 = akm ? (t % CPR) * VEC : (t % CPK) * VEC;
  const int b_r0 = bkm ? t / CPR : t / CPK, b_c0 = bkm ? (t % CPR) * VEC : (t % CPK) * VEC;
  const int a_rpp = akm ? RPP_M : RPP_K, b_rpp = bkm ? RPP_M : RPP_K;
  const int a_lds0 = a_r0 * (akm ? SM : SK) + a_c0, a_ldsq = a_rpp * (akm ? SM : SK);
  const int b_lds0 = 2 * G::LDSE + b_r0 * (bkm ? SM : SK) + b_c0, b_ldsq = b_rpp * (bkm ? SM : SK);

  // global pointers of chunk 0 at stage 0; they advance by a uniform stride per stage
  const T* pA = static_cast<const T*>(op.A) +
                (akm ? (size_t)(kbeg + a_r0) * lda + ti * TILE + a_c0 : (size_t)(ti * TILE + a_r0) * lda + kbeg + a_c0);
  const T* pB = static_cast<const T*>(op.B) +
                (bkm ? (size_t)(kbeg + b_r0) * ldb + tj * TILE + b_c0 : (size_t)(tj * TILE + b_r0) * ldb + kbeg + b_c0);
  const size_t a_step = akm ? (size_t)BK * lda : (size_t)BK, b_step = bkm ? (size_t)BK * ldb : (size_t)BK;
  const size_t a_qs = (size_t)a_rpp * lda, b_qs = (size_t)b_rpp * ldb;

  // two register sets: the loads of stage s+2 are in flight while stage s is computed and stage s+1 sits in LDS
  vec_t ra0[NCH], rb0[NCH], ra1[NCH], rb1[NCH];

  // Triangular operands (X = L^-1 in W2 / the model's copy) need no masking: the strict upper triangle of those buffers
  // is zero in memory (cleared when the slot is created; the diagonal blocks write explicit zeros above the diagonal).
  auto load_stage = [&](vec_t (&ra)[NCH], vec_t (&rb)[NCH]) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) ra[q] = *reinterpret_cast<const vec_t*>(pA + q * a_qs);
#pragma unroll
    for (int q = 0; q < NCH; ++q) rb[q] = *reinterpret_cast<const vec_t*>(pB + q * b_qs);
    pA += a_step;
    pB += b_step;
  };
  auto store_stage = [&](int buf, vec_t (&ra)[NCH], vec_t (&rb)[NCH]) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      C::lds_store(lds + buf * G::LDSE + a_lds0 + q * a_ldsq, ra[q]);
      C::lds_store(lds + buf * G::LDSE + b_lds0 + q * b_ldsq, rb[q]);
    }
  };

  acc_t acc[TM][TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
  // f32 only: the MFMA accumulates in f32, and a long sum of products of one sign (the diagonals of T T^T and X^T X) loses
  // every term below half an ulp of the running sum -- a systematic loss, measured as a -2e-4 relative bias of K^-1 at
  // n=2048, cond 7e4 (LAPACK f32: -4e-5).  Every F32_CHUNK contraction elements the accumulators are added to fp64 totals
  // and cleared (chunk 32: bias -3e-6 in a host emulation).  Chunk boundaries are absolute multiples of F32_CHUNK, so every
  // tile size, and the task-queue kernel, produce the same bits.
  constexpr bool CHUNKED = sizeof(T) == 4;
  double tot[CHUNKED ? TM : 1][CHUNKED ? TM : 1][4];
  if constexpr (CHUNKED) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[a][b][r] = 0.0;
  }
  auto flush = [&]() {
    if constexpr (CHUNKED) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[a][b][r] += (double)acc[a][b][r];
          acc[a][b] = acc_t{0, 0, 0, 0};
        }
    }
  };
  int kabs = kbeg;  // f32: absolute contraction index of the stage being computed (chunk boundaries are absolute multiples of F32_CHUNK)

  // fragment addressing: element (outer index o, contraction k) at o*so + k*sk
  const int soA = akm ? 1 : SK, skA = akm ? SM : 1;
  const int soB = bkm ? 1 : SK, skB = bkm ? SM : 1;
  const int fa0 = (wm * (TILE / 2) + (lane & 15)) * soA + (lane >> 4) * skA;
  const int fb0 = 2 * G::LDSE + (wn * (TILE / 2) + (lane & 15)) * soB + (lane >> 4) * skB;

  // Fragments are double-buffered in registers: the LDS reads of k-step k4+1 are issued before the MFMAs of k-step k4,
  // otherwise every k-step exposes one LDS round trip (the MFMAs of a step are issued long before they finish, but the
  // next reads cannot start until the last of them has been issued).
  auto compute_stage = [&](int buf) {
    const int ia = buf * G::LDSE + fa0, ib = buf * G::LDSE + fb0;
    T af[2][TM], bf[2][TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) af[0][a] = lds[ia + a * 16 * soA];
#pragma unroll
    for (int b = 0; b < TM; ++b) bf[0][b] = lds[ib + b * 16 * soB];
#pragma unroll
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      const int cur = k4 & 1, nxt = cur ^ 1;
      if (k4 + 1 < BK / 4) {
#pragma unroll
        for (int a = 0; a < TM; ++a) af[nxt][a] = lds[ia + a * 16 * soA + (k4 + 1) * 4 * skA];
#pragma unroll
        for (int b = 0; b < TM; ++b) bf[nxt][b] = lds[ib + b * 16 * soB + (k4 + 1) * 4 * skB];
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = C::mfma(af[cur][a], bf[cur][b], acc[a][b]);
      // pin the order: [LDS reads of step k4+1] then [MFMAs of step k4] (hipcc otherwise sinks half of the reads below
      // the MFMAs to save registers)
      if (k4 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM, 0);
      if (k4 + 1 < BK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TM, 0);
      if (CHUNKED && (kabs + (k4 + 1) * 4) % F32_CHUNK == 0) flush();
    }
    kabs += BK;
  };

  if constexpr (TILE == 128 && !HBEGP_T128_TWOSETS) {
    // one register set (the 128-tile already holds 128 accumulator registers): loads of stage s+1 fly during stage s
    if (nstages > 0) {
      load_stage(ra0, rb0);
      store_stage(0, ra0, rb0);
    }
    __syncthreads();
    int s = 0;
    for (; s + 1 < nstages; ++s) {
      const int buf = s & 1;
      load_stage(ra0, rb0);
      compute_stage(buf);
      __builtin_amdgcn_sched_barrier(0);
      store_stage(buf ^ 1, ra0, rb0);
      __syncthreads();
    }
    if (s < nstages) {
      compute_stage(s & 1);
      __syncthreads();
    }
  } else {
    if constexpr (TILE == 64 && HBEGP_XBAR) {
    // Software pipelining across the stage barrier: the fragments of a whole stage live in registers, the LDS stores of
    // the next stage are issued before the last-but-one k-step and the first fragments of the next stage are read
    // right after the barrier, under the MFMAs of the last k-step -- so neither the store latency nor the first read
    // latency sits between a barrier and an MFMA.
    constexpr int NK = BK / 4;
    T fa[NK][TM], fb[NK][TM];
    auto read_frags = [&](int buf, int k4) {
      const int ia = buf * G::LDSE + fa0 + k4 * 4 * skA, ib = buf * G::LDSE + fb0 + k4 * 4 * skB;
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[k4][a] = lds[ia + a * 16 * soA];
#pragma unroll
      for (int b2 = 0; b2 < TM; ++b2) fb[k4][b2] = lds[ib + b2 * 16 * soB];
    };
    auto mfma_step = [&](int k4) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b2 = 0; b2 < TM; ++b2) acc[a][b2] = C::mfma(fa[k4][a], fb[k4][b2], acc[a][b2]);
    };
    // one stage; `cur` = LDS buffer of this stage.  On entry fa/fb[0] hold its first k-step.
    auto stage = [&](int cur, bool do_load, vec_t (&la)[NCH], vec_t (&lb)[NCH], bool do_store, vec_t (&sa)[NCH],
                     vec_t (&sb)[NCH], bool has_next) {
      if (do_load) load_stage(la, lb);
#pragma unroll
      for (int k4 = 1; k4 < NK; ++k4) read_frags(cur, k4);
#pragma unroll
      for (int k4 = 0; k4 + 2 < NK; ++k4) mfma_step(k4);
      __builtin_amdgcn_sched_barrier(0);
      if (do_store) store_stage(cur ^ 1, sa, sb);
      if (NK >= 2) mfma_step(NK - 2);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // fragments of k-step NK-1 are still needed: the next stage's first fragments go to slot 0
      if (has_next) read_frags(cur ^ 1, 0);
      mfma_step(NK - 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM, 0);  // reads first: their latency hides under the MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TM, 0);
      __builtin_amdgcn_sched_barrier(0);
      static_assert(!CHUNKED || F32_CHUNK % BK == 0, "f32 64-tile: an accumulation chunk is a whole number of stages");
      kabs += BK;
      if (CHUNKED && kabs % F32_CHUNK == 0) flush();
    };
    if (nstages > 0) {
      load_stage(ra0, rb0);
      if (nstages > 1) load_stage(ra1, rb1);
      store_stage(0, ra0, rb0);
      __syncthreads();
      read_frags(0, 0);
      int s = 0;
      for (; s + 3 < nstages; s += 2) {
        stage(0, true, ra0, rb0, true, ra1, rb1, true);
        stage(1, true, ra1, rb1, true, ra0, rb0, true);
      }
      const int left = nstages - s;  // 1..3
      stage(0, left > 2, ra0, rb0, left > 1, ra1, rb1, left > 1);
      if (left > 1) stage(1, false, ra1, rb1, left > 2, ra0, rb0, left > 2);
      if (left > 2) stage(0, false, ra0, rb0, false, ra1, rb1, false);
      __syncthreads();  // the next tile of this workgroup overwrites LDS
    }
    } else {
    // prologue: stage 0 -> LDS buffer 0, stage 1 -> register set 1
    if (nstages > 0) {
      load_stage(ra0, rb0);
      if (nstages > 1) load_stage(ra1, rb1);
      store_stage(0, ra0, rb0);
    }
    __syncthreads();
    // steady state, unrolled by two so that the register sets are indexed statically:
    //   even s: loads(s+2) -> set 0 | compute LDS[0] | set 1 (stage s+1) -> LDS[1] | barrier
    //   odd  s: loads(s+2) -> set 1 | compute LDS[1] | set 0 (stage s+1) -> LDS[0] | barrier
    // Two things keep the loads two stages ahead of their use in the generated code:
    //  * the steady-state loop issues its loads unconditionally (the last stages are peeled off below): a load under an
    //    `if` makes hipcc's wait-count insertion assume the no-load path and emit vmcnt(0) before the LDS stores, i.e.
    //    wait for the loads issued a moment ago;
    //  * the scheduling fences keep it from hoisting the LDS stores (and their vmcnt wait) above the stage's MFMAs.
    // Left alone the exposed load latency costs 11 % (3 workgroups per CU) to 20 % (1 per CU) of the MFMA rate.
    int s = 0;
    for (; s + 3 < nstages; s += 2) {
      load_stage(ra0, rb0);
      compute_stage(0);
      __builtin_amdgcn_sched_barrier(0);
      store_stage(1, ra1, rb1);
      __syncthreads();
      load_stage(ra1, rb1);
      compute_stage(1);
      __builtin_amdgcn_sched_barrier(0);
      store_stage(0, ra0, rb0);
      __syncthreads();
    }
    // tail: 1..3 stages left (0 if the tile has none); LDS[0] holds stage s, register set 1 holds stage s+1
    if (s < nstages) {
      const int left = nstages - s;
      if (left > 2) load_stage(ra0, rb0);
      compute_stage(0);
      __builtin_amdgcn_sched_barrier(0);
      if (left > 1) store_stage(1, ra1, rb1);
      __syncthreads();
      if (left > 1) {
        compute_stage(1);
        __builtin_amdgcn_sched_barrier(0);
        if (left > 2) store_stage(0, ra0, rb0);
        __syncthreads();
        if (left > 2) {
          compute_stage(0);
          __syncthreads();
        }
      }
    }
    }
  }

  // epilogue
  flush();  // f32: what the last (partial) chunk holds
  if (bad != 0) return;
  T* Cg = static_cast<T*>(op.C);
  const int row0 = ti * TILE + wm * (TILE / 2), col0 = tj * TILE + wn * (TILE / 2) + (lane & 15);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + a * 16 + C::crow(lane, r);
        T* p = Cg + (size_t)row * op.ldc + col0 + b * 16;
        T v;
        if constexpr (CHUNKED) v = (T)tot[a][b][r];
        else v = acc[a][b][r];
        if (op.alpha_neg) v = -v;
        if (op.beta_one) v += *p;
        *p = v;
      }
  }  // work list
}

template <typename T, int TILE>
static void launch_gemm_t(const GemmLaunch& gl, hipStream_t s) {
  using G = GemmGeom<T, TILE>;
  GemmLaunch g = gl;
  int total = 0;
  for (int i = 0; i < g.nops; ++i) {
    GemmOp& op = g.op[i];
    op.ntiles = op.c_lower ? op.mi * (op.mi + 1) / 2 : op.mi * op.nj;
    // heaviest tiles first: contraction depth grows with tj (klim 1) or ti (klim 3) -> walk those tile lists backwards
    op.reverse = (op.klim == 1 || op.klim == 3) ? 1 : 0;
    total += op.ntiles;
  }
  if (g.sched_off) total = g.sched_nwg;
  if (total <= 0) return;
  size_t lds = (size_t)4 * G::LDSE * sizeof(T);
  // Static schedules assume exactly sched_nwg / 256 workgroups on every CU: pin that residency by asking for the matching
  // share of the 160 KiB LDS (otherwise the dispatcher may stack 3 workgroups on some CUs and 1 on others).
  if (g.sched_off && g.sched_nwg >= 256) {
    const size_t share = ((size_t)163840 / (size_t)(g.sched_nwg / 256)) / 4096 * 4096;  // LDS is allocated in coarse granules
    if (share > lds) lds = share;
  }
  static const size_t lds_min = getenv("HBEGP_GEMM_LDS_MIN") ? (size_t)atol(getenv("HBEGP_GEMM_LDS_MIN")) : 0;  // experiments: cap residency
  if (lds_min > lds) lds = lds_min;
  hipLaunchKernelGGL((gemm_kernel<T, TILE>), dim3(total), dim3(256), lds, s, g);
}

// ops are specified in NB(=128)-tile units by the host; rescale to the launch tile size here.
template <typename T>
void launch_gemm(const GemmLaunch& gl, int tile, hipStream_t s) {
  GemmLaunch g = gl;
  const int f = NB / tile;
  for (int i = 0; i < g.nops; ++i) {
    GemmOp& op = g.op[i];
    op.ci0 *= f; op.cj0 *= f; op.mi *= f; op.nj *= f; op.k0 *= f; op.k1 *= f;
  }
  if (tile == 128) launch_gemm_t<T, 128>(g, s);
  else if (tile == 64) launch_gemm_t<T, 64>(g, s);
  else launch_gemm_t<T, 32>(g, s);
}
template void launch_gemm<double>(const GemmLaunch&, int, hipStream_t);
template void launch_gemm<float>(const GemmLaunch&, int, hipStream_t);

// =================================================================================================================
// Leaf: 128x128 diagonal block.  L = chol(A) (lower), X = L^-1.
// =================================================================================================================
template <typename T>
struct LeafGeom {
  static constexpr int S = 130;          // LDS row stride of the block image (and of the image of the 16x16 inverses)
  static constexpr int YS = 18;          // row stride of a 16x16 scratch image
  static constexpr int YB = 16 * YS;     // elements per 16x16 scratch image
  static constexpr size_t LDS_BYTES = (size_t)(128 * S + 16 * S + 4 * YB + 128) * sizeof(T);  // 160,000 B for f64
};

// ---- static item tables of the diagonal block's MFMA waves (see leaf_body) ---------------------------------------------------
// Every 16x16x16 product the block needs is known in advance.  Per phase p the products of panels k = p-1 and k2 = p-2 are
// listed class by class; an entry is four words {a, b, c, x}: byte offsets (LDS, from the block image) of the A operand, the B
// operand and the result block, and one more number.  A wave reads an entry with ONE broadcast LDS read and adds its lane
// pattern -- no scalar decoding, no class switch inside a loop: one wave issues one instruction per ~4.5 cycles whatever its kind,
// so an item must stay below ~50 instructions to keep up with its four MFMAs (256 cycles of the SIMD's matrix pipe).
//   F(j):      X[k,j] = -Y_k S[k,j], j < k         {j, S^T block (j,k), same, byte offset of column block j in a row of X}
//   U(i,j):    A[i,j] -= L[i,k] L[j,k]^T           {L block (i,k), L block (j,k), block (i,j), 0},   p+1 <= j <= i <= 7
//   G(i,j):    S[i,j] += L[i,k2] X[k2,j]           {L block (i,k2), X^T block (j,k2) or Yt_k2, S^T block (j,i), 0},  i >= p+1, j <= k2
//   GU1(j):    S[p,j] += L[p,k2] X[k2,j], j <= k2  (as G)
//   GU2(j):    S[p,j] += L[p,k] X[k,j],  j <= k    (as G; for j < k it waits for F(j)'s flag)
#ifndef LEAF_WAVE4_WORKS
#define LEAF_WAVE4_WORKS 1  /* wave 4 shares a SIMD with wave 0 and slows its elimination by a fifth (2370 -> 2870-3140 cycles), but the MFMA items are what a phase waits for: 25.8 -> 24.7 us per block */
#endif
constexpr int LEAF_S = 130;
constexpr int LEAF_YT0 = 128 * LEAF_S;       // element offset of the image of the transposed 16x16 inverses: Y_q[r][c] at YT0 + c S + 16 q + r
constexpr int LEAF_MAXITEMS = 26;
struct LeafItemTab {
  unsigned e[8][LEAF_MAXITEMS][4];
  unsigned par[8][8][4];  // [phase][wave]: {first item of this wave in the F | U << 8 | G << 16 | GU << 24 list, number of MFMA waves, 0, 0}
};
constexpr int LEAF_TAB_WORDS = 8 * LEAF_MAXITEMS * 4 + 8 * 8 * 4;
constexpr unsigned leaf_blk_bytes(int i, int j) { return (unsigned)(((i * 16) * LEAF_S + j * 16) * 8); }
constexpr unsigned leaf_yt_bytes(int q) { return (unsigned)((LEAF_YT0 + 16 * q) * 8); }
// list sizes and offsets inside a phase's row (closed forms: leaf_body uses them instead of loading counts)
constexpr int leaf_nf(int p) { return p - 1; }
constexpr int leaf_nu(int p) { return (7 - p) * (8 - p) / 2; }
constexpr int leaf_ng(int p) { return p >= 2 ? (7 - p) * (p - 1) : 0; }
constexpr int leaf_ngu1(int p) { return p - 1; }
constexpr int leaf_ngu2(int p) { return p; }
constexpr LeafItemTab leaf_build_items(int xcol_bytes) {
  LeafItemTab t{};
  for (int p = 1; p < 8; ++p) {
    const int k = p - 1, k2 = p - 2;
    int n = 0;
    for (int j = 0; j < k; ++j, ++n) {
      t.e[p][n][0] = (unsigned)j; t.e[p][n][1] = leaf_blk_bytes(j, k); t.e[p][n][2] = leaf_blk_bytes(j, k); t.e[p][n][3] = (unsigned)(j * xcol_bytes);
    }
    for (int i = p + 1; i < 8; ++i)
      for (int j = p + 1; j <= i; ++j, ++n) {
        t.e[p][n][0] = leaf_blk_bytes(i, k); t.e[p][n][1] = leaf_blk_bytes(j, k); t.e[p][n][2] = leaf_blk_bytes(i, j);
      }
    if (k2 >= 0)
      for (int i = p + 1; i < 8; ++i)
        for (int j = 0; j <= k2; ++j, ++n) {
          t.e[p][n][0] = leaf_blk_bytes(i, k2); t.e[p][n][1] = j == k2 ? leaf_yt_bytes(k2) : leaf_blk_bytes(j, k2); t.e[p][n][2] = leaf_blk_bytes(j, i);
        }
    for (int j = 0; j <= k2; ++j, ++n) {
      t.e[p][n][0] = leaf_blk_bytes(p, k2); t.e[p][n][1] = j == k2 ? leaf_yt_bytes(k2) : leaf_blk_bytes(j, k2); t.e[p][n][2] = leaf_blk_bytes(j, p);
    }
    for (int j = 0; j <= k; ++j, ++n) {
      t.e[p][n][0] = leaf_blk_bytes(p, k); t.e[p][n][1] = j == k ? leaf_yt_bytes(k) : leaf_blk_bytes(j, k); t.e[p][n][2] = leaf_blk_bytes(j, p);
    }
    t.e[p][LEAF_MAXITEMS - 1][3] = (unsigned)n;  // for the consistency check below only
    // Round-robin dealing of the phase's items to its MFMA waves, continuing across the lists: waves 1, 2, 3, 5, 7 while wave 6
    // is the helper (p < 4), then 1, 2, 3, 5, 6, 7; wave 4 (it shares a SIMD with the eliminating wave 0) comes last.  The
    // first item of every wave in every list is tabulated: no division at run time (an integer modulo costs ~30
    // instructions -- more than half an item).
    const int nw = (p < 4 ? 5 : 6) + LEAF_WAVE4_WORKS;
    const int starts[4] = {0, leaf_nf(p), leaf_nf(p) + leaf_nu(p), leaf_nf(p) + leaf_nu(p) + leaf_ng(p)};  // GU: its own numbering (j)
    for (int w = 0; w < 8; ++w) {
      const int widx = w <= 3 ? w - 1 : (w == 4 ? nw - 1 : (w == 5 ? 3 : (p < 4 ? 4 : w - 2)));
      unsigned packed = 0;
      for (int l = 0; l < 4; ++l) {
        int f = widx - starts[l] % nw;
        if (f < 0) f += nw;
        packed |= (unsigned)(widx < 0 ? 255 : f) << (8 * l);
      }
      t.par[p][w][0] = packed;
      t.par[p][w][1] = (unsigned)nw;
    }
  }
  return t;
}
constexpr bool leaf_items_consistent() {
  const LeafItemTab t = leaf_build_items(128);
  for (int p = 1; p < 8; ++p) {
    const int n = leaf_nf(p) + leaf_nu(p) + leaf_ng(p) + leaf_ngu1(p) + leaf_ngu2(p);
    if ((int)t.e[p][LEAF_MAXITEMS - 1][3] != n || n > LEAF_MAXITEMS - 1) return false;
  }
  return true;
}
static_assert(leaf_items_consistent(), "item table sizes");
// copied into the LDS by every block (a scalar load from HBM per item would cost more than the item); one table per element type
// of X in HBM (the F items carry the byte offset of a column block)
__device__ const LeafItemTab g_leaf_items_f64 = leaf_build_items(16 * 8);
__device__ const LeafItemTab g_leaf_items_f32 = leaf_build_items(16 * 4);

// ---- lane broadcasts inside a 16-lane row, fused into the fp64 FMA (round 4) ------------------------------------------------
// gfx950 carries DPP for the double-precision ALU with one control: row_newbcast:n = "lane n of my 16-lane row".  With the 16
// pivot rows of a panel replicated in every 16-lane row of the wave, the multiplier l_jk every lane needs at pivot k is lane j of
// its own row: the trailing update a_ij -= l_ik l_jk is ONE v_fmac_f64_dpp (4.7 cycles issue, tools/leaf_ubench) where rounds 1-3
// spent two v_readlane_b32 + s_nop + v_fma_f64 (20 cycles) -- the elimination's instruction stream shrinks 2.5x.
// The compiler knows nothing about what is inside an asm statement, in particular not that a DPP operand read needs two wait
// states behind the VALU write of that register (CDNA3 ISA 4.5): every value that is read through DPP is therefore produced by
// mul_then_gap() / consumed by row_bcast_safe(), which carry the wait states inside their own asm text.
template <int J>
__device__ __forceinline__ void fmac_nbc(double& d, double s0, double s1) {  // d -= (lane J of my row: s0) * s1
  asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s0), "v"(s1), "n"(J));
}
template <int K>
__device__ __forceinline__ double row_bcast_safe(double v) {  // lane K of my row's v; v may have been written by the previous instruction
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
__device__ __forceinline__ double mul_then_gap(double a, double b) {  // a * b, safe to read through DPP right behind it
  double r;
  asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// One 16-column panel in registers.  a[]: the panel's 16 pivot rows, lane l holds row l & 15 (four replicas per wave);
// b[]: 64 more rows of the panel, one per lane (rows below the pivot block, or rows of the identity).  At pivot K every lane
// forms 1/sqrt(pivot) itself (the pivot is lane K of its row), scales its two columns K and applies the rank-1 update to the
// columns behind.  A row of b ends as b L_pp^-T: a row below the pivot block becomes its row of L, row r of the identity becomes
// column r of Y = L_pp^-1 (exact zeros above the diagonal) -- the inverse of the 16x16 factor costs no instruction of its own.
// Entries above the diagonal of the pivot block only ever feed themselves: no per-lane predicates.
// 1/sqrt(piv): hardware estimate (5e-8) + one third-order step r (1 + e/2 + 3e^2/8), e = 1 - piv r^2: 1.4e-16, the same as two
// Newton steps (tools/rsq_probe.hip) in 5 dependent operations instead of 8.  A pivot that is not positive (or NaN) gives NaN for
// its column and everything behind it: the caller tests the diagonal of L once, at the end.
template <int K, int J>
struct LeafCol {
  static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16], double lka, double lkb) {
    fmac_nbc<J>(a[J], lka, lka);
    fmac_nbc<J>(b[J], lka, lkb);
    LeafCol<K, J + 1>::run(a, b, lka, lkb);
  }
};
template <int K>
struct LeafCol<K, 16> {
  static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double, double) {}
};
template <int K>
struct LeafPivot {
  static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16]) {
    const double piv = row_bcast_safe<K>(a[K]);
    double rinv = __builtin_amdgcn_rsq(piv);
    {
      const double gg = piv * rinv;
      const double ee = __builtin_fma(-gg, rinv, 1.0);
      const double pp = __builtin_fma(ee, 0.375, 0.5);
      rinv = __builtin_fma(rinv, ee * pp, rinv);
    }
    const double lka = mul_then_gap(a[K], rinv);  // l_{row,K} (pivot row: sqrt(piv))
    const double lkb = b[K] * rinv;
    a[K] = lka;
    b[K] = lkb;
    LeafCol<K, K + 1>::run(a, b, lka, lkb);
    LeafPivot<K + 1>::run(a, b);
  }
};
template <>
struct LeafPivot<16> {
  static __device__ __forceinline__ void run(double (&)[16], double (&)[16]) {}
};

// Structure (8 panels of 16 columns, two barrier-separated phases per panel p):
//   (A) wave 0 eliminates the panel in registers (LeafPivot): pivot rows 16p..16p+15 replicated four times, plus the next (up to)
//       64 rows.  While rows remain beyond that window (p < 4) one helper wave (wave 6) runs the SAME elimination -- redundantly,
//       bitwise the same -- with the remaining rows, so they are ready together with the window without any communication.  The
//       first 16 spare lanes of the helper (p < 4) or of wave 0 (p >= 4) carry the rows of the identity: Y_p = L_pp^-1.
//       Meanwhile the other waves pull 16x16x16 MFMA items of panel p-1 from a pool (an LDS counter; an item's arithmetic does not
//       depend on who runs it): the rest of the trailing update, and the inverse X = L^-1 in right-looking form --
//         F(j):   X[k,j] = -Y_k S[k,j]            (block row k = p-1 of X is final)
//         G(i,j): S[i,j] += L[i,k] X[k,j], i > k  (every later block row collects its sum as soon as a term exists)
//       so that behind the last panel only X[7,j] = -Y_7 S[7,j] is left (rounds 1-3 formed whole block rows of X at the end:
//       4,900 of the block's 53,000 cycles).  S[i,j] and later X[i,j] live TRANSPOSED in the strict upper blocks of the image,
//       which is exactly the layout of an MFMA B operand: nothing is staged through a scratch image.
//   (B) block column p+1 is updated for every remaining block row (one 16x16 block per wave).
// fp64 MFMA for every 16x16x16 product, two accumulator chains per product (a dependent v_mfma_f64_16x16x4 issues every 66
// cycles, tools/leaf_ubench).  Block indices are wave-uniform (scalar registers); a lane's part of every address is one of four
// constants.
// T = arithmetic type of the block (always double: the block is latency-bound, so f32 problems are factored in f64 too),
// TIO = element type in HBM.
__device__ long long g_leaf_stamps[256];
#define LEAF_STAMP(i) do { if ((dbg & 8) && t == 0) g_leaf_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
void read_leaf_stamps(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_leaf_stamps), sizeof(long long) * 256); }

// Results another workgroup of the SAME launch will read (the task-queue kernel, dag_kernel.inc.hpp) are stored
// write-through (sc1) so that they need no release fence; between launches plain stores do.
template <bool SC1, typename TIO>
__device__ __forceinline__ void gstore(TIO* p, TIO v) {
  if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// FROM_LDS: the block image is in the LDS already (lower 16x16 blocks valid, strict upper blocks zero; the caller has met at a
// barrier): small_eval_kernel assembles the kernel matrix there and never writes it to HBM.
template <typename T, typename TIO, bool SC1, bool FROM_LDS = false>
__device__ __forceinline__ void leaf_body(TIO* __restrict__ W1, TIO* __restrict__ W2, int ld, int blk,
                                          TIO* __restrict__ ldiag, int* info, int dbg, char* smem_raw,
                                          TIO* __restrict__ W3 = nullptr) {
  static_assert(sizeof(T) == 8, "the diagonal block is factored in fp64");
  using C = Cfg<T>;
  using L = LeafGeom<T>;
  using acc_t = typename C::acc_t;
  using vec_t = typename Cfg<TIO>::vec_t;
  constexpr int S = L::S, YS = L::YS, YB = L::YB, VEC = Cfg<TIO>::VEC;

  T* As = reinterpret_cast<T*>(smem_raw);  // [128][S]: lower = A -> L ; strict upper blocks = S^T -> X^T
  T* Yt = As + 128 * S;                    // [16][S]: TRANSPOSED inverses of the diagonal 16x16 factors, Y_q[r][c] at Yt[c * S + 16 q + r]
  static_assert(S == LEAF_S, "the item tables are built for this row stride");
  T* Sc = Yt + 16 * S;                     // [16][YS]: copy of the next panel's pivot block for the helper wave
  T* Id = Sc + YB;                         // [17][YS]: the identity (rows 0-15) and a row of zeros (row 16)
  int* pool = reinterpret_cast<int*>(Id + 17 * YS + 2);  // [8]: flags of the F items (+ 8 spare)
  unsigned* tab = reinterpret_cast<unsigned*>(pool + 16);  // the item tables (g_leaf_items_*), 16-byte aligned
  static_assert(((128 * S + 16 * S + YB + 17 * YS + 2) * sizeof(T) + 64) % 16 == 0, "item table alignment");
  static_assert((YB + 17 * YS + 2) * sizeof(T) + 64 + LEAF_TAB_WORDS * 4 <= (4 * YB + 128) * sizeof(T), "the scratch region holds the item tables");

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform on purpose: block indices stay in scalar registers
  const int m16 = lane & 15, q4 = lane >> 4;
  const size_t g0 = (size_t)blk * NB * ld + (size_t)blk * NB;
  TIO* Ablk = W1 + g0;
  TIO* Xblk = W2 + g0;

  // Load the lower 16x16 blocks of the block; the strict upper blocks start as zeros (they will hold the sums S^T of the
  // inverse).  Wave w takes block row w (rows 16w .. 16w+15, four threads per row): the trip count is wave-uniform, and all
  // loads of a thread are issued before its first LDS store.
  if constexpr (!FROM_LDS) {
    const int r = t >> 2, sub = t & 3;           // row, position among the row's four threads
    constexpr int CPR = 16 / VEC;                // 16-byte chunks per 16 columns
    const int nch = (wave + 1) * CPR / 4;        // chunks of this thread: columns [0, 16 (wave + 1)) of its row, interleaved by 4
    const TIO* src = Ablk + (size_t)r * ld + sub * VEC;
    T* dst = As + r * S + sub * VEC;
    constexpr int MAXCH = 8 * CPR / 4;
    vec_t buf[MAXCH];
#pragma unroll
    for (int q = 0; q < MAXCH; ++q)
      if (q < nch) buf[q] = *reinterpret_cast<const vec_t*>(src + q * 4 * VEC);
#pragma unroll
    for (int q = 0; q < MAXCH; ++q) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[q * 4 * VEC + e] = q < nch ? (T)buf[q][e] : T(0);
    }
  }
  // the first diagonal 16x16 block once more, into Sc: the helper wave takes the pivot rows from there (see the elimination)
  if (t < 256) {
    if constexpr (FROM_LDS) Sc[(t >> 4) * YS + (t & 15)] = As[(t >> 4) * S + (t & 15)];
    else Sc[(t >> 4) * YS + (t & 15)] = (T)Ablk[(size_t)(t >> 4) * ld + (t & 15)];
  } else {
    for (int e = t - 256; e < 17 * 16; e += 256) Id[(e >> 4) * YS + (e & 15)] = ((e >> 4) == (e & 15)) ? T(1) : T(0);
    if (t < 256 + 16) pool[t - 256] = 0;  // the F flags
  }
  {
    const unsigned* gt = reinterpret_cast<const unsigned*>(sizeof(TIO) == 8 ? &g_leaf_items_f64 : &g_leaf_items_f32);
    for (int e = t; e < LEAF_TAB_WORDS; e += 512) tab[e] = gt[e];
  }
  __syncthreads();

  LEAF_STAMP(0);
  // ---- 16x16x16 MFMA items on 16x16 blocks of the LDS image.  A lane adds one of two patterns to an entry's byte offsets:
  //   P0 = row m16 of the block, element q4 (+ 4 per step): operands stored [outer][k], and a TRANSPOSED result
  //   P1 = row q4 (+ 4 per step), column m16: a result in normal orientation (U), and Y_q read from the Yt image as an A operand (F)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const int P0 = (m16 * S + q4) * (int)sizeof(T), P1 = (q4 * S + m16) * (int)sizeof(T);
  constexpr int STEP0 = 4 * (int)sizeof(T), STEP1 = 4 * S * (int)sizeof(T);
  constexpr int CLS_U = 0, CLS_G = 1, CLS_F = 2;
  char* lds0 = reinterpret_cast<char*>(As);
  int* fflag = pool;  // [8]: fflag[j] = p once F(j) of phase p has written X[p-1, j]
  auto ldsT = [&](int byte_off) -> T& { return *reinterpret_cast<T*>(lds0 + byte_off); };
  // An item in two halves, so that two items can be in flight in one wave: issue = operand loads + the four MFMAs on two
  // accumulator chains (a dependent v_mfma_f64_16x16x4 issues every 66 cycles), finish = combine, store.
  struct Prod {
    T cf[4];
    acc_t a0, a1;
  };
  // ya: byte offset of Y_k in the Yt image (F items: their A operand)
  auto item_issue = [&](auto cls, const u4 d, int ya, Prod& pr) {
    constexpr int CLS = decltype(cls)::value;
    T af[4], bf[4];
    if constexpr (CLS == CLS_F) {
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) af[k4] = ldsT(ya + P1 + k4 * STEP1);
    } else {
      const int pa = (int)d[0] + P0;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) af[k4] = ldsT(pa + k4 * STEP0);
    }
    const int pb = (int)d[1] + P0;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) bf[k4] = ldsT(pb + k4 * STEP0);
    if constexpr (CLS == CLS_U) {
      const int pc = (int)d[2] + P1;
#pragma unroll
      for (int r = 0; r < 4; ++r) pr.cf[r] = ldsT(pc + r * STEP1);
    } else if constexpr (CLS == CLS_G) {
      const int pc = (int)d[2] + P0;
#pragma unroll
      for (int r = 0; r < 4; ++r) pr.cf[r] = ldsT(pc + r * STEP0);
    }
    pr.a0 = acc_t{0, 0, 0, 0};
    pr.a1 = acc_t{0, 0, 0, 0};
    pr.a0 = C::mfma(af[0], bf[0], pr.a0);
    pr.a1 = C::mfma(af[2], bf[2], pr.a1);
    pr.a0 = C::mfma(af[1], bf[1], pr.a0);
    pr.a1 = C::mfma(af[3], bf[3], pr.a1);
  };
  // xrow: F items, this lane's first element of block row k of X in HBM (row q4 of the block row, column m16); dcopy: the
  // result is the next panel's pivot block, the helper wave will eliminate it a second time: keep a copy in Sc
  auto item_finish = [&](auto cls, const u4 d, const Prod& pr, TIO* xrow, bool dcopy) {
    constexpr int CLS = decltype(cls)::value;
    if constexpr (CLS == CLS_U) {
      const int pc = (int)d[2] + P1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T v = pr.cf[r] - (pr.a0[r] + pr.a1[r]);
        ldsT(pc + r * STEP1) = v;
        if (dcopy) Sc[(q4 + 4 * r) * YS + m16] = v;
      }
    } else if constexpr (CLS == CLS_G) {
      const int pc = (int)d[2] + P0;
#pragma unroll
      for (int r = 0; r < 4; ++r) ldsT(pc + r * STEP0) = pr.cf[r] + (pr.a0[r] + pr.a1[r]);
    } else {
      const int pc = (int)d[2] + P0;
      TIO* xp = reinterpret_cast<TIO*>(reinterpret_cast<char*>(xrow) + d[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T v = -pr.a0[r] - pr.a1[r];
        ldsT(pc + r * STEP0) = v;  // in place of S^T: every lane's operand reads are complete, the products have consumed them
        gstore<SC1>(xp + (size_t)(4 * r) * ld, (TIO)v);
      }
    }
  };
  using cU = std::integral_constant<int, CLS_U>;
  using cG = std::integral_constant<int, CLS_G>;
  using cF = std::integral_constant<int, CLS_F>;
  auto xrow_of = [&](int k) { return Xblk + (size_t)(k * 16 + q4) * ld + m16; };
  // Phase A(p) of the MFMA waves: the products of panels k = p-1 and k2 = p-2, class by class --
  //   F(j,k), j < k                       block row k of X is final (sets fflag[j] = p)
  //   U(i,j,k), p+1 <= j <= i <= 7        the rest of the trailing update of panel k
  //   G(i,j,k2), i >= p+1, j <= k2        the sums of the later block rows collect the terms of block row k2 of X (final since
  //                                       the previous phase; one phase late, so that they wait for nothing)
  //   GU(j), j <= k                       S[p,j] += L[p,k2] X[k2,j] and += L[p,k] X[k,j]: block row p is finalised in the NEXT
  //                                       phase, so its last term cannot be late -- it waits for F(j,k)'s flag instead
  // Items are dealt round-robin to the MFMA waves, continuing across the lists (an LDS atomic per claim cost ~300 cycles with
  // six waves asking at once -- more than an item; the dealing is tabulated, LeafItemTab::par).  F, U and G items are independent of one another: a wave keeps two of them in flight.  Every wave runs its F items
  // before its GU items, so a wait for an F flag always ends.
  auto pool_phase = [&](int p) {
    const int k = p - 1;
    // this wave's first item in each list and the number of MFMA waves: one broadcast read of the phase's parameter record
    const u4 par = *reinterpret_cast<const u4*>(tab + 8 * LEAF_MAXITEMS * 4 + (p * 8 + wave) * 4);
    const unsigned firsts = (unsigned)__builtin_amdgcn_readfirstlane((int)par[0]);
    const int nworkers = __builtin_amdgcn_readfirstlane((int)par[1]);
    const u4* tabp = reinterpret_cast<const u4*>(tab) + p * LEAF_MAXITEMS;
    const int ya = (int)leaf_yt_bytes(0) + k * 16 * (int)sizeof(T);
    TIO* xrow = xrow_of(k);
    auto first_of = [&](int list) { return (int)((firsts >> (8 * list)) & 255u); };
    auto run_list = [&](auto cls, int list, int o, int n) {
      const int first = first_of(list);
      for (int i = first; i < n; i += 2 * nworkers) {
        const bool two = i + nworkers < n;
        const u4 d0 = tabp[o + i];
        const u4 d1 = tabp[o + (two ? i + nworkers : i)];
        Prod p0, p1;
        item_issue(cls, d0, ya, p0);
        if (two) item_issue(cls, d1, ya, p1);
        item_finish(cls, d0, p0, xrow, false);
        if (two) item_finish(cls, d1, p1, xrow, false);
        if constexpr (decltype(cls)::value == CLS_F) {  // publish the columns: the last term of S[p, j] waits for them
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) {
            __hip_atomic_store(&fflag[d0[0]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (two) __hip_atomic_store(&fflag[d1[0]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    };
    const bool inv = !(dbg & 2);  // timing switches of tools/leaf_bench: dbg 2 = no inverse, dbg 4 = no trailing update
    if ((dbg & 8) && lane == 0) g_leaf_stamps[64 + (p * 8 + wave) * 2] = (long long)__builtin_readcyclecounter();
    int o = 0;
    if (inv) run_list(cF{}, 0, o, leaf_nf(p));
    o += leaf_nf(p);
    if (!(dbg & 4)) run_list(cU{}, 1, o, leaf_nu(p));
    o += leaf_nu(p);
    if (inv) run_list(cG{}, 2, o, leaf_ng(p));
    o += leaf_ng(p);
    if (inv) {
      const int o1 = o, o2 = o + leaf_ngu1(p);
      for (int j = first_of(3); j < p; j += nworkers) {
        Prod pr;
        if (j <= p - 2) {  // the term of k2
          const u4 d = tabp[o1 + j];
          item_issue(cG{}, d, ya, pr);
          item_finish(cG{}, d, pr, xrow, false);
        }
        if (j < k) {
          while (__hip_atomic_load(&fflag[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != p) __builtin_amdgcn_s_sleep(1);
          asm volatile("" ::: "memory");
        }
        const u4 d = tabp[o2 + j];  // the term of k: same result block (LDS accesses of one wave stay in order)
        item_issue(cG{}, d, ya, pr);
        item_finish(cG{}, d, pr, xrow, false);
      }
    }
    if ((dbg & 8) && lane == 0) g_leaf_stamps[65 + (p * 8 + wave) * 2] = (long long)__builtin_readcyclecounter();
  };
  auto blk_bytes = [&](int i, int j) { return (unsigned)(((i * 16) * S + j * 16) * (int)sizeof(T)); };

  typedef T v2 __attribute__((ext_vector_type(2)));
  for (int p = 0; p < 8; ++p) {
    // Elimination of panel p.  Wave 0: the pivot rows and the (up to) 64 rows behind them.  While p < 4 there are 48 - 16p rows
    // beyond that window: the helper (wave 6) runs the SAME elimination on the pivot rows -- bitwise the same values -- with
    // those far rows in its b lanes, so they are ready together with the window.  The 16 lanes behind a wave's last row carry
    // the identity (helper for p < 4, wave 0 from p = 4 on) and end as the columns of Y_p.
    const bool helper = p < 4 && wave == 6;
    if (wave == 0 || helper) {
      if (!(dbg & 1)) {
        const int nrows = helper ? 48 - 16 * p : min(64, 112 - 16 * p);  // rows of the block in this wave's b lanes
        const int brow = 16 * p + 16 + (helper ? 64 : 0) + lane;        // the row of lane < nrows
        const bool has_y = helper || p >= 4;
        const int yr = lane - nrows;                                   // has_y: identity row of lanes nrows .. nrows + 15
        // dbg bit 4 (tests): the helper wave starts ~3 us late -- longer than wave 0's whole elimination.  Results must not
        // change: nothing a helper reads is written by another wave in this phase (tests/test_gpu_dag.py).
        if ((dbg & 16) && helper) {
          __builtin_amdgcn_s_sleep(127);
          asm volatile("" ::: "memory");
        }
        T a[16], b[16];
        {
          // Wave 0 overwrites the pivot rows with L at the end of this phase, and no barrier orders the helper's reads of them
          // before that: the helper takes them from the copy in Sc (written a phase earlier, behind a barrier) -- same values.
          const T* srca = helper ? Sc + m16 * YS : As + (16 * p + m16) * S + p * 16;
          // lanes without a row of the block: a row of the identity, or zeros
          const T* srcb = lane < nrows ? As + brow * S + p * 16 : Id + ((has_y && yr < 16) ? yr : 16) * YS;
#pragma unroll
          for (int j = 0; j < 16; j += 2) {
            const v2 va = *reinterpret_cast<const v2*>(srca + j);
            a[j] = va[0];
            a[j + 1] = va[1];
          }
#pragma unroll
          for (int j = 0; j < 16; j += 2) {
            const v2 vb = *reinterpret_cast<const v2*>(srcb + j);
            b[j] = vb[0];
            b[j + 1] = vb[1];
          }
        }
        LEAF_STAMP(40 + 3 * p);
        LeafPivot<0>::run(a, b);
        LEAF_STAMP(41 + 3 * p);
        if (lane < nrows || (has_y && yr < 16)) {
          // a row of the block -> its row of L; row yr of the identity -> column yr of Y_p = row yr of Yt[p]
          T* dst = lane < nrows ? As + brow * S + p * 16 : Yt + yr * S + p * 16;
#pragma unroll
          for (int j = 0; j < 16; j += 2) *reinterpret_cast<v2*>(dst + j) = v2{b[j], b[j + 1]};
        }
        if (!helper && lane < 16) {
          // the pivot block: L_pp (what lies above its diagonal is never read again)
          T* dst = As + (16 * p + lane) * S + p * 16;
#pragma unroll
          for (int j = 0; j < 16; j += 2) *reinterpret_cast<v2*>(dst + j) = v2{a[j], a[j + 1]};
        }
        LEAF_STAMP(42 + 3 * p);
      }
    } else if (p > 0 && (LEAF_WAVE4_WORKS || wave != 4)) {
      // the other waves meanwhile
      pool_phase(p);
    }
    __syncthreads();
    LEAF_STAMP(1 + 2 * p);
    if (p == 7) break;
    if (dbg & 4) continue;

    // block column p+1 for every remaining block row (window rows from wave 0, far rows from the helper)
    if (p + 1 + wave < 8) {
      const u4 d = {blk_bytes(p + 1 + wave, p), blk_bytes(p + 1, p), blk_bytes(p + 1 + wave, p + 1), 0u};
      Prod pr;
      item_issue(cU{}, d, 0, pr);
      item_finish(cU{}, d, pr, nullptr, wave == 0 && p + 1 < 4);
    }
    __syncthreads();
    LEAF_STAMP(2 + 2 * p);
  }

  // ---------------- tail of the inverse: block row 7 of X ----------------
  if (!(dbg & 2)) {
    if (wave >= 1) {  // X[7,j] = -Y_7 S[7,j]: seven blocks, waves 1-7
      const int j = wave - 1;
      const u4 d = {(unsigned)j, blk_bytes(j, 7), blk_bytes(j, 7), (unsigned)(j * 16 * (int)sizeof(TIO))};
      Prod pr;
      item_issue(cF{}, d, (int)leaf_yt_bytes(7), pr);
      item_finish(cF{}, d, pr, xrow_of(7), false);
    }
    __syncthreads();
  }
  LEAF_STAMP(20);
  // diagonal sub-blocks of X
  for (int c = t; c < 8 * 256; c += 512) {
    const int pblk = c >> 8, r = (c >> 4) & 15, j = c & 15;
    gstore<SC1>(&Xblk[(size_t)(pblk * 16 + r) * ld + pblk * 16 + j], (TIO)Yt[j * S + 16 * pblk + r]);
  }
  // diag(L) (read by the alpha tasks of the same launch) and the positive-definite test: a pivot that was not positive has left
  // NaN on the diagonal from its column on
  if (t < 64) {
    const T dv0 = As[t * S + t], dv1 = As[(t + 64) * S + t + 64];
    gstore<SC1>(&ldiag[blk * NB + t], (TIO)dv0);
    gstore<SC1>(&ldiag[blk * NB + 64 + t], (TIO)dv1);
    const unsigned long long bad0 = __ballot(!(dv0 > T(0))), bad1 = __ballot(!(dv1 > T(0)));
    if (t == 0) pool[15] = (bad0 | bad1) != 0ull ? 1 : 0;  // small_eval_kernel reads it behind its next barrier
    if ((bad0 | bad1) != 0ull && t == 0) {
      const int first = bad0 != 0ull ? (int)__builtin_ctzll(bad0) : 64 + (int)__builtin_ctzll(bad1);
      atomicCAS(info, 0, 1 + blk * NB + (first & ~15));  // reported per 16-column panel, as rounds 1-3 did
    }
  }
  // optional: the factor itself (lower triangle of the block), for the f32 path's refinement of the panel solve
  if (W3) {
    TIO* Lblk = W3 + g0;
    for (int c = t; c < 128 * 128; c += 512) {
      const int r = c >> 7, j = c & 127;
      if (j <= r) gstore<SC1>(&Lblk[(size_t)r * ld + j], (TIO)As[r * S + j]);
    }
  }
}

template <typename T, typename TIO>
__global__ void __launch_bounds__(512, 2) leaf_kernel(TIO* __restrict__ W1, TIO* __restrict__ W2, int ld, int blk,
                                                   TIO* __restrict__ ldiag, int* info, int dbg, TIO* __restrict__ W3) {
  if (*info != 0) return;
  extern __shared__ __align__(16) char smem_raw[];
  leaf_body<T, TIO, false>(W1, W2, ld, blk, ldiag, info, dbg, smem_raw, W3);
}

template <typename T>
void launch_leaf(T* W1, T* W2, int ld, int blk, T* ldiag, int* info, hipStream_t s, int dbg, T* W3) {
  hipLaunchKernelGGL((leaf_kernel<double, T>), dim3(1), dim3(512), LeafGeom<double>::LDS_BYTES, s, W1, W2, ld, blk, ldiag, info, dbg, W3);
}
template void launch_leaf<double>(double*, double*, int, int, double*, int*, hipStream_t, int, double*);
template void launch_leaf<float>(float*, float*, int, int, float*, int*, hipStream_t, int, float*);

// =================================================================================================================
// Kernel-matrix assembly
// =================================================================================================================
// Matern map of matern_kernel.rs:65-80; nu2 = 2*nu in {1,3,5}; r = euclidean distance of length-scaled points.
// nu2 = 0 stands for nu = infinity, the squared-exponential kernel exp(-r^2/2): an extension the reference does not have
// (matern_kernel.rs:79 is unimplemented! for other nu); its oracle is sklearn's RBF (tests/golden/*_rbf.npz).
template <typename T>
__device__ __forceinline__ T matern_map(T r, int nu2) {
#pragma clang fp contract(off)
  if (nu2 == 0) return exp_nonpos(T(-0.5) * r * r);
  if (nu2 == 1) return exp_nonpos(-r);
  if (nu2 == 3) {
    const T k = r * T(1.7320508075688772);
    return (k + T(1)) * exp_nonpos(-k);
  }
  const T k = r * T(2.23606797749979);
  return __builtin_fma(k * k, T(1.0 / 3.0), T(1) + k) * exp_nonpos(-k);
}
// one entry of K from the squared scaled distance (matern_kernel.rs:37-81 x constant_kernel.rs x + noise on the diagonal, lml.rs:44)
template <typename T>
__device__ __forceinline__ T kmat_entry(T dist2, int nu2, T amp, T noise, bool diag) {
#pragma clang fp contract(off)
  const T v = amp * matern_map<T>(sqrt_nonneg(dist2), nu2);  // product_kernel.rs:37 (k1 * k2)
  return diag ? v + noise : v;
}

// the poison pattern run_eval looks for in outputs that were never written (a payload no arithmetic produces)
__device__ __forceinline__ void poison_out(EvalOut* out, int t) {
  const double nan = __longlong_as_double(0x7ff8000000005eedLL);
  if (t == 0) {
    out->lml = nan; out->yalpha = nan; out->logdet = nan;
    out->info = 0; out->n_warn = 0; out->done = 0;
  }
  if (t < MAXP) out->grad[t] = nan;
}
// One workgroup of 256 threads: parameters host -> device, poisoned result block, cleared task-queue control words.
__device__ __forceinline__ void eval_prologue(const EvalParams* __restrict__ P, const EvalPrologue& pro) {
  const int t = threadIdx.x;
  constexpr int PW = (int)(sizeof(EvalParams) / 8);
  static_assert(sizeof(EvalParams) % 8 == 0 && PW <= 256, "EvalParams is copied as 8-byte words by one workgroup");
  if (t < PW) reinterpret_cast<unsigned long long*>(pro.dP)[t] = reinterpret_cast<const unsigned long long*>(P)[t];
  poison_out(pro.out, t);
  int4* c4 = reinterpret_cast<int4*>(pro.ctrl);  // hipMalloc'd, a whole number of 16-byte words (hbegp.cpp: dag_ctrl_bytes)
  for (int i = t; i < pro.ctrl_words / 4; i += 256) c4[i] = int4{0, 0, 0, 0};
}

// ---- what kmat_kernel and gradtrace_kernel share (round 5) --------------------------------------------------------------
// Tiles of 64x64 entries; thread (tx = t&15, ty = t>>4) owns rows ty+16*r, cols 4*tx..4*tx+3.
//  * One workgroup per tile, dealt by the hardware dispatcher (which refills a CU the moment a workgroup leaves it).  Round 5
//    tried persistent workgroups fed through a software queue (one returning atomic on one queue word per tile):
//    2,080 grabs on one word cost more than the tiles (kmat 30 -> 45 us, gradtrace 48 -> 86 us alone, worse with more workgroups
//    per CU; profiles/r05_tail_kernels_queue.txt) -- removed.  Capping the occupancy through a larger LDS request (3 / 4 / 5 / 8
//    workgroups per CU, so that the dispatcher has tiles left to balance the end of the launch with) changed nothing either
//    (kmat 35.0 / 30.5 / 30.5 / 30.6 us, gradtrace 50.6 / 43.7 / 44.3 / 43.8 us): both kernels sit at ~64 % of the fp64 VALU issue
//    rate (kmat: ~75 fp64 instructions per entry, 8.4 M entries = 19 us at 100 %), bound by their sqrt / exp sequences.
//  * The j tile's features sit in the LDS so that a thread's four columns are ONE conflict-free 16-byte read (f32) or two
//    (f64: [k][half][tx][2] -- a 16-lane group then reads 256 contiguous bytes); the plain [k][64] layout made the f64 reads
//    2-way conflicted (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 45 % in kmat, 20 % in gradtrace, profiles/r04_pmc_sq.json).
template <typename T>
__device__ __forceinline__ int xj_index(int k, int col) {  // LDS element index of feature k of column `col` (0..63) of the j tile
  if constexpr (sizeof(T) == 8) return k * 64 + ((col >> 1) & 1) * 32 + (col >> 2) * 2 + (col & 1);
  else return k * 64 + col;
}
template <typename T>
__device__ __forceinline__ void read_xj4(const T* xj, int k, int tx, T (&b)[4]) {  // features k of columns 4 tx .. 4 tx + 3
  if constexpr (sizeof(T) == 8) {
    const d2 lo = *reinterpret_cast<const d2*>(xj + k * 64 + tx * 2), hi = *reinterpret_cast<const d2*>(xj + k * 64 + 32 + tx * 2);
    b[0] = lo[0]; b[1] = lo[1]; b[2] = hi[0]; b[3] = hi[1];
  } else {
    const f4 v = *reinterpret_cast<const f4*>(xj + k * 64 + tx * 4);
    b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
  }
}
// NU2: the Matern order at compile time (no branch per entry); tiles that lie wholly inside the n x n matrix skip the
// per-entry bounds tests.
template <typename T, int NU2>
__global__ void __launch_bounds__(256) kmat_kernel(const T* __restrict__ X, int n, int d, int np,
                                                   const EvalParams* __restrict__ P, T* __restrict__ W,
                                                   const int* info, EvalPrologue pro) {
  if (pro.dP) {
    // first kernel of an evaluation (engine.hpp, EvalPrologue): nothing has failed yet, `info` still holds the previous
    // evaluation's value and is not read; workgroup 0 prepares the device-side blocks for the kernels behind this one
    if (blockIdx.x == 0) eval_prologue(P, pro);
  } else if (*info != 0) {
    return;
  }
  extern __shared__ __align__(16) char smem_raw[];
  T* xi = reinterpret_cast<T*>(smem_raw);  // [d][64] scaled rows of the i tile (feature-major)
  T* xj = xi + (size_t)d * 64;             // [d] x 64 columns of the j tile, xj_index layout
  const int t = threadIdx.x;
  // the parameters once per workgroup: in an evaluation driven through pinned memory P is a host block, and every read of it is
  // an uncached round trip over the host link (per-thread reads made the launch 10 us longer at n=4096)
  __shared__ double sp[MAXP];
  if (t < d + 2) sp[t] = reinterpret_cast<const double*>(P)[t];
  static_assert(offsetof(EvalParams, noise) == 0 && offsetof(EvalParams, amp) == 8 && offsetof(EvalParams, ell) == 16, "EvalParams: noise, amp, ell[]");
  const int tx = t & 15, ty = t >> 4;
  typedef T vec4 __attribute__((ext_vector_type(4)));
  __syncthreads();
  {
    const int tile = blockIdx.x;
    const int li = tri_row(tile), lj = tile - li * (li + 1) / 2;
    const int i0 = li * 64, j0 = lj * 64;
    for (int e = t; e < 64 * d; e += 256) {
      const int row = e / d, k = e - row * d;
      const T ell = (T)sp[2 + k];  // A::from_f (matern_kernel.rs:50)
      const int gi = i0 + row, gj = j0 + row;
      xi[k * 64 + row] = (gi < n) ? X[(size_t)gi * d + k] / ell : T(0);  // matern_kernel.rs:51-60
      xj[xj_index<T>(k, row)] = (gj < n) ? X[(size_t)gj * d + k] / ell : T(0);
    }
    __syncthreads();
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = T(0);
    for (int k = 0; k < d; ++k) {  // cdist accumulation order (matern_kernel.rs:274-278)
      T a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = xi[k * 64 + ty + 16 * r];
      read_xj4<T>(xj, k, tx, b);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const T df = a[r] - b[c];
          acc[r][c] += df * df;
        }
    }
    const T amp = (T)sp[1], noise = (T)sp[0];
    if (i0 + 64 <= n && j0 + 64 <= n) {
      const bool dtile = li == lj;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + ty + 16 * r;
        vec4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = kmat_entry<T>(acc[r][c], NU2, amp, noise, dtile && gi == j0 + tx * 4 + c);
        *reinterpret_cast<vec4*>(W + (size_t)gi * np + j0 + tx * 4) = out;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = i0 + ty + 16 * r;
      vec4 out;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gj = j0 + tx * 4 + c;
        const T v = kmat_entry<T>(acc[r][c], NU2, amp, noise, gi == gj);
        out[c] = (gi < n && gj < n) ? v : ((gi == gj) ? T(1) : T(0));  // identity padding
      }
      *reinterpret_cast<vec4*>(W + (size_t)gi * np + j0 + tx * 4) = out;
    }
  }
}

template <typename T>
void launch_kmat(const T* X, int n, int d, int np, int nu2, const EvalParams* P, T* W, const int* info, hipStream_t s,
                 const EvalPrologue* pro) {
  const int nt = np / 64, ntiles = nt * (nt + 1) / 2;
  const size_t lds = (size_t)2 * d * 64 * sizeof(T);
  const dim3 grid(ntiles), block(256);
  const EvalPrologue pr = pro ? *pro : EvalPrologue{};
  switch (nu2) {
    case 0: hipLaunchKernelGGL((kmat_kernel<T, 0>), grid, block, lds, s, X, n, d, np, P, W, info, pr); break;
    case 1: hipLaunchKernelGGL((kmat_kernel<T, 1>), grid, block, lds, s, X, n, d, np, P, W, info, pr); break;
    case 3: hipLaunchKernelGGL((kmat_kernel<T, 3>), grid, block, lds, s, X, n, d, np, P, W, info, pr); break;
    default: hipLaunchKernelGGL((kmat_kernel<T, 5>), grid, block, lds, s, X, n, d, np, P, W, info, pr); break;
  }
}
template void launch_kmat<double>(const double*, int, int, int, int, const EvalParams*, double*, const int*, hipStream_t, const EvalPrologue*);
template void launch_kmat<float>(const float*, int, int, int, int, const EvalParams*, float*, const int*, hipStream_t, const EvalPrologue*);

// =================================================================================================================
// alpha = K^-1 y through the explicit inverse factor: w = X y, alpha = X^T w;  lml pieces (lml.rs:54-59)
// =================================================================================================================
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// block sum for 256 threads; result valid in thread 0. red: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// What the LAST workgroup of a launch to finish does for the whole launch (the ticket pattern).  The L2s of the eight XCDs
// are not coherent with each other, and an agent-scope release fence writes back the whole L2 of the XCD (measured on config M:
// one per workgroup of the gradient launch doubled that launch, 138 -> 290 us, and slowed the other slots' task-queue
// launches by 10 %: it evicts their working set too).  So, as in the task-queue kernel: the few words a workgroup hands over
// (thread 0 writes them) are stored write-through (publish_f64), thread 0 drains them (s_waitcnt vmcnt(0)) and takes a ticket
// with a relaxed agent-scope atomic; the workgroup that draws the last ticket runs ONE agent-scope acquire (drops its CU's L1
// and the stale lines of its L2) and then sees everybody's words.  The ticket word is left at zero for the next launch.
// Returns true (uniformly) in the last workgroup.
__device__ __forceinline__ void publish_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool last_workgroup(int* ticket) {
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = tk == (int)gridDim.x * (int)gridDim.y - 1;
    if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const bool last = s_last != 0;
  if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return last;
}

// w = X y (X lower triangular, row-major): w_i = sum_{k<=i} X[i][k] y[k]   (first half of alpha = X^T (X y), lml.rs:54)
// Round 5: one wave takes TWO rows, i and np-1-i (together np+1 entries: every wave the same work -- with one wave per row the
// launch ended with its longest rows), reads them with 16-byte loads, four independent loads in flight per lane (round 1-4:
// one 8-byte load per lane per round trip, 92 % of the wave cycles in SQ_WAIT_ANY).  Fixed summation order: lane l adds the
// entries k with (k / VEC) % 64 == l in ascending k into accumulator (k / (64 VEC)) % 4, then ((a0 + a1) + (a2 + a3)), then
// the wave sum.  Entries above the diagonal are skipped by the k <= i test, not read as zeros.
template <typename T>
__global__ void __launch_bounds__(256) trmv_n_kernel(const T* __restrict__ Xinv, int np, int n, const T* __restrict__ y,
                                                     T* __restrict__ w, const int* info) {
  if (*info != 0) return;
  using C = Cfg<T>;
  using vec_t = typename C::vec_t;
  constexpr int VEC = C::VEC, SPAN = 64 * VEC;
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (2 * gw >= np) return;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int row = half == 0 ? gw : np - 1 - gw;
    if (half == 1 && row == gw) break;  // odd np: the middle row once
    const T* xr = Xinv + (size_t)row * np;
    const int len = min(row + 1, n);    // k <= row && k < n
    double acc[4] = {0, 0, 0, 0};
    for (int k0 = lane * VEC; k0 < len; k0 += 4 * SPAN) {
      vec_t xv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * SPAN;
        const bool in = k < len;  // a whole vector lies inside the row's padded storage: np is a multiple of 128 >= VEC
        xv[u] = in ? *reinterpret_cast<const vec_t*>(xr + k) : vec_t{};
        yv[u] = in ? *reinterpret_cast<const vec_t*>(y + k) : vec_t{};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (k0 + u * SPAN + e < len) acc[u] = __builtin_fma((double)xv[u][e], (double)yv[u][e], acc[u]);
    }
    double a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    a = wave_sum(a);
    if (lane == 0) w[row] = (T)a;
  }
}

// partial[chunk][j] = sum_{i in chunk, i>=j} X[i][j] w[i]; chunk = 256 rows.  Round 5: a thread owns VEC adjacent columns
// (16-byte loads: 64 lanes cover 128 columns in f64, 256 in f32), wave q of the four takes the rows i = q (mod 4); eight loads in
// flight per lane.  Fixed order: per column the rows of a wave in ascending order, then the four waves' sums in order.
template <typename T>
__global__ void __launch_bounds__(256) trmv_t_kernel(const T* __restrict__ Xinv, int np, const T* __restrict__ w,
                                                     double* __restrict__ part, const int* info) {
  if (*info != 0) return;
  using C = Cfg<T>;
  using vec_t = typename C::vec_t;
  constexpr int VEC = C::VEC, COLS = 64 * VEC;
  __shared__ double red[4][COLS];
  const int lane = threadIdx.x & 63, sgrp = threadIdx.x >> 6;
  const int col0 = blockIdx.x * COLS, j0 = col0 + lane * VEC, chunk = blockIdx.y;
  const int i_begin = chunk * 256, i_end = min(i_begin + 256, np);
  if (i_end <= col0) return;  // the whole chunk lies above these columns' diagonal entries (uniform)
  double acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0;
  if (j0 < np) {
    int i = i_begin + sgrp;
    for (; i + 28 < i_end; i += 32) {
      vec_t xv[8];
      T wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int iu = i + 4 * u;
        xv[u] = (iu >= j0) ? *reinterpret_cast<const vec_t*>(Xinv + (size_t)iu * np + j0) : vec_t{};
        wv[u] = w[iu];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (i + 4 * u >= j0 + e) acc[e] = __builtin_fma((double)xv[u][e], (double)wv[u], acc[e]);
    }
    for (; i < i_end; i += 4) {
      if (i < j0) continue;
      const vec_t xv = *reinterpret_cast<const vec_t*>(Xinv + (size_t)i * np + j0);
      const double wi = (double)w[i];
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (i >= j0 + e) acc[e] = __builtin_fma((double)xv[e], wi, acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[sgrp][lane * VEC + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < COLS; c += 256)
    if (col0 + c < np) part[(size_t)chunk * np + col0 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// lml = -1/2 y^T alpha - sum log L_ii - n/2 log(2 pi)   (lml.rs:57-59); fixed summation order; one thread
__device__ __forceinline__ void lml_final(const double* __restrict__ sums, int nblocks, int n, EvalOut* out) {
  double s1 = 0, s2 = 0;
  for (int b = 0; b < nblocks; ++b) {
    s1 += sums[2 * b];
    s2 += sums[2 * b + 1];
  }
  out->yalpha = s1;
  out->logdet = s2;
  out->lml = __builtin_fma(-0.5, s1, -s2) - (double)n / 2.0 * log(2.0 * 3.14159265358979323846);
  atomicOr(&out->done, 1);
}

// alpha_j = sum_chunks part[c][j]; per-workgroup partial sums of y^T alpha and sum log L_jj  (256 columns per workgroup)
template <typename T>
__global__ void __launch_bounds__(256) alpha_reduce_kernel(const double* __restrict__ part, int nchunks, int np, int n,
                                                           const T* __restrict__ y, const T* __restrict__ ldiag,
                                                           T* __restrict__ alpha, double* __restrict__ sums, const int* info,
                                                           EvalOut* out_final, int* ticket) {
  if (*info != 0) return;
  __shared__ double red[4];
  const int j = blockIdx.x * 256 + threadIdx.x;
  double ya = 0, ld = 0;
  if (j < np) {
    double a = 0;
    // same sum, chunk by chunk in ascending order; the loads of eight chunks are issued together (as a plain loop each one was
    // waited for before the next: 16 dependent round trips at n = 4096)
    int c = j / 256;
    for (; c + 8 <= nchunks; c += 8) {
      double pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[u] = part[(size_t)(c + u) * np + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += pv[u];
    }
    for (; c < nchunks; ++c) a += part[(size_t)c * np + j];
    const T at = (T)a;
    alpha[j] = (j < n) ? at : T(0);
    if (j < n) {
      ya = (double)y[j] * (double)at;
      ld = log((double)ldiag[j]);
    }
  }
  const double s1 = block_sum(ya, red);
  const double s2 = block_sum(ld, red);
  if (threadIdx.x == 0) {
    publish_f64(&sums[2 * blockIdx.x], s1);
    publish_f64(&sums[2 * blockIdx.x + 1], s2);
  }
  // out_final: the last workgroup to finish forms the lml itself (one launch less on every evaluation's serial tail)
  if (out_final && last_workgroup(ticket) && threadIdx.x == 0) lml_final(sums, (int)gridDim.x, n, out_final);
}

// lml = -1/2 y^T alpha - sum log L_ii - n/2 log(2 pi)   (lml.rs:57-59); fixed summation order
__global__ void lml_final_kernel(const double* __restrict__ sums, int nblocks, int n, EvalOut* out, const int* info) {
  if (*info != 0) return;
  if (threadIdx.x == 0) lml_final(sums, nblocks, n, out);
}

template <typename T>
void launch_alpha_lml(const T* Xinv, int np, int n, const T* y, const T* ldiag, T* wbuf, double* part, T* alpha,
                      EvalOut* out, const int* info, hipStream_t s, int* ticket) {
  hipLaunchKernelGGL((trmv_n_kernel<T>), dim3((np / 2 + 3) / 4), dim3(256), 0, s, Xinv, np, n, y, wbuf, info);
  const int nchunks = np / 256 > 0 ? (np + 255) / 256 : 1;
  constexpr int TCOLS = 64 * Cfg<T>::VEC;
  hipLaunchKernelGGL((trmv_t_kernel<T>), dim3((np + TCOLS - 1) / TCOLS, nchunks), dim3(256), 0, s, Xinv, np, wbuf, part, info);
  // the per-workgroup sums live behind the chunk partials (part has room for nchunks*np + 2*np/256 doubles)
  double* sums = part + (size_t)nchunks * np;
  const int nblocks = (np + 255) / 256;
  hipLaunchKernelGGL((alpha_reduce_kernel<T>), dim3(nblocks), dim3(256), 0, s, part, nchunks, np, n, y, ldiag, alpha, sums, info,
                     ticket ? out : nullptr, ticket);
  if (!ticket) hipLaunchKernelGGL(lml_final_kernel, dim3(1), dim3(64), 0, s, sums, nblocks, n, out, info);
}
template void launch_alpha_lml<double>(const double*, int, int, const double*, const double*, double*, double*, double*,
                                       EvalOut*, const int*, hipStream_t, int*);
template void launch_alpha_lml<float>(const float*, int, int, const float*, const float*, float*, double*, float*,
                                      EvalOut*, const int*, hipStream_t, int*);

// =================================================================================================================
// Gradient of the lml: g_j = 1/2 sum_ik (alpha_i alpha_k - Kinv_ik) dK_ik/dtheta_j   (lml.rs:62-70)
// theta order [noise, amplitude, ell_1..ell_d]; dK formulas: constant_kernel.rs:31-38, product_kernel.rs:56-67,
// matern_kernel.rs:88-131.  One pass over the lower triangle of Kinv (off-diagonal entries weighted twice).
// =================================================================================================================
constexpr int GT_CHUNK = 8;  // length-scale parameters accumulated per register pass

// Sums of NV per-thread values over the 256 threads of a workgroup, all at once: every thread parks its values in the LDS
// ([v][256]), wave q adds up the values v = q, q + 4, ... (lane l: threads l, l + 64, l + 128, l + 192 in that order, then the
// wave sum) and its lane 0 stores the result write-through.  Two barriers for up to GT_CHUNK + 2 values (round 1-4: a block sum
// with two barriers per value -- 20 barriers per tile at d = 8, as much time as the tile's arithmetic).  Fixed order.
template <int NV>
__device__ __forceinline__ void block_sums_publish(const double (&v)[NV], int nv, double* red, double* dst) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __syncthreads();  // the previous use of red has been read
#pragma unroll
  for (int q = 0; q < NV; ++q)
    if (q < nv) red[q * 256 + t] = v[q];
  __syncthreads();
  for (int q = wave; q < nv; q += 4) {
    const double* r = red + q * 256;
    double a = ((r[lane] + r[lane + 64]) + r[lane + 128]) + r[lane + 192];
    a = wave_sum(a);
    if (lane == 0) publish_f64(&dst[q], a);
  }
}

template <typename T, int NU2>
__device__ __forceinline__ void gradtrace_tile(int tile, const T* __restrict__ X, int n, int d, int np, T amp, T noise,
                                               const double* inv_l2, const T* __restrict__ Kinv, const T* __restrict__ alpha,
                                               double* __restrict__ part, T* xi, T* xj, double* red) {
  const int li = tri_row(tile), lj = tile - li * (li + 1) / 2;
  const int i0 = li * 64, j0 = lj * 64;
  const int t = threadIdx.x;
  for (int e = t; e < 64 * d; e += 256) {
    const int row = e / d, k = e - row * d;
    const int gi = i0 + row, gj = j0 + row;
    xi[k * 64 + row] = (gi < n) ? X[(size_t)gi * d + k] : T(0);
    xj[xj_index<T>(k, row)] = (gj < n) ? X[(size_t)gj * d + k] : T(0);
  }
  __syncthreads();
  const int tx = t & 15, ty = t >> 4;

  // pass A: per-element coefficient coef = wgt * W * c * g(r) and the noise / amplitude sums
  T coef[4][4];
  double g_noise = 0, g_amp = 0;
  {
    T dsum[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) dsum[r][c] = T(0);
    for (int k = 0; k < d; ++k) {
      const T il2 = (T)inv_l2[k];
      T a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = xi[k * 64 + ty + 16 * r];
      read_xj4<T>(xj, k, tx, b);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const T df = a[r] - b[c];
          dsum[r][c] += df * df * il2;
        }
    }
    // Matern value km and gradient factor gr of one entry: dK/dlog(ell_k) = c * gr * d_k (matern_kernel.rs:102-131)
    auto km_gr = [&](T ds, T* km, T* gr) {
#pragma clang fp contract(off)
      if (NU2 == 5) {
        const T tt = sqrt_nonneg(ds * T(5));
        const T e = exp_nonpos(-tt);
        *km = __builtin_fma(tt * tt, T(1.0 / 3.0), T(1) + tt) * e;
        *gr = T(5.0 / 3.0) * (tt + T(1)) * e;  // matern_kernel.rs:119-131
      } else if (NU2 == 3) {
        const T tt = sqrt_nonneg(ds * T(3));
        const T e = exp_nonpos(-tt);
        *km = (tt + T(1)) * e;
        *gr = T(3) * e;  // matern_kernel.rs:112-118
      } else if (NU2 == 0) {
        *km = exp_nonpos(T(-0.5) * ds);  // squared exponential: dK/dlog(ell_k) = K * d_k
        *gr = *km;
      } else {
        const T rr = sqrt_nonneg(ds);
        *km = exp_nonpos(-rr);
        *gr = (rr > T(0)) ? *km / rr : T(0);  // matern_kernel.rs:102-111 (non-finite -> 0)
      }
    };
    typedef T vec4 __attribute__((ext_vector_type(4)));
    if (li > lj && i0 + 64 <= n) {
      // a tile strictly below the diagonal and wholly inside the matrix: every entry counts twice, no bounds tests
      const vec4 aj = *reinterpret_cast<const vec4*>(alpha + j0 + tx * 4);
      vec4 kv[4];
      T ai[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // the four rows' loads together
        const int gi = i0 + ty + 16 * r;
        ai[r] = alpha[gi];
        kv[r] = *reinterpret_cast<const vec4*>(Kinv + (size_t)gi * np + j0 + tx * 4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const T w = ai[r] * aj[c] - kv[r][c];  // lml.rs:62 (tmp)
          T km, gr;
          km_gr(dsum[r][c], &km, &gr);
          g_amp += (double)(T(2) * w * (amp * km));  // constant_kernel.rs:31-38 x K_matern
          coef[r][c] = T(2) * w * amp * gr;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + ty + 16 * r;
        const T ai = (gi < n) ? alpha[gi] : T(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gj = j0 + tx * 4 + c;
          T cf = T(0);
          if (gi < n && gj <= gi) {
            const T w = ai * alpha[gj] - Kinv[(size_t)gi * np + gj];  // lml.rs:62 (tmp)
            const T wgt = (gi == gj) ? T(1) : T(2);
            T km, gr;
            km_gr(dsum[r][c], &km, &gr);
            if (gi == gj) g_noise += (double)(w * noise);      // noise gradient = eye * noise (lml.rs:41)
            g_amp += (double)(wgt * w * (amp * km));            // constant_kernel.rs:31-38 x K_matern
            cf = wgt * w * amp * gr;
          }
          coef[r][c] = cf;
        }
      }
    }
  }
  const int p = d + 2;
  double* my = part + (size_t)tile * p;
  // pass B: length-scale gradients, GT_CHUNK parameters at a time; the first chunk carries the noise / amplitude sums along
  for (int kc = 0; kc < d; kc += GT_CHUNK) {
    double vals[GT_CHUNK + 2];
#pragma unroll
    for (int u = 0; u < GT_CHUNK + 2; ++u) vals[u] = 0;
    const int lead = kc == 0 ? 2 : 0;
    if (kc == 0) { vals[0] = g_noise; vals[1] = g_amp; }
#pragma unroll
    for (int u = 0; u < GT_CHUNK; ++u) {
      const int k = kc + u;
      if (k < d) {
        const T il2 = (T)inv_l2[k];
        T a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = xi[k * 64 + ty + 16 * r];
        read_xj4<T>(xj, k, tx, b);
        T sacc = T(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const T df = a[r] - b[c];
            sacc += coef[r][c] * (df * df * il2);
          }
        if (kc == 0) vals[2 + u] = (double)sacc;
        else vals[u] = (double)sacc;
      }
    }
    const int nv = lead + min(GT_CHUNK, d - kc);
    block_sums_publish<GT_CHUNK + 2>(vals, nv, red, my + (kc == 0 ? 0 : 2 + kc));
  }
}

// grad[j] = 0.5 * sum_blocks part[b][j] as finalize_grad_kernel forms it (thread t of 256 adds blocks t, t+256, ...; wave sums;
// the four wave sums in order), computed by ONE wave per parameter: lane l carries the four threads l, l+64, l+128, l+192.
__device__ __forceinline__ void finalize_grad_in_block(const double* __restrict__ part, int nblocks, int p, EvalOut* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < p; j += 4) {
    double acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      for (int b = q * 64 + lane; b < nblocks; b += 256) acc[q] += part[(size_t)b * p + j];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0) publish_f64(&out->grad[j], 0.5 * (acc[0] + acc[1] + acc[2] + acc[3]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached the L2 before the caller's barrier opens
}

// device result block -> pinned host block, then the evaluation's serial number behind a system-scope fence (engine.hpp).
// Called by one whole workgroup (>= 128 threads) whose own writes to `out` are complete behind a barrier.
__device__ __forceinline__ void publish_out_in_block(const EvalOut* out, EvalOut* hout, const EvalParams* __restrict__ P) {
  const int t = threadIdx.x;
  constexpr int OW = (int)(offsetof(EvalOut, seq) / 8);
  static_assert(offsetof(EvalOut, seq) % 8 == 0 && OW <= 128, "EvalOut is copied as 8-byte words by one workgroup");
  if (t < OW) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(out) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    reinterpret_cast<unsigned long long*>(hout)[t] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&hout->seq, P->seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// fin.out != null: the launch also finalises the gradient and publishes the evaluation (its last workgroup does), also when
// the factorisation failed (info != 0: the tiles are skipped, the host still gets its answer).
struct GradFinish {
  EvalOut* out = nullptr;
  EvalOut* hout = nullptr;   // pinned result block (null: finalise only)
  int* ticket = nullptr;
};
template <typename T, int NU2>
__global__ void __launch_bounds__(256) gradtrace_kernel(const T* __restrict__ X, int n, int d, int np,
                                                        const EvalParams* __restrict__ P, const T* __restrict__ Kinv,
                                                        const T* __restrict__ alpha, double* __restrict__ part,
                                                        const int* info, GradFinish fin) {
  extern __shared__ __align__(16) char smem_raw[];
  T* xi = reinterpret_cast<T*>(smem_raw);  // [d][64] raw features of the i tile
  T* xj = xi + (size_t)d * 64;             // j tile, xj_index layout
  __shared__ double red[(GT_CHUNK + 2) * 256];
  __shared__ double inv_l2[MAXD];
  const bool failed = *info != 0;
  const int t = threadIdx.x;
  if (!failed) {
    if (t < d) {
      const T ell = (T)P->ell[t];
      inv_l2[t] = (double)(T(1) / (ell * ell));  // 1/scales_k_square (matern_kernel.rs:94-98)
    }
    const T amp = (T)P->amp, noise = (T)P->noise;
    gradtrace_tile<T, NU2>((int)blockIdx.x, X, n, d, np, amp, noise, inv_l2, Kinv, alpha, part, xi, xj, red);
  }
  if (!fin.out) return;
  // every publishing lane (lane 0 of each wave) has its stores in the L2 before thread 0 takes the launch's ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!last_workgroup(fin.ticket)) return;
  if (!failed) {
    finalize_grad_in_block(part, (int)gridDim.x, d + 2, fin.out);
    if (threadIdx.x == 0) atomicOr(&fin.out->done, 2);
  }
  __syncthreads();
  if (fin.hout) publish_out_in_block(fin.out, fin.hout, P);
}

// grad[j] = 0.5 * sum_blocks part[b][j]  (fixed summation order -> bitwise reproducible)
__global__ void __launch_bounds__(256) finalize_grad_kernel(const double* __restrict__ part, int nblocks, int p,
                                                            EvalOut* out, const int* info) {
  if (*info != 0) return;
  __shared__ double red[4];
  const int j = blockIdx.x;
  double acc = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) acc += part[(size_t)b * p + j];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) {
    out->grad[j] = 0.5 * s;
    if (j == p - 1) atomicOr(&out->done, 2);  // the last block alone cannot vouch for the others; the host checks every entry for NaN
  }
}

size_t gradtrace_part_elems(int np, int d) {
  const int nt = np / 64;
  return (size_t)(nt * (nt + 1) / 2) * (d + 2);
}

template <typename T>
void launch_gradtrace(const T* X, int n, int d, int np, int nu2, const EvalParams* P, const T* Kinv, const T* alpha,
                      double* part, EvalOut* out, const int* info, hipStream_t s, int* ticket, EvalOut* hout) {
  const int nt = np / 64;
  const int ntiles = nt * (nt + 1) / 2;
  const size_t lds = (size_t)2 * d * 64 * sizeof(T);
  const dim3 grid(ntiles), block(256);
  GradFinish fin;
  if (ticket) { fin.out = out; fin.hout = hout; fin.ticket = ticket; }
  switch (nu2) {
    case 0: hipLaunchKernelGGL((gradtrace_kernel<T, 0>), grid, block, lds, s, X, n, d, np, P, Kinv, alpha, part, info, fin); break;
    case 1: hipLaunchKernelGGL((gradtrace_kernel<T, 1>), grid, block, lds, s, X, n, d, np, P, Kinv, alpha, part, info, fin); break;
    case 3: hipLaunchKernelGGL((gradtrace_kernel<T, 3>), grid, block, lds, s, X, n, d, np, P, Kinv, alpha, part, info, fin); break;
    default: hipLaunchKernelGGL((gradtrace_kernel<T, 5>), grid, block, lds, s, X, n, d, np, P, Kinv, alpha, part, info, fin); break;
  }
  if (!ticket) hipLaunchKernelGGL(finalize_grad_kernel, dim3(d + 2), dim3(256), 0, s, part, ntiles, d + 2, out, info);
}
template void launch_gradtrace<double>(const double*, int, int, int, int, const EvalParams*, const double*, const double*,
                                       double*, EvalOut*, const int*, hipStream_t, int*, EvalOut*);
template void launch_gradtrace<float>(const float*, int, int, int, int, const EvalParams*, const float*, const float*,
                                      double*, EvalOut*, const int*, hipStream_t, int*, EvalOut*);

// =================================================================================================================
// symmetrize: mirror the lower triangle into the upper one (what invc() hands back, lml.rs:62)
// =================================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) symmetrize_kernel(T* __restrict__ A, int np) {
  __shared__ T tile[64][65];
  const int li = tri_row(blockIdx.x), lj = blockIdx.x - li * (li + 1) / 2;
  const int i0 = li * 64, j0 = lj * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) tile[r][tx] = A[(size_t)(i0 + r) * np + j0 + tx];
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    // element (j0+r, i0+tx) = A[i0+tx][j0+r]
    if (li != lj || tx > r) A[(size_t)(j0 + r) * np + i0 + tx] = tile[tx][r];
  }
}
template <typename T>
void launch_symmetrize(T* A, int np, hipStream_t s) {
  const int nt = np / 64;
  hipLaunchKernelGGL((symmetrize_kernel<T>), dim3(nt * (nt + 1) / 2), dim3(256), 0, s, A, np);
}
template void launch_symmetrize<double>(double*, int, hipStream_t);
template void launch_symmetrize<float>(float*, int, hipStream_t);

// flag |= 1 when a[0..count) and b[0..count) differ anywhere (bitwise comparison of the stored values)
template <typename T>
__global__ void __launch_bounds__(256) prefix_differs_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t count,
                                                             int* __restrict__ flag) {
  bool diff = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) diff |= !(a[i] == b[i]);
  if (diff) atomicOr(flag, 1);
}
template <typename T>
void launch_prefix_differs(const T* a, const T* b, size_t count, int* flag, hipStream_t s) {
  if (count == 0) return;
  const int blocks = (int)std::min<size_t>(1024, (count + 255) / 256);
  hipLaunchKernelGGL((prefix_differs_kernel<T>), dim3(blocks), dim3(256), 0, s, a, b, count, flag);
}
template void launch_prefix_differs<double>(const double*, const double*, size_t, int*, hipStream_t);
template void launch_prefix_differs<float>(const float*, const float*, size_t, int*, hipStream_t);

// =================================================================================================================
// predict (predict.rs:7-52)
// =================================================================================================================
// Kstar[mp][np]: row = candidate.  64x64 tiles as in kmat, no noise, zero padding.
template <typename T>
__global__ void __launch_bounds__(256) kstar_kernel(const T* __restrict__ Xs, int m, const T* __restrict__ X, int n, int d,
                                                    int np, int nu2, const EvalParams* __restrict__ P, T* __restrict__ Ks) {
  extern __shared__ __align__(16) char smem_raw[];
  T* xi = reinterpret_cast<T*>(smem_raw);
  T* xj = xi + (size_t)d * 64;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  for (int e = t; e < 64 * d; e += 256) {
    const int row = e / d, k = e - row * d;
    const T ell = (T)P->ell[k];
    const int gi = i0 + row, gj = j0 + row;
    xi[k * 64 + row] = (gi < m) ? Xs[(size_t)gi * d + k] / ell : T(0);
    xj[k * 64 + row] = (gj < n) ? X[(size_t)gj * d + k] / ell : T(0);
  }
  __syncthreads();
  const int tx = t & 15, ty = t >> 4;
  T acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = T(0);
  for (int k = 0; k < d; ++k) {
    T a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = xi[k * 64 + ty + 16 * r];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = xj[k * 64 + tx * 4 + c];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T df = a[r] - b[c];
        acc[r][c] += df * df;
      }
  }
  const T amp = (T)P->amp;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = i0 + ty + 16 * r;
    T* p = Ks + (size_t)gi * np + j0 + tx * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gj = j0 + tx * 4 + c;
      p[c] = (gi < m && gj < n) ? kmat_entry<T>(acc[r][c], nu2, amp, T(0), false) : T(0);
    }
  }
}
template <typename T>
void launch_kstar(const T* Xs, int m, int mp, const T* X, int n, int d, int np, int nu2, const EvalParams* P, T* Ks,
                  hipStream_t s) {
  const size_t lds = (size_t)2 * d * 64 * sizeof(T);
  hipLaunchKernelGGL((kstar_kernel<T>), dim3(np / 64, mp / 64), dim3(256), lds, s, Xs, m, X, n, d, np, nu2, P, Ks);
}
template void launch_kstar<double>(const double*, int, int, const double*, int, int, int, int, const EvalParams*, double*,
                                   hipStream_t);
template void launch_kstar<float>(const float*, int, int, const float*, int, int, int, int, const EvalParams*, float*,
                                  hipStream_t);

// mean_k = sum_j Kstar[k][j] alpha[j]  (predict.rs:19); one wave per candidate
template <typename T>
__global__ void __launch_bounds__(256) pred_mean_kernel(const T* __restrict__ Ks, int m, int np, const T* __restrict__ alpha,
                                                        T* __restrict__ mean) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  double acc = 0;
  const T* kr = Ks + (size_t)row * np;
  for (int j = lane; j < np; j += 64) acc += (double)kr[j] * (double)alpha[j];
  acc = wave_sum(acc);
  if (lane == 0) mean[row] = (T)acc;
}
template <typename T>
void launch_pred_mean(const T* Ks, int m, int np, const T* alpha, T* mean, hipStream_t s) {
  hipLaunchKernelGGL((pred_mean_kernel<T>), dim3((m + 3) / 4), dim3(256), 0, s, Ks, m, np, alpha, mean);
}
template void launch_pred_mean<double>(const double*, int, int, const double*, double*, hipStream_t);
template void launch_pred_mean<float>(const float*, int, int, const float*, float*, hipStream_t);

// var_k = c + 1e-5 - sum_j Q[k][j] Kstar[k][j], clamp negatives, count those below -sqrt(1e-5)  (predict.rs:25-48)
template <typename T>
__global__ void __launch_bounds__(256) pred_var_kernel(const T* __restrict__ Ks, const T* __restrict__ Q, int m, int np,
                                                       const EvalParams* __restrict__ P, T* __restrict__ var, EvalOut* out) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  double acc = 0;
  const T* kr = Ks + (size_t)row * np;
  const T* qr = Q + (size_t)row * np;
  for (int j = lane; j < np; j += 64) acc += (double)kr[j] * (double)qr[j];
  acc = wave_sum(acc);
  if (lane == 0) {
    const T min_noise = (T)1e-5;
    T v = (T)P->amp + min_noise - (T)acc;
    if (v < -sqrt(min_noise)) atomicAdd(&out->n_warn, 1);
    if (v < T(0)) v = T(0);
    var[row] = v;
  }
}
template <typename T>
void launch_pred_var(const T* Ks, const T* Q, int m, int np, const EvalParams* P, T* var, EvalOut* out, hipStream_t s) {
  hipLaunchKernelGGL((pred_var_kernel<T>), dim3((m + 3) / 4), dim3(256), 0, s, Ks, Q, m, np, P, var, out);
}
template void launch_pred_var<double>(const double*, const double*, int, int, const EvalParams*, double*, EvalOut*, hipStream_t);
template void launch_pred_var<float>(const float*, const float*, int, int, const EvalParams*, float*, EvalOut*, hipStream_t);

// =================================================================================================================
// predict for a handful of candidates (m <= PRED_SMALL_MAX): the caller's acquisition and selection loops issue
// thousands of single-point predicts per generation (acquisition.rs:46-64, minimize.rs:656-714).  The batched path pads
// to 128 candidate rows and runs a tile GEMM over all of L^-1 (0.2 ms at n=4096 whatever m is); here L^-1 is read once,
// row by row, against the m cross-kernel vectors:
//   kstar_small:  Ks[q][j] = c * Matern(x*_q, x_j)  (same arithmetic as kstar_kernel), partial sums of Ks[q][j] alpha_j
//   rowdot_small: w[i][q] = sum_{j<=i} Linv[i][j] Ks[q][j]          one wave per row, fp64 accumulation
//   finish_small: mean_q = sum partials; var_q = c + 1e-5 - sum_i w[i][q]^2, clamp, warning count   (predict.rs:18-48)
// All sums run in a fixed order: results are bitwise reproducible.
// =================================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) kstar_small_kernel(const T* __restrict__ Xs, int m, const T* __restrict__ X, int n, int d, int np,
                                                          int nu2, const EvalParams* __restrict__ P, const T* __restrict__ alpha,
                                                          T* __restrict__ Ks, double* __restrict__ pmean) {
  __shared__ double red[4];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const T amp = (T)P->amp;
  for (int q = 0; q < m; ++q) {
    T v = T(0);
    if (j < n) {
      T acc = T(0);
      for (int k = 0; k < d; ++k) {  // cdist accumulation order (matern_kernel.rs:274-278), operands scaled as matern_kernel.rs:50-60
        const T ell = (T)P->ell[k];
        const T df = Xs[(size_t)q * d + k] / ell - X[(size_t)j * d + k] / ell;
        acc += df * df;
      }
      v = kmat_entry<T>(acc, nu2, amp, T(0), false);
    }
    if (j < np) Ks[(size_t)q * np + j] = v;
    const double part = block_sum((j < n) ? (double)v * (double)alpha[j] : 0.0, red);
    if (threadIdx.x == 0) pmean[(size_t)blockIdx.x * PRED_SMALL_MAX + q] = part;
  }
}

template <typename T, int MQ>
__global__ void __launch_bounds__(256) rowdot_small_kernel(const T* __restrict__ Linv, int np, int n, const T* __restrict__ Ks,
                                                           double* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int row = n - 1 - (blockIdx.x * 4 + (threadIdx.x >> 6));  // longest rows first
  if (row < 0) return;
  double acc[MQ];
#pragma unroll
  for (int q = 0; q < MQ; ++q) acc[q] = 0.0;
  const T* xr = Linv + (size_t)row * np;
  for (int k = lane; k <= row; k += 64) {
    const double x = (double)xr[k];
#pragma unroll
    for (int q = 0; q < MQ; ++q) acc[q] += x * (double)Ks[(size_t)q * np + k];
  }
#pragma unroll
  for (int q = 0; q < MQ; ++q) {
    const double s = wave_sum(acc[q]);
    if (lane == 0) w[(size_t)row * PRED_SMALL_MAX + q] = s;
  }
}

// one workgroup finishes all candidates: every thread sums a strided share of the rows, fixed-order block reduction
template <typename T>
__global__ void __launch_bounds__(256) finish_small_kernel(const double* __restrict__ pmean, int nblk, const double* __restrict__ w, int n,
                                                           int m, const EvalParams* __restrict__ P, int want_var, T* __restrict__ mean,
                                                           T* __restrict__ var, int* __restrict__ n_warn) {
  __shared__ double red[4];
  const int t = threadIdx.x;
  for (int q = 0; q < m; ++q) {
    double mu = 0.0;
    for (int b = t; b < nblk; b += 256) mu += pmean[(size_t)b * PRED_SMALL_MAX + q];
    const double mu_all = block_sum(mu, red);
    double ss = 0.0;
    if (want_var)
      for (int i = t; i < n; i += 256) {
        const double v = w[(size_t)i * PRED_SMALL_MAX + q];
        ss += v * v;
      }
    const double ss_all = want_var ? block_sum(ss, red) : 0.0;
    if (t == 0) {
      mean[q] = (T)mu_all;
      if (want_var) {
        const T min_noise = (T)1e-5;
        T v = (T)P->amp + min_noise - (T)ss_all;  // predict.rs:25-37
        if (v < -sqrt(min_noise)) atomicAdd(n_warn, 1);
        if (v < T(0)) v = T(0);
        var[q] = v;
      }
    }
  }
}

template <typename T>
void launch_predict_small(const T* Xs, int m, const T* X, int n, int d, int np, int nu2, const EvalParams* P, const T* alpha, const T* Linv,
                          T* Ks, double* pmean, double* w, int want_var, T* mean, T* var, int* n_warn, hipStream_t s) {
  const int nblk = (np + 255) / 256;
  hipLaunchKernelGGL((kstar_small_kernel<T>), dim3(nblk), dim3(256), 0, s, Xs, m, X, n, d, np, nu2, P, alpha, Ks, pmean);
  if (want_var) {
    const dim3 grid((n + 3) / 4), block(256);
    if (m <= 1) hipLaunchKernelGGL((rowdot_small_kernel<T, 1>), grid, block, 0, s, Linv, np, n, Ks, w);
    else if (m <= 2) hipLaunchKernelGGL((rowdot_small_kernel<T, 2>), grid, block, 0, s, Linv, np, n, Ks, w);
    else if (m <= 4) hipLaunchKernelGGL((rowdot_small_kernel<T, 4>), grid, block, 0, s, Linv, np, n, Ks, w);
    else if (m <= 8) hipLaunchKernelGGL((rowdot_small_kernel<T, 8>), grid, block, 0, s, Linv, np, n, Ks, w);
    else hipLaunchKernelGGL((rowdot_small_kernel<T, 16>), grid, block, 0, s, Linv, np, n, Ks, w);
  }
  hipLaunchKernelGGL((finish_small_kernel<T>), dim3(1), dim3(256), 0, s, pmean, nblk, w, n, m, P, want_var, mean, var, n_warn);
}
template void launch_predict_small<double>(const double*, int, const double*, int, int, int, int, const EvalParams*, const double*, const double*,
                                           double*, double*, double*, int, double*, double*, int*, hipStream_t);
template void launch_predict_small<float>(const float*, int, const float*, int, int, int, int, const EvalParams*, const float*, const float*,
                                          float*, double*, double*, int, float*, float*, int*, hipStream_t);

// =================================================================================================================
// One evaluation in ONE launch for problems of at most 128 rows (np = 128, d <= 32): the reference's own regime
// (minimize.rs:118-120 keeps n at 100-200).  The five launches of the general path (kmat, diagonal block, trmv x2 + reductions,
// K^-1 = X^T X, gradient) cost ~2 us of launch gap and one HBM round trip each -- more than their arithmetic at this size.
// Here one workgroup keeps everything in the LDS:
//   K = c Matern + s2 I      into the block image (the arithmetic of kmat_kernel, entry for entry; K never goes to HBM)
//   L, X = L^-1              leaf_body on that image (X and diag(L) also go to HBM: the model keeps them)
//   alpha = X^T (X y), lml   fp64 dot products against the image (X lives transposed in its strict upper blocks)
//   K^-1 = X^T X             120 16x16x16 products on MFMA, into the lower blocks of the image (L is no longer needed) and to HBM
//   g_j = 1/2 sum (alpha alpha^T - K^-1) o dK_j    the arithmetic of gradtrace_kernel, K^-1 and alpha read from the LDS
// =================================================================================================================
template <typename TIO, int NU2>
__device__ __forceinline__ void small_eval_body(const SmallEval& g, char* smem_raw) {
  using T = double;
  using C = Cfg<T>;
  using L = LeafGeom<T>;
  using acc_t = typename C::acc_t;
  constexpr int S = L::S, YS = L::YS, YB = L::YB;
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Yt = As + 128 * S;
  T* Sc = Yt + 16 * S;       // the diagonal block's scratch region (pivot-block copy, identity, flags, item tables): 1280 doubles,
                             // and 480 more to the end of the CU's LDS.  Outside leaf_body it is this kernel's scratch:
  T* yv = Sc;                //   [128] y
  T* wv = Sc + 128;          //   [128] w = X y
  T* av = Sc + 256;          //   [128] alpha (rounded to the element type, as the general path stores it)
  T* red = Sc + 384;         //   [8 x 35] per-wave partial sums
  T* ellv = Sc + 664;        //   [32] length scales, then [32] their inverse squares (in the element type's arithmetic)
  T* feat = Sc + 728;        //   [8][128] a chunk of eight features of all rows, [feature][row]; 8 doubles to the end of the CU's LDS stay free
                             //   (small_fit_kernel keeps two flags there)
  static_assert(728 + 8 * 128 + 8 <= (163840 - (128 * S + 16 * S) * 8) / 8, "the scratch of small_eval_kernel fits behind the block image");
  int* pool = reinterpret_cast<int*>(Sc + YB + 17 * YS + 2);  // leaf_body's flags (pool[15]: not positive definite)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m16 = lane & 15, q4 = lane >> 4;
  const int n = g.n, d = g.d;
  const TIO* X = static_cast<const TIO*>(g.X);
  const EvalParams* P = g.P;
  // the parameters are read with agent-scope loads: inside small_fit_kernel they change between two evaluations of ONE launch,
  // and a scalar load (what a uniform address compiles to) goes through the scalar cache, which no vector store updates
  auto pld = [](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const TIO amp = (TIO)pld(&P->amp), noise = (TIO)pld(&P->noise);
  if (t == 0) g_leaf_stamps[200] = (long long)__builtin_readcyclecounter();
  // start of an evaluation (what launch_reset_out does for the general path): the outputs are poisoned, the flags cleared
  EvalOut* out = g.hout;
  {
    const double poison = __longlong_as_double(0x7ff8000000005eedLL);
    if (t < MAXP) out->grad[t] = poison;
    if (t == 0) {
      out->lml = poison; out->yalpha = poison; out->logdet = poison;
      out->info = 0; out->n_warn = 0; out->done = 0;
      g.out->info = 0;
    }
  }
  // thread (tx, ty, half) owns, in each of the three lower 64x64 tiles (0,0), (1,0), (1,1): rows ty + 16 (2 half + r), r = 0, 1,
  // columns 4 tx .. 4 tx + 3 -- per entry the sums of kmat_kernel / gradtrace_kernel in their order
  const int tx = t & 15, ty = (t >> 4) & 15, half = t >> 8;
  auto tile_i0 = [](int tile) { return tile == 0 ? 0 : 64; };
  auto tile_j0 = [](int tile) { return tile == 2 ? 64 : 0; };
  // eight features of all 128 rows -> feat[u][row]; scaled: divided by the length scale (kmat), else raw (gradient)
  auto stage_features = [&](int kc, bool scaled) {
    for (int e = t; e < 8 * 128; e += 512) {
      const int row = e >> 3, u = e & 7, k = kc + u;
      TIO v = TIO(0);
      if (row < n && k < d) {
        v = X[(size_t)row * d + k];
        if (scaled) v = v / (TIO)ellv[k];  // matern_kernel.rs:51-60
      }
      feat[u * 128 + row] = (T)v;
    }
  };
  if (t < 32) {
    const TIO ell = t < d ? (TIO)pld(&P->ell[t]) : TIO(1);  // A::from_f (matern_kernel.rs:50)
    ellv[t] = (T)ell;
    ellv[32 + t] = (T)(TIO(1) / (ell * ell));          // 1/scales_k_square (matern_kernel.rs:94-98)
  }
  __syncthreads();

  // ---- K into the image (never to HBM): squared scaled distances accumulated feature by feature, eight features per staging
  {
    TIO acc[3][2][4];
#pragma unroll
    for (int tile = 0; tile < 3; ++tile)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[tile][r][c] = TIO(0);
    for (int kc = 0; kc < d; kc += 8) {
      stage_features(kc, true);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kc + u < d) {  // uniform; cdist accumulation order (matern_kernel.rs:274-278)
#pragma unroll
          for (int tile = 0; tile < 3; ++tile) {
            TIO a[2], b[4];
#pragma unroll
            for (int r = 0; r < 2; ++r) a[r] = (TIO)feat[u * 128 + tile_i0(tile) + ty + 16 * (2 * half + r)];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = (TIO)feat[u * 128 + tile_j0(tile) + tx * 4 + c];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const TIO df = a[r] - b[c];
                acc[tile][r][c] += df * df;
              }
          }
        }
      }
      __syncthreads();
    }
    // zeros everywhere first (the strict upper 16x16 blocks must start as zeros), then the three tiles' entries
    for (int e = t; e < 128 * (S / 2); e += 512) reinterpret_cast<d2*>(As)[e] = d2{0, 0};
    __syncthreads();
#pragma unroll
    for (int tile = 0; tile < 3; ++tile)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gi = tile_i0(tile) + ty + 16 * (2 * half + r);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gj = tile_j0(tile) + tx * 4 + c;
          if ((gj >> 4) <= (gi >> 4)) {  // lower 16x16 blocks only
            const TIO v = kmat_entry<TIO>(acc[tile][r][c], NU2, amp, noise, gi == gj);
            As[gi * S + gj] = (T)((gi < n && gj < n) ? v : ((gi == gj) ? TIO(1) : TIO(0)));  // identity padding
          }
        }
      }
    __syncthreads();
  }
  if (t == 0) g_leaf_stamps[201] = (long long)__builtin_readcyclecounter();

  // ---- L and X = L^-1
  TIO* W2 = static_cast<TIO*>(g.W2);
  leaf_body<T, TIO, false, true>(nullptr, W2, 128, 0, static_cast<TIO*>(g.ldiag), &g.out->info, 0, smem_raw, nullptr);
  __syncthreads();
  if (t == 0) g_leaf_stamps[202] = (long long)__builtin_readcyclecounter();
  if (pool[15] != 0) {  // not positive definite: the outputs stay poisoned (lml.rs:47-50)
    if (t == 0) out->info = g.out->info;
    return;
  }
  if (!(g.mode & 1)) return;
  __syncthreads();  // every thread has read the flag: the scratch may be reused

  // X[i][k] (i >= k) in the image: block (I, K), I > K, lives transposed in block (K, I); the diagonal blocks in Yt (transposed)
  auto xel = [&](int i, int k) -> T {
    const int I = i >> 4, K = k >> 4;
    return I > K ? As[(K * 16 + (k & 15)) * S + I * 16 + (i & 15)] : Yt[(k & 15) * S + 16 * I + (i & 15)];
  };
  if (t < 128) yv[t] = t < n ? (T)static_cast<const TIO*>(g.y)[t] : T(0);
  if (t >= 128 && t < 160) {  // the length scales again (leaf_body's scratch went over them)
    const int k = t - 128;
    const TIO ell = k < d ? (TIO)pld(&P->ell[k]) : TIO(1);
    ellv[k] = (T)ell;
    ellv[32 + k] = (T)(TIO(1) / (ell * ell));
  }
  __syncthreads();
  {
    // w_i = sum_{k <= i} X[i][k] y[k]: four threads per row, fp64, four independent partial sums per thread
    const int i = t >> 2, sub = t & 3;
    T acc[4] = {0, 0, 0, 0};
    int k = sub;
    for (; k + 12 <= i; k += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_fma(xel(i, k + 4 * u), yv[k + 4 * u], acc[u]);
    }
    for (; k <= i; k += 4) acc[0] = __builtin_fma(xel(i, k), yv[k], acc[0]);
    T a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (sub == 0) wv[i] = (T)(TIO)a;  // the general path keeps w in the element type
  }
  __syncthreads();
  {
    // alpha_j = sum_{i >= j} X[i][j] w_i
    const int j = t >> 2, sub = t & 3;
    T acc[4] = {0, 0, 0, 0};
    int i = j + sub;
    for (; i + 12 < 128; i += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_fma(xel(i + 4 * u, j), wv[i + 4 * u], acc[u]);
    }
    for (; i < 128; i += 4) acc[0] = __builtin_fma(xel(i, j), wv[i], acc[0]);
    T a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (sub == 0) {
      const TIO at = j < n ? (TIO)a : TIO(0);
      static_cast<TIO*>(g.alpha)[j] = at;
      av[j] = (T)at;
    }
  }
  __syncthreads();
  {
    // lml = -1/2 y^T alpha - sum log L_ii - n/2 log 2 pi (lml.rs:57-59)
    T ya = 0, ld = 0;
    if (t < n) {
      ya = yv[t] * av[t];
      ld = log((T)(TIO)As[t * S + t]);
    }
    ya = wave_sum(ya);
    ld = wave_sum(ld);
    if (lane == 0) { red[wave * 2] = ya; red[wave * 2 + 1] = ld; }
    __syncthreads();
    if (t == 0) {
      T s1 = 0, s2 = 0;
      for (int w = 0; w < 8; ++w) { s1 += red[2 * w]; s2 += red[2 * w + 1]; }
      out->yalpha = s1;
      out->logdet = s2;
      out->lml = __builtin_fma(-0.5, s1, -s2) - (double)n / 2.0 * log(2.0 * 3.14159265358979323846);
      out->done = 1;  // one writer
    }
  }
  if (t == 0) g_leaf_stamps[203] = (long long)__builtin_readcyclecounter();
  if (!(g.mode & 2)) return;

  // ---- K^-1 = X^T X, lower blocks (I, J): sum_{K >= I} X[K,I]^T X[K,J].  Both operands are rows of the image in the layout an
  // MFMA fragment wants (X[K,I]^T = block (I,K) of the image, or Yt_I for K = I).  36 blocks, 120 products, dealt to the 8 waves.
  {
    const int P0 = (m16 * S + q4) * (int)sizeof(T), P1 = (q4 * S + m16) * (int)sizeof(T);
    char* lds0 = reinterpret_cast<char*>(As);
    auto ldsT = [&](int byte_off) -> T& { return *reinterpret_cast<T*>(lds0 + byte_off); };
    TIO* Kinv = static_cast<TIO*>(g.Kinv);
    int q = 0;
    for (int I = 0; I < 8; ++I)
      for (int J = 0; J <= I; ++J, ++q) {
        if ((q & 7) != wave) continue;
        acc_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        for (int K = I; K < 8; ++K) {
          const int pa = (K == I ? (int)leaf_yt_bytes(I) : (int)leaf_blk_bytes(I, K)) + P0;
          const int pb = (K == J ? (int)leaf_yt_bytes(J) : (int)leaf_blk_bytes(J, K)) + P0;
          T af[4], bf[4];
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            af[k4] = ldsT(pa + k4 * 32);
            bf[k4] = ldsT(pb + k4 * 32);
          }
          a0 = C::mfma(af[0], bf[0], a0);
          a1 = C::mfma(af[2], bf[2], a1);
          a0 = C::mfma(af[1], bf[1], a0);
          a1 = C::mfma(af[3], bf[3], a1);
        }
        const int pc = (int)leaf_blk_bytes(I, J) + P1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T v = a0[r] + a1[r];
          ldsT(pc + r * 4 * S * (int)sizeof(T)) = v;  // over L: nobody reads the factor any more
          Kinv[(size_t)(I * 16 + q4 + 4 * r) * 128 + J * 16 + m16] = (TIO)v;
        }
      }
  }
  if (!(g.mode & 4)) return;
  __syncthreads();
  if (t == 0) g_leaf_stamps[204] = (long long)__builtin_readcyclecounter();

  // ---- gradient (lml.rs:62-70): the arithmetic of gradtrace_kernel on the three lower 64x64 tiles; K^-1 and alpha come from the
  // LDS, the raw features are staged eight at a time
  {
    const int p = d + 2;
    double gsum[2] = {0, 0};          // noise, amplitude
    TIO coef[3][2][4], dsum[3][2][4];
#pragma unroll
    for (int tile = 0; tile < 3; ++tile)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dsum[tile][r][c] = TIO(0);
    // pass A: scaled squared distances (matern_kernel.rs:94-98 form), then per entry the coefficient wgt * W * c * g(r)
    for (int kc = 0; kc < d; kc += 8) {
      stage_features(kc, false);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kc + u < d) {
          const TIO il2 = (TIO)ellv[32 + kc + u];
#pragma unroll
          for (int tile = 0; tile < 3; ++tile) {
            TIO a[2], b[4];
#pragma unroll
            for (int r = 0; r < 2; ++r) a[r] = (TIO)feat[u * 128 + tile_i0(tile) + ty + 16 * (2 * half + r)];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = (TIO)feat[u * 128 + tile_j0(tile) + tx * 4 + c];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const TIO df = a[r] - b[c];
                dsum[tile][r][c] += df * df * il2;
              }
          }
        }
      }
      if (kc + 8 < d) __syncthreads();  // the last chunk stays staged: pass B starts with it when d <= 8
    }
#pragma unroll
    for (int tile = 0; tile < 3; ++tile)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gi = tile_i0(tile) + ty + 16 * (2 * half + r);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gj = tile_j0(tile) + tx * 4 + c;
          TIO cf = TIO(0);
          if (gi < n && gj <= gi) {
#pragma clang fp contract(off)
            const TIO w = (TIO)av[gi] * (TIO)av[gj] - (TIO)As[gi * S + gj];  // lml.rs:62 (tmp)
            const TIO wgt = (gi == gj) ? TIO(1) : TIO(2);
            TIO km, gr;
            const TIO ds = dsum[tile][r][c];
            if (NU2 == 5) {
              const TIO tt = sqrt_nonneg(ds * TIO(5));
              const TIO e = exp_nonpos(-tt);
              km = __builtin_fma(tt * tt, TIO(1.0 / 3.0), TIO(1) + tt) * e;
              gr = TIO(5.0 / 3.0) * (tt + TIO(1)) * e;  // matern_kernel.rs:119-131
            } else if (NU2 == 3) {
              const TIO tt = sqrt_nonneg(ds * TIO(3));
              const TIO e = exp_nonpos(-tt);
              km = (tt + TIO(1)) * e;
              gr = TIO(3) * e;  // matern_kernel.rs:112-118
            } else if (NU2 == 0) {
              km = exp_nonpos(TIO(-0.5) * ds);
              gr = km;
            } else {
              const TIO rr = sqrt_nonneg(ds);
              km = exp_nonpos(-rr);
              gr = (rr > TIO(0)) ? km / rr : TIO(0);  // matern_kernel.rs:102-111
            }
            if (gi == gj) gsum[0] += (double)(w * noise);
            gsum[1] += (double)(wgt * w * (amp * km));
            cf = wgt * w * amp * gr;
          }
          coef[tile][r][c] = cf;
        }
      }
    {
      const double s0 = wave_sum(gsum[0]), s1 = wave_sum(gsum[1]);
      if (lane == 0) { red[wave * 35] = s0; red[wave * 35 + 1] = s1; }  // (the lml's sums in red were read before the K^-1 stage's barrier)
    }
    // pass B: the length-scale sums, eight parameters per staging
    const int last_kc = ((d - 1) >> 3) << 3;  // the chunk pass A left staged
    for (int kc = last_kc; kc >= 0; kc -= 8) {
      if (kc != last_kc) {
        __syncthreads();
        stage_features(kc, false);
        __syncthreads();
      }
      double acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kc + u < d) {
          const TIO il2 = (TIO)ellv[32 + kc + u];
          TIO sacc = TIO(0);
#pragma unroll
          for (int tile = 0; tile < 3; ++tile) {
            TIO a[2], b[4];
#pragma unroll
            for (int r = 0; r < 2; ++r) a[r] = (TIO)feat[u * 128 + tile_i0(tile) + ty + 16 * (2 * half + r)];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = (TIO)feat[u * 128 + tile_j0(tile) + tx * 4 + c];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const TIO df = a[r] - b[c];
                sacc += coef[tile][r][c] * (df * df * il2);
              }
          }
          acc[u] = (double)sacc;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (kc + u < d) {
          const double sk = wave_sum(acc[u]);
          if (lane == 0) red[wave * 35 + 2 + kc + u] = sk;
        }
      }
    }
    __syncthreads();
    if (t < p) {
      double sacc = 0;
      for (int w = 0; w < 8; ++w) sacc += red[w * 35 + t];
      out->grad[t] = 0.5 * sacc;
    }
    __syncthreads();
    if (t == 0) {
      __threadfence_system();
      out->done = 3;
      g_leaf_stamps[205] = (long long)__builtin_readcyclecounter();
    }
  }
}

template <typename TIO, int NU2>
__global__ void __launch_bounds__(512, 2) small_eval_kernel(SmallEval g) {
  extern __shared__ __align__(16) char smem_raw[];
  small_eval_body<TIO, NU2>(g, smem_raw);
}

// ---- the bounded L-BFGS step of lbfgs_step.hpp, one wavefront wide: lane i owns dimension i (the GP has at most 34 parameters
// here), a dot product is a reduction across the wave.  Same method, same constants, same decisions, same evaluation counting as
// lbfgs_begin / lbfgs_advance (which stay the host's implementation and this code's specification: a single thread walking
// the host form on the device took 70 us per evaluation -- longer than the evaluation -- because every load of the optimiser's
// state is a dependent round trip); sums run in another order, so iterates agree with the host form to rounding, not bit for
// bit (tests/test_gpu_fit.py compares the two on the same fits).  The vectors live in the run's LbfgsState in global memory
// (lane-indexed loads: never the scalar cache), the scalars in the wave's registers.
struct WaveLbfgs {
  int phase, nevals, iterations, converged, hcount;
  double f, step, gs;
  double rho[LBFGS_MAXM];
};

template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum / maximum over the 64 lanes, the result in every lane; fixed order: within quads, octets, rows of 16 (DPP), then the four rows
__device__ __forceinline__ double wave_allsum(double v) {
  v += dpp64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp64<0x141>(v);   // row_half_mirror
  v += dpp64<0x140>(v);   // row_mirror
  double r = readlane(v, 0);
  r += readlane(v, 16);
  r += readlane(v, 32);
  r += readlane(v, 48);
  return r;
}
__device__ __forceinline__ double wave_allmax(double v) {
  v = fmax(v, dpp64<0xB1>(v));
  v = fmax(v, dpp64<0x4E>(v));
  v = fmax(v, dpp64<0x141>(v));
  v = fmax(v, dpp64<0x140>(v));
  return fmax(fmax(readlane(v, 0), readlane(v, 16)), fmax(readlane(v, 32), readlane(v, 48)));
}

// proposes xn = clip(x + step d) (lane-wise); false: the step no longer changes x
__device__ __forceinline__ bool wl_propose(WaveLbfgs& w, LbfgsState* st, int i, bool act) {
  double dx = 0, gsl = 0;
  if (act) {
    double v = st->x[i] + w.step * st->d[i];
    v = fmin(fmax(v, st->lo[i]), st->hi[i]);
    st->xn[i] = v;
    dx = v - st->x[i];
    gsl = st->g[i] * dx;
  }
  w.gs = wave_allsum(gsl);
  return wave_allmax(fabs(dx)) != 0.0;
}

// start of an outer iteration at (x, f, g): convergence test, search direction, first trial point; returns the next phase
__device__ __forceinline__ int wl_begin_iteration(WaveLbfgs& w, LbfgsState* st, int i, bool act, int maxeval, double pgtol) {
  for (int guard = 0; guard < 3; ++guard) {
    if (w.nevals >= maxeval) return 2;
    const double xi = act ? st->x[i] : 0.0, gi = act ? st->g[i] : 0.0;
    const double loi = act ? st->lo[i] : 0.0, hii = act ? st->hi[i] : 0.0;
    const bool at_lo = xi <= loi && gi > 0, at_hi = xi >= hii && gi < 0;
    const double pg = (act && !(at_lo || at_hi)) ? gi : 0.0;
    if (wave_allmax(fabs(pg)) <= pgtol * fmax(1.0, fabs(w.f))) {
      w.converged = 1;
      return 2;
    }
    // two-loop recursion on the free variables; the history pairs of this lane's dimension are fetched up front (independent
    // loads: one round trip to the L2 instead of one per pair inside the dependent chain of reductions)
    double q = pg;
    double a[LBFGS_MAXM], sh[LBFGS_MAXM], yh[LBFGS_MAXM];
#pragma unroll
    for (int h = 0; h < LBFGS_MAXM; ++h) {
      const bool on = act && h < w.hcount;
      sh[h] = on ? st->S[h][i] : 0.0;
      yh[h] = on ? st->Y[h][i] : 0.0;
    }
#pragma unroll
    for (int h = LBFGS_MAXM - 1; h >= 0; --h) {
      a[h] = 0;
      if (h < w.hcount) {
        a[h] = w.rho[h] * wave_allsum(sh[h] * q);
        q -= a[h] * yh[h];
      }
    }
    double gamma = 1.0;
    if (w.hcount > 0) {
      double sl = 0, yl = 0;
#pragma unroll
      for (int h = 0; h < LBFGS_MAXM; ++h)
        if (h == w.hcount - 1) { sl = sh[h]; yl = yh[h]; }
      const double sy = wave_allsum(sl * yl), yy = wave_allsum(yl * yl);
      if (yy > 0) gamma = sy / yy;
    }
    q *= gamma;
#pragma unroll
    for (int h = 0; h < LBFGS_MAXM; ++h) {
      if (h < w.hcount) {
        const double bq = w.rho[h] * wave_allsum(yh[h] * q);
        q += sh[h] * (a[h] - bq);
      }
    }
    double dv = (pg == 0.0) ? 0.0 : -q;
    double dg = wave_allsum(dv * gi);
    if (!(dg < 0)) {  // not a descent direction: fall back to projected steepest descent
      w.hcount = 0;
      dv = -pg;
      dg = wave_allsum(dv * gi);
    }
    if (act) st->d[i] = dv;
    w.step = 1.0;
    if (w.hcount == 0) {
      const double dn = wave_allsum(dv * dv);
      w.step = fmin(1.0, 1.0 / sqrt(fmax(dn, 1e-300)));
    }
    if (wl_propose(w, st, i, act)) return 1;
    if (w.hcount > 0) {  // the first step does not move x: retry once from steepest descent with a clean history
      w.hcount = 0;
      continue;
    }
    return 2;
  }
  return 2;
}

// feeds the evaluation of the requested point (fe, the wave's gradient entry ge); true while another evaluation is wanted
__device__ __forceinline__ bool wl_advance(WaveLbfgs& w, LbfgsState* st, int i, bool act, double fe, double ge, int maxeval, int m,
                                           double pgtol, double ftol, bool fixed_work) {
  ++w.nevals;
  if (fe != fe) fe = INFINITY;
  int next = 2;
  if (w.phase == 0) {
    w.f = fe;
    if (act) st->g[i] = ge;
    next = (fe - fe == 0.0) ? wl_begin_iteration(w, st, i, act, maxeval, pgtol) : 2;
  } else if (w.phase == 1) {
    const double fn = fe;
    if ((fn - fn == 0.0) && fn <= w.f + 1e-4 * w.gs) {
      ++w.iterations;
      const double sv = act ? st->xn[i] - st->x[i] : 0.0, yv = act ? ge - st->g[i] : 0.0;
      const double sy = wave_allsum(sv * yv), ss = wave_allsum(sv * sv), yy = wave_allsum(yv * yv);
      if (sy > 1e-10 * sqrt(ss * yy) && sy > 0) {
        if (w.hcount == m) {  // drop the oldest pair
          for (int h = 1; h < w.hcount; ++h) {
            if (act) {
              st->S[h - 1][i] = st->S[h][i];
              st->Y[h - 1][i] = st->Y[h][i];
            }
          }
#pragma unroll
          for (int h = 1; h < LBFGS_MAXM; ++h) w.rho[h - 1] = w.rho[h];
          --w.hcount;
        }
        if (act) {
          st->S[w.hcount][i] = sv;
          st->Y[w.hcount][i] = yv;
        }
#pragma unroll
        for (int h = 0; h < LBFGS_MAXM; ++h)
          if (h == w.hcount) w.rho[h] = 1.0 / sy;
        ++w.hcount;
      }
      const double fold = w.f;
      if (act) {
        st->x[i] = st->xn[i];
        st->g[i] = ge;
      }
      w.f = fn;
      if (fold - w.f <= ftol * fmax(1.0, fabs(w.f))) {
        w.converged = 1;
        next = 2;
      } else {
        next = wl_begin_iteration(w, st, i, act, maxeval, pgtol);
      }
    } else {
      double nstep = 0.5 * w.step;
      if ((fn - fn == 0.0) && w.gs < 0) {
        const double denom = 2.0 * (fn - w.f - w.gs);
        if (denom > 0) nstep = fmin(fmax(-w.gs * w.step / denom, 0.1 * w.step), 0.5 * w.step);
      }
      w.step = nstep;
      const bool again = !(w.step < 1e-20) && w.nevals < maxeval && wl_propose(w, st, i, act);
      if (again) {
        next = 1;
      } else if (w.hcount > 0) {
        w.hcount = 0;
        next = w.nevals < maxeval ? wl_begin_iteration(w, st, i, act, maxeval, pgtol) : 2;
      } else {
        next = 2;
      }
    }
  }
  if (next == 1) {
    w.phase = 1;
    return true;
  }
  if (fixed_work && w.nevals < maxeval) {
    w.phase = 2;
    return true;
  }
  w.phase = 3;
  return false;
}

// out-of-line copy of the evaluation for small_fit_kernel: a register allocation of its own (inlined into the loop the kernel
// spilled 216 VGPRs)
template <typename TIO, int NU2>
__device__ __attribute__((noinline)) void small_eval_call(const SmallEval* g, char* smem_raw) {
  small_eval_body<TIO, NU2>(*g, smem_raw);
}

// ---- a whole optimiser run in ONE launch (n <= 128): the evaluation above in a loop, with the bounded L-BFGS step and the fit's
// objective wrapper (fit.rs:94-133: clamped parameters, failed factorisation -> +inf, capture of the best evaluation) on the
// device, in wave 0.  The host launches one such kernel per optimiser run (gradmin.rs:19-31: the runs are independent), all
// side by side, and collects.  The ~40 us round trip through the host per evaluation (graph launch + completion wake-up +
// three host threads on the runtime's lock) becomes ~5 us of wave-wide vector arithmetic.
template <typename TIO, int NU2>
__global__ void __launch_bounds__(512, 2) small_fit_kernel(const SmallFit* __restrict__ fs) {
  // one workgroup = one optimiser run: the runs of a fit are the workgroups of ONE launch (round 4: one launch per run on a
  // stream of its own -- three streams per fit, and HIP maps streams onto a handful of hardware queues: fits side by side
  // then queued behind each other's persistent kernels)
  const SmallFit f = fs[blockIdx.x];
  extern __shared__ __align__(16) char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int p = f.ev.d + 2;
  LbfgsState* st = f.st;
  int* flags = reinterpret_cast<int*>(smem_raw + 163840 - 64);  // [0] another evaluation is wanted, [1] its ping-pong target
  SmallFitResult* res = f.res;
  EvalParams* P = const_cast<EvalParams*>(f.ev.P);
  const bool act = wave == 0 && lane < p;
  // wave 0's view of the run (registers: a scalar stored to global memory and re-read through a uniform address could come back
  // stale from the scalar cache)
  WaveLbfgs w{};
  int best_idx = -1, best_eval = 0, n_evals = 0, n_not_pd = 0;
  double best_lml = -INFINITY;
  double cur_param = 0.0;  // wave 0, lane = parameter: what the evaluation in flight runs with
  double best_th = 0.0, best_par = 0.0;  // wave 0, lane = parameter: the captured evaluation's (for the pinned copy of the result)
  // wave 0: the requested point -> clamped linear-space parameters (fit.rs:94-96; the noise is not clamped, :96)
  auto publish_request = [&]() {
    if (act) {
      const double th = w.phase == 1 ? st->xn[lane] : st->x[lane];
      double v = exp(th);
      if (lane >= 1) v = v < f.lo[lane] ? f.lo[lane] : (f.hi[lane] < v ? f.hi[lane] : v);  // bounded_value.rs:43-56
      cur_param = v;
      if (lane == 0) P->noise = v;
      else if (lane == 1) P->amp = v;
      else P->ell[lane - 2] = v;
    }
    if (lane == 0) flags[1] = best_idx < 0 ? 0 : 1 - best_idx;  // never overwrite the captured best (fit.rs:116-125)
    __threadfence();
  };
  if (wave == 0) {
    // log-space box (fit.rs:137-146) and the clipped start point
    if (act) {
      const double l = log(f.lo[lane]), h = log(f.hi[lane]);
      st->lo[lane] = l;
      st->hi[lane] = h;
      st->x[lane] = fmin(fmax(f.x0[lane], l), h);
    }
    w.phase = 0;
    w.f = INFINITY;
    w.step = 1.0;
    if (lane == 0) flags[0] = 1;
    publish_request();
  }
  __syncthreads();
  for (int eval_idx = 0; eval_idx < f.maxeval + 1; ++eval_idx) {  // bounded whatever the state machine does
    const int target = flags[1];
    __syncthreads();  // the flags sit behind the evaluation's scratch: read before it starts
    SmallEval g = f.ev;
    g.Kinv = f.Kinv[target];
    g.alpha = f.alpha[target];
    if (f.Xinv[target]) {
      g.W2 = f.Xinv[target];
      g.ldiag = f.ldiag[target];
    }
    g.mode = 7;
    small_eval_call<TIO, NU2>(&g, smem_raw);
    __threadfence();
    __syncthreads();
    if (wave == 0) {
      const EvalOut* out = g.hout;
      double lml = __hip_atomic_load(&out->lml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      double ge = act ? __hip_atomic_load(&out->grad[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      const int info = __hip_atomic_load(&out->info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ok = info == 0 && (lml - lml == 0.0) && wave_allmax((ge - ge == 0.0) ? 0.0 : 1.0) == 0.0;
      if (!ok) {  // handled like a failed factorisation (lml.rs:47-50 -> fit.rs:105-112): objective +inf, zero gradient
        lml = -INFINITY;
        ge = 0.0;
        ++n_not_pd;
      }
      ++n_evals;
      const double th = act ? (w.phase == 1 ? st->xn[lane] : st->x[lane]) : 0.0;
      if (eval_idx < f.trace_cap) {
        if (act) {
          f.trace_theta[(size_t)eval_idx * p + lane] = th;
          f.trace_grad[(size_t)eval_idx * p + lane] = ge;
        }
        if (lane == 0) f.trace_lml[eval_idx] = lml;
      }
      if (ok && (best_idx < 0 || lml > best_lml)) {  // strictly greater wins; ties keep the earlier evaluation
        best_idx = target;
        best_lml = lml;
        best_eval = eval_idx;
        if (act) {
          res->best_theta[lane] = th;
          res->best_params[lane] = cur_param;
          best_th = th;
          best_par = cur_param;
        }
      }
      // fit.rs:128-133: the optimiser minimises -lml
      const bool more = wl_advance(w, st, lane, act, ok ? -lml : INFINITY, -ge, f.maxeval, f.memory < 1 ? 1 : (f.memory > LBFGS_MAXM ? LBFGS_MAXM : f.memory),
                                   f.pgtol, f.ftol, f.fixed_work != 0);
      if (lane == 0) flags[0] = more ? 1 : 0;
      if (more) publish_request();
    }
    __syncthreads();
    const int more = flags[0];
    __syncthreads();
    if (!more) break;
  }
  if (t == 0) {
    res->best_idx = best_idx;
    res->best_lml = best_lml;
    res->best_eval = best_eval;
    res->n_evals = n_evals;
    res->n_not_pd = n_not_pd;
  }
  // the run is over: its result to the pinned block and the word its host thread polls (every wave's device writes of the last
  // evaluation are behind an agent-scope fence and a barrier; a later kernel of any stream starts with an acquire of its own)
  if (wave == 0 && f.hres) {
    SmallFitResult* h = f.hres;
    if (act) {
      h->best_theta[lane] = best_th;
      h->best_params[lane] = best_par;
    }
    if (lane == 0) {
      h->best_idx = best_idx;
      h->best_lml = best_lml;
      h->best_eval = best_eval;
      h->n_evals = n_evals;
      h->n_not_pd = n_not_pd;
    }
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(f.hdone, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <typename T>
void launch_small_fit(const SmallFit* fs_dev, int nruns, int nu2, hipStream_t s) {
  const dim3 grid(nruns), block(512);
  const size_t lds = 163840;  // the whole LDS of a CU
  switch (nu2) {
    case 0: hipLaunchKernelGGL((small_fit_kernel<T, 0>), grid, block, lds, s, fs_dev); break;
    case 1: hipLaunchKernelGGL((small_fit_kernel<T, 1>), grid, block, lds, s, fs_dev); break;
    case 3: hipLaunchKernelGGL((small_fit_kernel<T, 3>), grid, block, lds, s, fs_dev); break;
    default: hipLaunchKernelGGL((small_fit_kernel<T, 5>), grid, block, lds, s, fs_dev); break;
  }
}
template void launch_small_fit<double>(const SmallFit*, int, int, hipStream_t);
template void launch_small_fit<float>(const SmallFit*, int, int, hipStream_t);

template <typename T>
void launch_small_eval(const SmallEval& g, int nu2, hipStream_t s) {
  const dim3 grid(1), block(512);
  const size_t lds = 163840;  // the whole LDS of a CU: the block image + the kernel's scratch
  switch (nu2) {
    case 0: hipLaunchKernelGGL((small_eval_kernel<T, 0>), grid, block, lds, s, g); break;
    case 1: hipLaunchKernelGGL((small_eval_kernel<T, 1>), grid, block, lds, s, g); break;
    case 3: hipLaunchKernelGGL((small_eval_kernel<T, 3>), grid, block, lds, s, g); break;
    default: hipLaunchKernelGGL((small_eval_kernel<T, 5>), grid, block, lds, s, g); break;
  }
}
template void launch_small_eval<double>(const SmallEval&, int, hipStream_t);
template void launch_small_eval<float>(const SmallEval&, int, hipStream_t);

// Per-device one-time setup: kernels that use more than 64 KiB of dynamic LDS need the attribute raised.  Called from
// hbegp_ctx_create() for every device, before any stream capture.
static void init_dag_kernels();  // dag_kernel.inc.hpp (end of this file)
static void set_lds_attr(const void* fn, int bytes, const char* what) {
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
template <typename T, int TILE>
static void init_gemm_attr() {
  set_lds_attr(reinterpret_cast<const void*>(&gemm_kernel<T, TILE>), 163840, "gemm_kernel: dynamic LDS limit");
}
// Throws std::runtime_error (caught in hbegp_ctx_create): a kernel that cannot get its LDS would be rejected at every launch.
void init_kernels() {
  init_gemm_attr<double, 128>(); init_gemm_attr<double, 64>(); init_gemm_attr<double, 32>();
  init_gemm_attr<float, 128>(); init_gemm_attr<float, 64>(); init_gemm_attr<float, 32>();
  set_lds_attr(reinterpret_cast<const void*>(&leaf_kernel<double, double>), (int)LeafGeom<double>::LDS_BYTES, "leaf_kernel<f64>: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&leaf_kernel<double, float>), (int)LeafGeom<double>::LDS_BYTES, "leaf_kernel<f32>: dynamic LDS limit");
  init_dag_kernels();
  const int lb = 163840;
  // kmat / gradtrace: 2 d 64 elements of dynamic LDS (64 KiB at d = 64 in f64) beside ~21 KiB of static LDS
  {
    const void* fns[] = {
        (const void*)&kmat_kernel<double, 0>, (const void*)&kmat_kernel<double, 1>, (const void*)&kmat_kernel<double, 3>, (const void*)&kmat_kernel<double, 5>,
        (const void*)&kmat_kernel<float, 0>, (const void*)&kmat_kernel<float, 1>, (const void*)&kmat_kernel<float, 3>, (const void*)&kmat_kernel<float, 5>,
        (const void*)&gradtrace_kernel<double, 0>, (const void*)&gradtrace_kernel<double, 1>, (const void*)&gradtrace_kernel<double, 3>, (const void*)&gradtrace_kernel<double, 5>,
        (const void*)&gradtrace_kernel<float, 0>, (const void*)&gradtrace_kernel<float, 1>, (const void*)&gradtrace_kernel<float, 3>, (const void*)&gradtrace_kernel<float, 5>};
    for (const void* fn : fns) set_lds_attr(fn, 2 * MAXD * 64 * 8, "kmat / gradtrace kernel: dynamic LDS limit");
  }
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<double, 0>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<double, 1>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<double, 3>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<double, 5>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<float, 0>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<float, 1>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<float, 3>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_eval_kernel<float, 5>), lb, "small_eval_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<double, 0>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<double, 1>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<double, 3>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<double, 5>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<float, 0>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<float, 1>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<float, 3>), lb, "small_fit_kernel: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&small_fit_kernel<float, 5>), lb, "small_fit_kernel: dynamic LDS limit");
}

__global__ void set_info_kernel(int* info, int value) { *info = value; }
void launch_set_info(int* info, int value, hipStream_t s) { hipLaunchKernelGGL(set_info_kernel, dim3(1), dim3(1), 0, s, info, value); }

__global__ void reset_out_kernel(EvalOut* out) { poison_out(out, threadIdx.x); }
void launch_reset_out(EvalOut* out, hipStream_t s) { hipLaunchKernelGGL(reset_out_kernel, dim3(1), dim3(128), 0, s, out); }

// device result block -> pinned host block, then the evaluation's serial number behind a system-scope fence (engine.hpp)
__global__ void __launch_bounds__(128) publish_out_kernel(const EvalOut* out, EvalOut* hout, const EvalParams* __restrict__ P) {
  publish_out_in_block(out, hout, P);
}
void launch_publish_out(const EvalOut* out, EvalOut* hout, const EvalParams* P, hipStream_t s) {
  hipLaunchKernelGGL(publish_out_kernel, dim3(1), dim3(128), 0, s, out, hout, P);
}

#include "dag_kernel.inc.hpp"

}  // namespace hbegp
