// dag_kernel.inc.hpp — included at the end of kernels.hip (inside namespace hbegp).
//
// Device-scheduled Cholesky + inverse-factor recursion: ONE persistent launch replaces the chain of diagonal-block and
// tile-GEMM launches of chol_inv (lml.rs:47 `factorizec` + the factor's inverse that lml.rs:62 `invc` needs).
//
//   * Workgroups of 512 threads (one per CU: the diagonal-block task needs the whole LDS) pull tasks from an ordered
//     queue with one returning agent-scope atomic add.  A task is a 128x128 diagonal block (leaf_body), or a 128x64 /
//     64x64 output tile of one of the recursion's products over its whole contraction range.
//   * Dependencies are counters in global memory: a task waits until each of its (at most DAG_MAXWAIT) counters has
//     reached its value, and bumps its own counter once its results are visible.  The queue order is a topological
//     order of the graph (host: dag_plan.hpp, checked by dag_plan_validate), so whichever workgroups are resident make
//     progress -- nothing assumes co-residency, dispatch order or placement.
//   * Visibility (CDNA4: per-CU L1 is never refreshed by other CUs' stores, per-XCD L2s are write-back): results are
//     stored write-through (sc1), every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a
//     barrier, then one lane bumps the counter.  The consumer polls with relaxed agent-scope loads from one lane, runs
//     ONE agent-scope acquire (buffer_inv sc1: drops this CU's L1 lines), waits for it, and the workgroup meets at a
//     barrier before any wave loads operands with plain loads.
//   * Every wait is bounded (DagLaunch::wait_ticks of the constant 100 MHz clock; the host sets max(2 s, 200 x the plan's
//     simulated makespan), HBEGP_DAG_WAIT_S overrides -- the clock keeps running while a queue is preempted): on expiry the task index is recorded, info becomes
//     DAG_INFO_TIMEOUT and every workgroup drains out -- a scheduling bug ends in an error code, not in a hung GPU.
//   * Not positive definite (info > 0, set by a diagonal block): later tasks skip their work but still bump their
//     counters, so the queue drains at once.
//
// Arithmetic per output element is the same sequence of MFMA accumulations as in gemm_kernel (k ascending in steps of
// four, operands identical), so results are bitwise equal to the launch-per-product path.

// Contraction depth of one pipeline stage of the task-queue tiles, in units of Cfg<T>::BK (128 bytes of k).  Experiment (round
// 3): f64 with 2 -> 32 elements per stage, as f32 has -- twice the MFMAs between two stage barriers (32 per wave), half as many
// barrier bubbles per flop; the old values of a beta = 1 tile then live in registers for both tile shapes (the 128x64 stash no
// longer fits the LDS).  Measured SLOWER: 2048-deep 128x64 tile 127.3 vs 123.3 us, fit+predict/s 1.625 vs 1.665 (same box,
// 0 spills either way): the barrier bubble is not what costs the 9 % between the stage loop and the MFMA rate.  Stays 1.
#ifndef DAG_F64_KMUL
#define DAG_F64_KMUL 1
#endif
// Stage-loop form per tile shape (dag_gemm_tile): 0 = the shipped loop (a stage's fragments in registers, reads / loads / stores in
// clumps); 1 = one k-step's fragments double-buffered, every non-MFMA instruction in the shadow of an MFMA.  Form 1 is round 5's
// experiment: bitwise equal, +1.6 % on the tile alone, -0.5 % in the fit -- NOT in the product build (these
// defaults), kept as a compile-time form for tools/tile_ubench.hip, which also switches parts of it off (PIPE = 1 + 16 * ABL) to
// price the loop's data path.  Two more forms were built on it, measured and removed (commit 392ba09 has them): three register
// sets with loads three stages ahead (88.3 % of the MFMA rate alone, -4 % in the fit) and an LDS-DMA ring (global_load_lds into
// unpadded, source-swizzled stage images: 80-83 %).
#ifndef DAG_PIPE_128x128
#define DAG_PIPE_128x128 0
#endif
#ifndef DAG_PIPE_128x64
#define DAG_PIPE_128x64 0
#endif
#ifndef DAG_PIPE_64x64
#define DAG_PIPE_64x64 0
#endif
template <typename T, int TA, int TB>
struct DagGeom {
  using C = Cfg<T>;
  static constexpr int NT = 512;
  static constexpr int BK = sizeof(T) == 8 ? DAG_F64_KMUL * C::BK : C::BK;
  static constexpr int SK = BK + 2;                    // LDS row stride, operand stored [outer][k]
  static constexpr int SMA = TA + 16, SMB = TB + 16;   // LDS row stride, operand stored [k][outer]
  static constexpr int LDSA = (TA * SK > BK * SMA) ? TA * SK : BK * SMA;
  static constexpr int LDSB = (TB * SK > BK * SMB) ? TB * SK : BK * SMB;
  static constexpr int NCHA = TA * BK / C::VEC / NT, NCHB = TB * BK / C::VEC / NT;  // 16-byte chunks per thread per stage
  static constexpr int WM = 4, WN = 2;                 // wave grid
  static constexpr int TMA = TA / WM / 16, TMB = TB / WN / 16;  // 16x16 MFMA blocks per wave
  static_assert(NCHA >= 1 && NCHB >= 1 && TMA >= 1 && TMB >= 1, "tile too small for 512 threads");
};

constexpr int DAG_LDS_CTL_OFF = (int)((LeafGeom<double>::LDS_BYTES + 15) / 16 * 16);
constexpr int DAG_LDS_BYTES = DAG_LDS_CTL_OFF + 128;  // two control blocks of 16 dwords
static_assert(DAG_LDS_BYTES <= 163840, "the diagonal block and the control words must fit the CU's LDS");

// pull / fetch: the task loop's hooks for claiming the NEXT queue entry late in this tile's life -- pull() (thread 0: the
// atomic on the queue head) once no operand load is left to issue, fetch() (wave 0: the entry's descriptor) before the last
// stage, so that both round trips hide under the last MFMA stages and the entry is claimed only ~2 us before this workgroup is
// free (claiming it a whole task earlier parks the chain's tasks behind bulk tiles: measured 2.05 -> 3.06 ms per evaluation).
// -DDAG_STAMP_INNER (diagnostic build, tools/trace_inner.py): two more time stamps per tile task -- first stage in the LDS (the
// first MFMA can start) and last MFMA issued -- packed into the trace's CU-id word (low 32 bits of the 100 MHz clock each).
template <typename T, int TA, int TB, int PIPE, typename PullFn, typename FetchFn>
__device__ __forceinline__ void dag_gemm_tile(int flags, int row0, int col0, int kbeg, int kend, T* __restrict__ W1,
                                              T* __restrict__ W2, T* __restrict__ W3, T* __restrict__ Kinv, int ld, char* smem_raw,
                                              PullFn pull, FetchFn fetch, unsigned long long* inner = nullptr) {
  using C = Cfg<T>;
  using G = DagGeom<T, TA, TB>;
  using vec_t = typename C::vec_t;
  using acc_t = typename C::acc_t;
  constexpr int VEC = C::VEC, BK = G::BK, SK = G::SK, SMA = G::SMA, SMB = G::SMB, NCHA = G::NCHA, NCHB = G::NCHB;
  constexpr int TMA = G::TMA, TMB = G::TMB, NT = G::NT;

#ifdef DAG_HEAD_DELAY  /* diagnostic build: every tile task starts DAG_HEAD_DELAY x 64 cycles late -- does the fit rate follow the per-task fixed cost? */
  for (int q = 0; q < DAG_HEAD_DELAY; ++q) __builtin_amdgcn_s_sleep(1);
#endif
  const int akm = (flags & DAGF_AKM) ? 1 : 0, bkm = (flags & DAGF_BKM) ? 1 : 0;
  const T* Ag = (flags & DAGF_A3) ? W3 : ((flags & DAGF_ABUF) ? W2 : W1);
  const T* Bg = (flags & DAGF_B3) ? W3 : ((flags & DAGF_BBUF) ? W2 : W1);
  T* Cg = (flags & DAGF_CKINV) ? Kinv : ((flags & DAGF_C3) ? W3 : ((flags & DAGF_CBUF) ? W2 : W1));
  const int nstages = (kend - kbeg) / BK;

  T* lds = reinterpret_cast<T*>(smem_raw);  // [A buf0 | A buf1 | B buf0 | B buf1]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  constexpr int CPK = BK / VEC;                   // chunks per slab row, operand stored [outer][k]
  constexpr int RPP_K = NT / CPK;                 // slab rows per pass
  constexpr int CPRA = TA / VEC, CPRB = TB / VEC; // chunks per slab row, operand stored [k][outer]
  constexpr int RPP_MA = NT / CPRA, RPP_MB = NT / CPRB;
  const int a_r0 = akm ? t / CPRA : t / CPK, a_c0 = akm ? (t % CPRA) * VEC : (t % CPK) * VEC;
  const int b_r0 = bkm ? t / CPRB : t / CPK, b_c0 = bkm ? (t % CPRB) * VEC : (t % CPK) * VEC;
  const int a_rpp = akm ? RPP_MA : RPP_K, b_rpp = bkm ? RPP_MB : RPP_K;
  const int a_lds0 = a_r0 * (akm ? SMA : SK) + a_c0, a_ldsq = a_rpp * (akm ? SMA : SK);
  const int b_lds0 = 2 * G::LDSA + b_r0 * (bkm ? SMB : SK) + b_c0, b_ldsq = b_rpp * (bkm ? SMB : SK);

  const T* pA = Ag + (akm ? (size_t)(kbeg + a_r0) * ld + row0 + a_c0 : (size_t)(row0 + a_r0) * ld + kbeg + a_c0);
  const T* pB = Bg + (bkm ? (size_t)(kbeg + b_r0) * ld + col0 + b_c0 : (size_t)(col0 + b_r0) * ld + kbeg + b_c0);
  const size_t a_step = akm ? (size_t)BK * ld : (size_t)BK, b_step = bkm ? (size_t)BK * ld : (size_t)BK;
  const size_t a_qs = (size_t)a_rpp * ld, b_qs = (size_t)b_rpp * ld;

  vec_t ra0[NCHA], rb0[NCHB], ra1[NCHA], rb1[NCHB];
  auto load_stage = [&](vec_t (&ra)[NCHA], vec_t (&rb)[NCHB]) {
#pragma unroll
    for (int q = 0; q < NCHA; ++q) ra[q] = *reinterpret_cast<const vec_t*>(pA + q * a_qs);
#pragma unroll
    for (int q = 0; q < NCHB; ++q) rb[q] = *reinterpret_cast<const vec_t*>(pB + q * b_qs);
    pA += a_step;
    pB += b_step;
  };
  auto store_stage = [&](int buf, vec_t (&ra)[NCHA], vec_t (&rb)[NCHB]) {
#pragma unroll
    for (int q = 0; q < NCHA; ++q) C::lds_store(lds + buf * G::LDSA + a_lds0 + q * a_ldsq, ra[q]);
#pragma unroll
    for (int q = 0; q < NCHB; ++q) C::lds_store(lds + buf * G::LDSB + b_lds0 + q * b_ldsq, rb[q]);
  };

  acc_t acc[TMA][TMB];
#pragma unroll
  for (int a = 0; a < TMA; ++a)
#pragma unroll
    for (int b = 0; b < TMB; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
  // beta = 1: the old values of the output tile are fetched now and added in the epilogue (same arithmetic as loading them
  // there; the loads' latency hides under the contraction -- it was a third of a 128-deep update's time)
  // The 64x64 tile keeps them in registers (8); the 128x64 tile (16: the kernel's diagonal block would spill) parks them in
  // the LDS behind the stage buffers until the epilogue.  Measured on the 128-deep 128x64 update: 13.8 -> 9.9 us.
  // the 128x128 tile has no room in the LDS for the old values beside the stage buffers: with beta = 1 it fetches them in the
  // epilogue (their latency is exposed once per task -- a task that replaces two); it never continues a sum (no DAGF_CINIT:
  // dag_plan_validate checks)
  constexpr bool NOACC = TB == 128;
  const bool accum = (flags & DAGF_ACC) != 0;
  const bool cinit = !NOACC && sizeof(T) == 8 && (flags & DAGF_CINIT) != 0;  // the old values start the accumulation (engine.hpp)
  constexpr bool PREFETCH_C = TA == 64 || (sizeof(T) == 8 && DAG_F64_KMUL > 1);
  constexpr int STASH_OFF = 2 * (G::LDSA + G::LDSB);  // in elements of T, behind [A buf0 | A buf1 | B buf0 | B buf1]
  static_assert(NOACC || PREFETCH_C || (size_t)(STASH_OFF + TA * TB) * sizeof(T) <= (size_t)DAG_LDS_CTL_OFF, "the stash must fit in front of the control words");
  static_assert((size_t)STASH_OFF * sizeof(T) <= (size_t)DAG_LDS_CTL_OFF, "the stage buffers must fit in front of the control words");
  T cold[PREFETCH_C ? TMA : 1][PREFETCH_C ? TMB : 1][4];
  if constexpr (PREFETCH_C) {
    const int er0p = row0 + wm * (TA / G::WM), ec0p = col0 + wn * (TB / G::WN) + (lane & 15);
#pragma unroll
    for (int a = 0; a < TMA; ++a)
#pragma unroll
      for (int b = 0; b < TMB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cold[a][b][r] = accum ? Cg[(size_t)(er0p + a * 16 + C::crow(lane, r)) * ld + ec0p + b * 16] : T(0);
    if constexpr (sizeof(T) == 8) {
      if (cinit) {
#pragma unroll
        for (int a = 0; a < TMA; ++a)
#pragma unroll
          for (int b = 0; b < TMB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = cold[a][b][r];
      }
    }
  }

  // f32: fp64 totals per F32_CHUNK contraction elements, exactly as gemm_kernel does (same chunk boundaries: same bits)
  constexpr bool CHUNKED = sizeof(T) == 4;
  static_assert(!CHUNKED || F32_CHUNK % BK == 0, "an accumulation chunk is a whole number of stages");
  int kabs = kbeg;  // f32: absolute contraction index behind the stages computed so far
  double tot[CHUNKED ? TMA : 1][CHUNKED ? TMB : 1][4];
  if constexpr (CHUNKED) {
#pragma unroll
    for (int a = 0; a < TMA; ++a)
#pragma unroll
      for (int b = 0; b < TMB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[a][b][r] = 0.0;
  }
  auto flush = [&]() {
    if constexpr (CHUNKED) {
#pragma unroll
      for (int a = 0; a < TMA; ++a)
#pragma unroll
        for (int b = 0; b < TMB; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[a][b][r] += (double)acc[a][b][r];
          acc[a][b] = acc_t{0, 0, 0, 0};
          // one 16x16 block at a time: left to itself the scheduler converts many blocks ahead of their additions, and the
          // 128x128 tile (64 total registers) then no longer fits beside the kernel's own live values (round 4: 4 spilled VGPRs)
          if constexpr (TA * TB >= 128 * 128) __builtin_amdgcn_sched_barrier(0);
        }
    }
  };

  const int soA = akm ? 1 : SK, skA = akm ? SMA : 1;
  const int soB = bkm ? 1 : SK, skB = bkm ? SMB : 1;
  const int fa0 = (wm * (TA / G::WM) + (lane & 15)) * soA + (lane >> 4) * skA;
  const int fb0 = 2 * G::LDSA + (wn * (TB / G::WN) + (lane & 15)) * soB + (lane >> 4) * skB;

  constexpr int NK = BK / 4;
  if constexpr (PIPE == 0) {
    // Same software pipeline as gemm_kernel's 64-tile: global loads two stages ahead in two register sets, LDS double
    // buffer, the fragments of a whole stage in registers, the next stage's first fragments read under the MFMAs of the
    // last k-step.
    T fa[NK][TMA], fb[NK][TMB];
    auto read_frags = [&](int buf, int k4) {
      const int ia = buf * G::LDSA + fa0 + k4 * 4 * skA, ib = buf * G::LDSB + fb0 + k4 * 4 * skB;
#pragma unroll
      for (int a = 0; a < TMA; ++a) fa[k4][a] = lds[ia + a * 16 * soA];
#pragma unroll
      for (int b2 = 0; b2 < TMB; ++b2) fb[k4][b2] = lds[ib + b2 * 16 * soB];
    };
    auto mfma_step = [&](int k4) {
#pragma unroll
      for (int a = 0; a < TMA; ++a)
#pragma unroll
        for (int b2 = 0; b2 < TMB; ++b2) acc[a][b2] = C::mfma(fa[k4][a], fb[k4][b2], acc[a][b2]);
    };
    auto stage = [&](int cur, bool do_load, vec_t (&la)[NCHA], vec_t (&lb)[NCHB], bool do_store, vec_t (&sa)[NCHA],
                     vec_t (&sb)[NCHB], bool has_next) {
      if (do_load) load_stage(la, lb);
      // f32 128x128 (NK = 8: 48 fragment registers beside 64 fp64 chunk totals): the fragments of a stage are read in two
      // halves, the second under the first half's MFMAs -- with all of them up front the kernel spilled (round 4: 4 VGPRs)
      constexpr bool SPLIT_READS = NK >= 8 && TA * TB >= 128 * 128;
      constexpr int KH = SPLIT_READS ? NK / 2 : NK - 1;  // last k-step read up front
#pragma unroll
      for (int k4 = 1; k4 <= KH; ++k4) read_frags(cur, k4);
      if constexpr (SPLIT_READS) {
#pragma unroll
        for (int k4 = 0; k4 + 1 < KH; ++k4) mfma_step(k4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = KH + 1; k4 < NK; ++k4) read_frags(cur, k4);
#pragma unroll
        for (int k4 = KH - 1; k4 + 2 < NK; ++k4) mfma_step(k4);
      } else {
#pragma unroll
        for (int k4 = 0; k4 + 2 < NK; ++k4) mfma_step(k4);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_store) store_stage(cur ^ 1, sa, sb);
      if (NK >= 2) mfma_step(NK - 2);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) read_frags(cur ^ 1, 0);
      mfma_step(NK - 1);
      __builtin_amdgcn_sched_group_barrier(0x100, TMA + TMB, 0);  // reads first: their latency hides under the MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, TMA * TMB, 0);
      __builtin_amdgcn_sched_barrier(0);
      kabs += BK;
      if (CHUNKED && kabs % F32_CHUNK == 0) flush();
    };
    if (nstages > 0) {
      load_stage(ra0, rb0);
      if (nstages > 1) load_stage(ra1, rb1);
      if constexpr (!PREFETCH_C && !NOACC) {
        if (accum) {  // old values of the output tile -> LDS stash (each thread its own 16 slots, lane-contiguous)
          T cst[TMA][TMB][4];
          const int er0p = row0 + wm * (TA / G::WM), ec0p = col0 + wn * (TB / G::WN) + (lane & 15);
#pragma unroll
          for (int a = 0; a < TMA; ++a)
#pragma unroll
            for (int b = 0; b < TMB; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) cst[a][b][r] = Cg[(size_t)(er0p + a * 16 + C::crow(lane, r)) * ld + ec0p + b * 16];
          store_stage(0, ra0, rb0);
          bool stash = true;
          if constexpr (sizeof(T) == 8) {
            if (cinit) {
              stash = false;
#pragma unroll
              for (int a = 0; a < TMA; ++a)
#pragma unroll
                for (int b = 0; b < TMB; ++b)
#pragma unroll
                  for (int r = 0; r < 4; ++r) acc[a][b][r] = cst[a][b][r];
            }
          }
          if (stash) {
#pragma unroll
            for (int a = 0; a < TMA; ++a)
#pragma unroll
              for (int b = 0; b < TMB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[STASH_OFF + ((a * TMB + b) * 4 + r) * NT + t] = cst[a][b][r];
          }
        } else {
          store_stage(0, ra0, rb0);
        }
      } else {
        store_stage(0, ra0, rb0);
      }
      __syncthreads();
      read_frags(0, 0);
  #ifdef DAG_STAMP_INNER
      unsigned long long st1 = 0;
      if (inner && t == 0) st1 = __builtin_amdgcn_s_memrealtime();
  #endif
      int s = 0;
      for (; s + 3 < nstages; s += 2) {
        stage(0, true, ra0, rb0, true, ra1, rb1, true);
        stage(1, true, ra1, rb1, true, ra0, rb0, true);
      }
      const int left = nstages - s;  // 1..3
      if (left <= 2) pull();
      if (left == 1) fetch();
      stage(0, left > 2, ra0, rb0, left > 1, ra1, rb1, left > 1);
      if (left > 2) pull();
      if (left == 2) fetch();
      if (left > 1) stage(1, false, ra1, rb1, left > 2, ra0, rb0, left > 2);
      if (left > 2) {
        fetch();
        stage(0, false, ra0, rb0, false, ra1, rb1, false);
      }
  #ifdef DAG_STAMP_INNER
      if (inner && t == 0) *inner = ((st1 & 0xffffffffull) << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull);
  #endif
    } else {
      pull();
      fetch();
    }
  } else {
    // PIPE 1 (round 5): the stage loop with every wave's LDS reads, global loads and LDS stores dealt one by one into the
    // shadows of its own MFMAs.  Why: the two waves of a SIMD share one fp64 MFMA pipe and the pipe serves the older wave
    // first, so the older wave runs ahead to the stage barrier and the younger one then runs most of its stage ALONE -- and
    // whenever a lone wave issues a clump of non-MFMA instructions (PIPE 0: 18 fragment reads + 4 global loads at the head of a
    // stage, 4 LDS stores in the middle) the pipe idles: the stage loop ran at 92 % of the MFMA rate.  Here the fragments of
    // ONE k-step are double-buffered (k-step k lives in set k & 1: 2 x (TMA + TMB) values instead of NK x), each MFMA is
    // followed by at most one fragment read of the NEXT k-step and one load or store, and sched_group_barrier pins that
    // order, so a wave alone keeps the pipe busy.  Same k-ascending chain of MFMA accumulations per element: same bits.
    // An interval = what lies between two stage barriers: MFMAs of the last k-step of stage s-1 (fragments already in
    // registers), then k-steps 0 .. NK-2 of stage s; the loads of stage s+2 sit in the first group, the LDS stores of stage s+1
    // in the group before the last one.
    static_assert(NK % 2 == 0 && NK >= 2, "k-step k lives in fragment set k & 1 across stage boundaries");
    // tools/tile_ubench only (wrong results, timing only): PIPE = 1 + 16 * ABL switches parts of the loop off -- 1: no global
    // loads, 2: no LDS stores, 4: no stage barrier, 8: no fragment reads (what is each worth beside the MFMAs?)
    constexpr int ABL = PIPE >> 4;
    constexpr int NF = TMA + TMB, NM = TMA * TMB;
    constexpr int RPM = (NF + NM - 1) / NM;  // fragment reads dealt behind one MFMA
    T fr[2][NF];
    auto read_frag = [&](int set, int buf, int k4, int i) {
      if (i < TMA) fr[set][i] = lds[buf * G::LDSA + fa0 + k4 * 4 * skA + i * 16 * soA];
      else fr[set][i] = lds[buf * G::LDSB + fb0 + k4 * 4 * skB + (i - TMA) * 16 * soB];
    };
    // the MFMAs of one k-step (fragments in set ms) with the reads of k-step rk of buffer rbuf (into set ms ^ 1) in their shadows
    auto group = [&](int ms, bool rd, int rbuf, int rk) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i / TMB][i % TMB] = C::mfma(fr[ms][i / TMB], fr[ms][TMA + i % TMB], acc[i / TMB][i % TMB]);
        if (rd) {
#pragma unroll
          for (int q = 0; q < RPM; ++q)
            if (i * RPM + q < NF) read_frag(ms ^ 1, rbuf, rk, i * RPM + q);
        }
      }
    };
    // pins the issue order of one group: MFMA, its fragment reads, then (first NV MFMAs) one global load / (first NW) one LDS store
    auto pin = [&](bool rd, int nv, int nw) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (rd && i * RPM < NF) __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
        if (i < nv) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (i < nw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    };
    constexpr int KST = NK >= 4 ? NK - 3 : 0;  // the k-step group that carries the LDS stores of the next stage
    auto interval = [&](int cur, bool g0, bool do_load, vec_t (&la)[NCHA], vec_t (&lb)[NCHB], bool do_store, vec_t (&sa)[NCHA],
                        vec_t (&sb)[NCHB]) {
      __builtin_amdgcn_sched_barrier(0);
      if (g0) {
        if (do_load && !(ABL & 1)) load_stage(la, lb);
        group((NK - 1) & 1, !(ABL & 8), cur, 0);
        pin(true, NCHA + NCHB, 0);
        __builtin_amdgcn_sched_barrier(0);
        kabs += BK;
        if (CHUNKED && kabs % F32_CHUNK == 0) flush();
      } else {
        if (do_load && !(ABL & 1)) load_stage(la, lb);
#pragma unroll
        for (int i = 0; i < NF; ++i) read_frag(0, cur, 0, i);
        if constexpr ((ABL & 8) != 0) {
#pragma unroll
          for (int i = 0; i < NF; ++i) read_frag(1, cur, 1, i);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k4 = 0; k4 + 1 < NK; ++k4) {
        const bool st = do_store && k4 == KST && !(ABL & 2);
        if (st) store_stage(cur ^ 1, sa, sb);
        group(k4 & 1, !(ABL & 8), cur, k4 + 1);
        pin(true, 0, k4 == KST ? NCHA + NCHB : 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!(ABL & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    };
    if (nstages > 0) {
      load_stage(ra0, rb0);
      if (nstages > 1) load_stage(ra1, rb1);
      if constexpr (!PREFETCH_C && !NOACC) {
        if (accum) {  // old values of the output tile -> LDS stash (as PIPE 0)
          T cst[TMA][TMB][4];
          const int er0p = row0 + wm * (TA / G::WM), ec0p = col0 + wn * (TB / G::WN) + (lane & 15);
#pragma unroll
          for (int a = 0; a < TMA; ++a)
#pragma unroll
            for (int b = 0; b < TMB; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) cst[a][b][r] = Cg[(size_t)(er0p + a * 16 + C::crow(lane, r)) * ld + ec0p + b * 16];
          store_stage(0, ra0, rb0);
          bool stash = true;
          if constexpr (sizeof(T) == 8) {
            if (cinit) {
              stash = false;
#pragma unroll
              for (int a = 0; a < TMA; ++a)
#pragma unroll
                for (int b = 0; b < TMB; ++b)
#pragma unroll
                  for (int r = 0; r < 4; ++r) acc[a][b][r] = cst[a][b][r];
            }
          }
          if (stash) {
#pragma unroll
            for (int a = 0; a < TMA; ++a)
#pragma unroll
              for (int b = 0; b < TMB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[STASH_OFF + ((a * TMB + b) * 4 + r) * NT + t] = cst[a][b][r];
          }
        } else {
          store_stage(0, ra0, rb0);
        }
      } else {
        store_stage(0, ra0, rb0);
      }
      __syncthreads();
#ifdef DAG_STAMP_INNER
      unsigned long long st1 = 0;
      if (inner && t == 0) st1 = __builtin_amdgcn_s_memrealtime();
#endif
      int s = 0;
      bool g0 = false;
        for (; s + 3 < nstages; s += 2) {
          interval(0, g0, true, ra0, rb0, true, ra1, rb1);
          g0 = true;
          interval(1, true, true, ra1, rb1, true, ra0, rb0);
        }
        const int left = nstages - s;  // 1..3
        if (left <= 2) pull();
        if (left == 1) fetch();
        interval(0, g0, left > 2, ra0, rb0, left > 1, ra1, rb1);
        if (left > 2) pull();
        if (left == 2) fetch();
        if (left > 1) interval(1, true, false, ra1, rb1, left > 2, ra0, rb0);
        if (left > 2) {
          fetch();
          interval(0, true, false, ra0, rb0, false, ra1, rb1);
        }
      // the last k-step of the last stage
      group((NK - 1) & 1, false, 0, 0);
      kabs += BK;
#ifdef DAG_STAMP_INNER
      if (inner && t == 0) *inner = ((st1 & 0xffffffffull) << 32) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull);
#endif
    } else {
      pull();
      fetch();
    }
  }

  flush();  // f32: what the last (partial) chunk holds
  // epilogue: write-through stores (read by other workgroups of this launch)
  const int er0 = row0 + wm * (TA / G::WM), ec0 = col0 + wn * (TB / G::WN) + (lane & 15);
  const bool neg = (flags & DAGF_NEG) != 0;
  T cend[NOACC ? TMA : 1][NOACC ? TMB : 1][4];
  if constexpr (NOACC) {
    if (accum) {  // uniform: all the loads first, one latency
#pragma unroll
      for (int a = 0; a < TMA; ++a)
#pragma unroll
        for (int b = 0; b < TMB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) cend[a][b][r] = Cg[(size_t)(er0 + a * 16 + C::crow(lane, r)) * ld + ec0 + b * 16];
    }
  }
#pragma unroll
  for (int a = 0; a < TMA; ++a)
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = er0 + a * 16 + C::crow(lane, r);
        T* p = Cg + (size_t)row * ld + ec0 + b * 16;
        T v;
        if constexpr (CHUNKED) v = (T)tot[a][b][r];
        else v = acc[a][b][r];
        if (neg) v = -v;
        if constexpr (PREFETCH_C) {
          if (accum && !cinit) v += cold[a][b][r];
        } else if constexpr (!NOACC) {
          if (accum && !cinit) v += lds[STASH_OFF + ((a * TMB + b) * 4 + r) * NT + t];
        } else {
          if (accum) v += cend[a][b][r];
        }
        gstore<true>(p, v);
      }
}

// The chain's tile (DAG_GEMM_32x64, engine.hpp): 32 x 64 outputs, one 16x16 block per wave (2 x 4 wave grid), contraction in
// passes of at most 128 elements.  All operand loads of a pass are issued at once (12 16-byte loads per thread in f64), land in
// the LDS as [outer][k] rows, one barrier, then every wave runs its chain of dependent MFMA steps (32 for a 128-deep pass:
// ~0.9 us with two waves per SIMD).  The staged 64x64 tile spends 5.8-6.1 us on the same 128-deep product (1.7 us of MFMA per
// wave, eight stage barriers, two-stage prefetch); between two diagonal blocks that is paid twice.
template <typename T, typename PullFn, typename FetchFn>
__device__ __forceinline__ void dag_gemm_tile_chain(int flags, int row0, int col0, int kbeg, int kend, T* __restrict__ W1,
                                                    T* __restrict__ W2, T* __restrict__ W3, int ld, char* smem_raw, PullFn pull,
                                                    FetchFn fetch) {
  using C = Cfg<T>;
  using vec_t = typename C::vec_t;
  using acc_t = typename C::acc_t;
  constexpr int TA = 32, TB = 64, KC = 128, VEC = C::VEC, NT = 512;
  constexpr int SK = KC + 4;               // LDS row stride (rows stay 16-byte aligned)
  constexpr int CPR = KC / VEC;            // 16-byte chunks per row of a pass
  constexpr int RPP = NT / CPR;            // rows per load pass
  constexpr int NLOAD = (TA + TB) / RPP;   // loads per thread per pass (f64: 12, f32: 6)
  static_assert(NT % CPR == 0 && (TA + TB) % RPP == 0 && TA % RPP == 0, "the load passes tile the operand rows");
  static_assert((size_t)(TA + TB) * SK * sizeof(T) <= (size_t)DAG_LDS_CTL_OFF, "both operands of a pass fit in front of the control words");
  const T* Ag = (flags & DAGF_A3) ? W3 : ((flags & DAGF_ABUF) ? W2 : W1);
  const T* Bg = (flags & DAGF_B3) ? W3 : ((flags & DAGF_BBUF) ? W2 : W1);
  T* Cg = (flags & DAGF_C3) ? W3 : ((flags & DAGF_CBUF) ? W2 : W1);
  T* lds = reinterpret_cast<T*>(smem_raw);  // [TA + TB][SK]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int lc = (t % CPR) * VEC, lr = t / CPR;
  const bool accum = (flags & DAGF_ACC) != 0, neg = (flags & DAGF_NEG) != 0;
  const int er0 = row0 + wm * 16, ec = col0 + wn * 16 + (lane & 15);
  T cold[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) cold[r] = accum ? Cg[(size_t)(er0 + C::crow(lane, r)) * ld + ec] : T(0);
  acc_t acc = acc_t{0, 0, 0, 0};
  constexpr bool CHUNKED = sizeof(T) == 4;
  double tot[4] = {0, 0, 0, 0};
  const int fa = (wm * 16 + (lane & 15)) * SK + (lane >> 4), fb = (TA + wn * 16 + (lane & 15)) * SK + (lane >> 4);
  bool hooked = false;
  for (int k0 = kbeg; k0 < kend; k0 += KC) {
    const int kc = min(KC, kend - k0);
    vec_t v[NLOAD];
    const bool in_range = lc < kc;
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
      const int row = q * RPP + lr;  // < TA: a row of A, else a row of B (uniform per q: TA % RPP == 0)
      const T* src = (q * RPP < TA) ? Ag + (size_t)(row0 + row) * ld + k0 + lc : Bg + (size_t)(col0 + row - TA) * ld + k0 + lc;
      v[q] = in_range ? *reinterpret_cast<const vec_t*>(src) : vec_t{};
    }
    if (!hooked) pull();
    if (k0 != kbeg) __syncthreads();  // the previous pass's fragments have been read
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) C::lds_store(lds + (q * RPP + lr) * SK + lc, v[q]);
    if (!hooked) fetch();
    hooked = true;
    __syncthreads();
    const int nsteps = kc / 4;
#pragma unroll 8
    for (int s4 = 0; s4 < nsteps; ++s4) {
      acc = C::mfma(lds[fa + 4 * s4], lds[fb + 4 * s4], acc);
      if constexpr (CHUNKED) {
        if ((k0 + 4 * s4 + 4) % F32_CHUNK == 0) {  // fp64 totals at the same absolute boundaries as the staged tiles
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[r] += (double)acc[r];
          acc = acc_t{0, 0, 0, 0};
        }
      }
    }
  }
  if (!hooked) { pull(); fetch(); }
  if constexpr (CHUNKED) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tot[r] += (double)acc[r];  // what a last partial chunk holds (zeros otherwise)
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    T vv;
    if constexpr (CHUNKED) vv = (T)tot[r];
    else vv = acc[r];
    if (neg) vv = -vv;
    if (accum) vv += cold[r];
    gstore<true>(Cg + (size_t)(er0 + C::crow(lane, r)) * ld + ec, vv);
  }
}

// DAG_LEAF_NOINLINE=1 compiles the diagonal block as a function of its own: nothing of it spills in the kernel body then (the
// callee saves 25 registers on its stack instead).  Round 2 found that build not reproducible run to run and kept it off; the
// cause (round 3, profiles/r03_leaf_race.txt) was not the call but a race inside leaf_body that every build had -- the helper
// waves re-read pivot rows that wave 0 overwrites in the same phase -- and the called build merely lost it more often.  With
// the pivot-row copy (LEAF_DIAG_COPY) the called build is bitwise reproducible too (0 deviations in 39,600 concurrent
// evaluations, 5 without the copy).  Variant 2 names the workgroup's dynamic LDS itself and keeps ds_ accesses.
#ifndef DAG_LEAF_NOINLINE
#define DAG_LEAF_NOINLINE 1  /* round 3: on.  Kernel body 0 spilled VGPRs (inlined: 21 f64 / 53 f32); fit+predict/s 1.690 vs 1.682 */
#endif
template <typename T>
#if DAG_LEAF_NOINLINE == 2
// variant 2: the function names the workgroup's dynamic LDS itself, so the block keeps local-address-space (ds_) accesses
__device__ __attribute__((noinline)) void dag_leaf_task(T* W1, T* W2, int ld, int blk, T* ldiag, int* info, char*, int dbg) {
  extern __shared__ __align__(16) char leaf_smem[];
  leaf_body<double, T, true>(W1, W2, ld, blk, ldiag, info, dbg, leaf_smem);
}
#elif DAG_LEAF_NOINLINE
__device__ __attribute__((noinline)) void dag_leaf_task(T* W1, T* W2, int ld, int blk, T* ldiag, int* info, char* smem_raw, int dbg) {
  leaf_body<double, T, true>(W1, W2, ld, blk, ldiag, info, dbg, smem_raw);
}
#else
__device__ __forceinline__ void dag_leaf_task(T* W1, T* W2, int ld, int blk, T* ldiag, int* info, char* smem_raw, int dbg) {
  leaf_body<double, T, true>(W1, W2, ld, blk, ldiag, info, dbg, smem_raw);
}
#endif

// The task loop.  A workgroup claims its next queue entry and fetches its descriptor under the last MFMA stages of the task it
// runs (dag_gemm_tile's pull / fetch hooks).  Between two tasks the hand-off then costs only what cannot overlap:
//   epilogue stores issued -> [wave 0: ONE look at the NEXT task's counters, in flight together with the stores' drain]
//   -> every wave: s_waitcnt vmcnt(0) -> [wave 0: next task ready? then acquire (buffer_inv sc1) and stage its descriptor]
//   -> barrier -> lanes 0-2 bump this task's counters -> the next task starts at once.
// Only when that early look fails (typically: the next task waits for THIS task's counter) does wave 0 spin on the counters
// behind the barrier and a second barrier follows -- round 2 did that for every task, plus a descriptor fetch in between
// (0.8 us look + 0.5 us fetch + a barrier per task, of 4.4 us fixed cost).  Two LDS control blocks alternate so that the next
// task can be staged while the current one's is still being read.
//   ctl block (16 dwords): [0] task index, [1] 0 run / 1 skip / 2 leave, [2] 1 = staged by the early look, [4..15] the task
template <typename T>
__global__ void __launch_bounds__(512, 2) dag_kernel(DagLaunch g) {
  extern __shared__ __align__(16) char smem_raw[];
  int* ctl_base = reinterpret_cast<int*>(smem_raw + DAG_LDS_CTL_OFF);
  const int t = threadIdx.x;
  T* W1 = static_cast<T*>(g.W1);
  T* W2 = static_cast<T*>(g.W2);
  constexpr int TASK_DW = (int)(sizeof(DagTask) / 4);
  static_assert(DAG_MAXWAIT == 4 && offsetof(DagTask, nwait) == 20 && offsetof(DagTask, wcnt) == 28 && offsetof(DagTask, wval) == 36,
                "DagTask dword layout: nwait = low half of dword 5, wcnt = dwords 7-8, wval = dwords 9-10");
  // wave 0, one look at a task's counters and at the evaluation's flag: ONE round trip (lane w reads counter w, lane 8 the flag).
  // desc: the task's descriptor, dword t in lane t.  Returns through *inf the flag, and whether every counter is there.
  auto look_issue = [&](int desc, int* cnt_out, int* need_out, int* inf_out) {
    const unsigned d5 = (unsigned)__builtin_amdgcn_readlane(desc, 5), d7 = (unsigned)__builtin_amdgcn_readlane(desc, 7);
    const unsigned d8 = (unsigned)__builtin_amdgcn_readlane(desc, 8), d9 = (unsigned)__builtin_amdgcn_readlane(desc, 9);
    const unsigned d10 = (unsigned)__builtin_amdgcn_readlane(desc, 10);
    const int nw = (int)(d5 & 0xffffu);
    const unsigned wc2 = (t & 2) ? d8 : d7, wv2 = (t & 2) ? d10 : d9;
    const int my_wcnt = (int)((t & 1) ? wc2 >> 16 : wc2 & 0xffffu), my_wval = (int)((t & 1) ? wv2 >> 16 : wv2 & 0xffffu);
    int c = 0x7fffffff, need = 0, inf = 0;
    if (t < nw) {
      c = __hip_atomic_load(g.ctrl + DAG_CTRL_WORDS + my_wcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      need = my_wval;
    }
    if (t == 8) inf = __hip_atomic_load(g.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *cnt_out = c; *need_out = need; *inf_out = inf;
  };

  int nxt = 0;    // thread 0: queue index of the NEXT task, claimed late in the current task's life (pull hook)
  int desc = 0;   // wave 0, lanes < TASK_DW: descriptor of the task being staged (dword t in lane t)
  int cur_idx = 0;  // wave 0 (uniform): its queue index
  if (t == 0) cur_idx = __hip_atomic_fetch_add(g.ctrl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < 64) {
    cur_idx = __builtin_amdgcn_readfirstlane(cur_idx);
    if (cur_idx < g.ntasks && t < TASK_DW) desc = reinterpret_cast<const int*>(g.tasks + cur_idx)[t];
  }
  int par = 0;          // control block of the task being staged / run
  bool staged = false;  // uniform: the early look of the previous round has staged this task already
  for (;;) {
    int* ctl = ctl_base + 16 * par;
    if (!staged) {
      if (t < 64) {
        // wave 0 waits for the task's dependencies: every look is one round trip; bounded (g.wait_ticks of the 100 MHz clock,
        // then the task is recorded, the flag set and every workgroup drains out).  A flag raised in this very instant may be
        // missed once: the task then computes on data nobody will use, the next task sees it.
        int status = cur_idx >= g.ntasks ? 2 : 0;
        if (g.trace && t == 0 && status == 0) g.trace[(size_t)cur_idx * 5 + 0] = __builtin_amdgcn_s_memrealtime();
        if (status == 0) {
          int inf = 0;
          unsigned long long t0 = 0;
          for (unsigned spins = 0;; ++spins) {
            int c, need;
            look_issue(desc, &c, &need, &inf);
            const bool all_ok = __ballot(c < need) == 0ull;
            inf = __shfl(inf, 8, 64);
            if (all_ok || inf < 0) break;
            if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > g.wait_ticks) {
              if (t == 0) {
                int expected = 0;
                __hip_atomic_compare_exchange_strong(g.ctrl + 1, &expected, cur_idx + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g.info, DAG_INFO_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              inf = DAG_INFO_TIMEOUT;
              break;
            }
          }
          if (inf < 0) status = 2;
          if (status == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's L1 lines: operands were written by other CUs
            status = inf > 0 ? 1 : 0;
          }
        }
        if (t < TASK_DW) ctl[4 + t] = desc;
        if (t == 0) {
          ctl[0] = cur_idx;
          ctl[1] = status;
          ctl[2] = 0;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the barrier below must not open before the invalidate is complete
          if (g.trace && status != 2) g.trace[(size_t)cur_idx * 5 + 1] = __builtin_amdgcn_s_memrealtime();
        }
      }
      __syncthreads();
    }
    const int status = __builtin_amdgcn_readfirstlane(ctl[1]);
    if (status == 2) break;
    // the task, as uniform (scalar) values: dword layout of DagTask
    const int kf = __builtin_amdgcn_readfirstlane(ctl[4]);
    const int kind = kf & 0xffff, flags = (kf >> 16) & 0xffff;
    const int row0 = __builtin_amdgcn_readfirstlane(ctl[5]), col0 = __builtin_amdgcn_readfirstlane(ctl[6]);
    const int kbeg = __builtin_amdgcn_readfirstlane(ctl[7]), kend = __builtin_amdgcn_readfirstlane(ctl[8]);
    const int sig0 = (__builtin_amdgcn_readfirstlane(ctl[9]) >> 16) & 0xffff, sig12 = __builtin_amdgcn_readfirstlane(ctl[10]);
    const int sig1 = sig12 & 0xffff, sig2 = (sig12 >> 16) & 0xffff;
    const int this_idx = __builtin_amdgcn_readfirstlane(ctl[0]);
    // hooks: claim the next queue entry (thread 0) and fetch its descriptor (wave 0); a tile task calls them late in its
    // pipeline, everything else right after its work
    int ndesc = 0, nidx = 0;
    bool pulled = false, fetched = false;
    auto pull = [&]() {
      if (t == 0) nxt = __hip_atomic_fetch_add(g.ctrl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pulled = true;
    };
    auto fetch = [&]() {
      if (t < 64) {
        nidx = __builtin_amdgcn_readfirstlane(nxt);
        if (nidx < g.ntasks && t < TASK_DW) ndesc = reinterpret_cast<const int*>(g.tasks + nidx)[t];
      }
      fetched = true;
    };
#ifdef DAG_STAMP_INNER
    unsigned long long* inner_stamp = g.trace ? g.trace + (size_t)this_idx * 5 + 4 : nullptr;
#else
    unsigned long long* inner_stamp = nullptr;
#endif
    if (status == 0) {
      if (kind == DAG_LEAF) {
        dag_leaf_task<T>(W1, W2, g.ld, row0, static_cast<T*>(g.ldiag), g.info, smem_raw, g.leaf_dbg & 16);  // only the test bit: the timing bits stay with tools/leaf_bench
      } else if ((flags & DAGF_CKINV) && g.Kinv == nullptr) {
        // factorisation-only launch: the K^-1 tiles are not wanted
      } else if (kind == DAG_GEMM_128x64) {
        dag_gemm_tile<T, 128, 64, DAG_PIPE_128x64>(flags, row0, col0, kbeg, kend, W1, W2, static_cast<T*>(g.W3), static_cast<T*>(g.Kinv), g.ld, smem_raw, pull, fetch, inner_stamp);
      } else if (kind == DAG_GEMM_64x64) {
        dag_gemm_tile<T, 64, 64, DAG_PIPE_64x64>(flags, row0, col0, kbeg, kend, W1, W2, static_cast<T*>(g.W3), static_cast<T*>(g.Kinv), g.ld, smem_raw, pull, fetch, inner_stamp);
      } else if (kind == DAG_GEMM_128x128) {
        dag_gemm_tile<T, 128, 128, DAG_PIPE_128x128>(flags, row0, col0, kbeg, kend, W1, W2, static_cast<T*>(g.W3), static_cast<T*>(g.Kinv), g.ld, smem_raw, pull, fetch, inner_stamp);
      } else if (kind == DAG_GEMM_32x64) {
        dag_gemm_tile_chain<T>(flags, row0, col0, kbeg, kend, W1, W2, static_cast<T*>(g.W3), g.ld, smem_raw, pull, fetch);
      }
    }
    if (!pulled) pull();
    if (!fetched) fetch();
    if (g.trace && t == 0) g.trace[(size_t)this_idx * 5 + 2] = __builtin_amdgcn_s_memrealtime();
    // ---- hand-off.  Wave 0 looks at the next task's counters while this task's write-through stores drain.
    int* nctl = ctl_base + 16 * (par ^ 1);
    int look_c = 0, look_need = 0, look_inf = 0;
    const bool have_next = t < 64 && nidx < g.ntasks;  // wave-uniform
    if (have_next) look_issue(ndesc, &look_c, &look_need, &look_inf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores are in the L2 / memory; wave 0: the look has returned
    if (t < 64) {
      int nstatus = 2, early = 1;  // no next entry: the workgroup leaves after this task
      if (have_next) {
        const bool all_ok = __ballot(look_c < look_need) == 0ull;
        const int inf = __shfl(look_inf, 8, 64);
        if (inf < 0) {
          nstatus = 2;
        } else if (all_ok) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          nstatus = inf > 0 ? 1 : 0;
          if (g.trace && t == 0) g.trace[(size_t)nidx * 5 + 0] = __builtin_amdgcn_s_memrealtime();
        } else {
          early = 0;  // wait behind the barrier (the usual case: the next task needs what this one is about to publish)
        }
      }
      if (early) {
        if (t < TASK_DW) nctl[4 + t] = ndesc;
        if (t == 0) {
          nctl[0] = nidx;
          nctl[1] = nstatus;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate is complete before the barrier opens
          if (g.trace && nstatus == 0) g.trace[(size_t)nidx * 5 + 1] = __builtin_amdgcn_s_memrealtime();
        }
      }
      if (t == 0) nctl[2] = early;
      desc = ndesc;
      cur_idx = nidx;
    }
    __syncthreads();
    {
      const int mysig = t == 0 ? sig0 : (t == 1 ? sig1 : sig2);
      if (t < DAG_MAXSIG && mysig != DAG_NOSIG)
        __hip_atomic_fetch_add(g.ctrl + DAG_CTRL_WORDS + mysig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g.trace && t == 0) {
        g.trace[(size_t)this_idx * 5 + 3] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
#ifndef DAG_STAMP_INNER
        g.trace[(size_t)this_idx * 5 + 4] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
#else
        (void)xcc; (void)hwid;
#endif
      }
    }
    par ^= 1;
    staged = __builtin_amdgcn_readfirstlane(ctl_base[16 * par + 2]) != 0;
  }
}

template <typename T>
void launch_dag(const DagLaunch& g, int nwg, hipStream_t s) {
  if (g.ntasks <= 0 || nwg <= 0) return;
  hipLaunchKernelGGL((dag_kernel<T>), dim3(nwg), dim3(512), DAG_LDS_BYTES, s, g);
}
template void launch_dag<double>(const DagLaunch&, int, hipStream_t);
template void launch_dag<float>(const DagLaunch&, int, hipStream_t);

int dag_stage_depth(bool is_f32) { return is_f32 ? DagGeom<float, 128, 64>::BK : DagGeom<double, 128, 64>::BK; }

static void init_dag_kernels() {
  set_lds_attr(reinterpret_cast<const void*>(&dag_kernel<double>), DAG_LDS_BYTES, "dag_kernel<f64>: dynamic LDS limit");
  set_lds_attr(reinterpret_cast<const void*>(&dag_kernel<float>), DAG_LDS_BYTES, "dag_kernel<f32>: dynamic LDS limit");
}
