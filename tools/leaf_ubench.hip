// leaf_ubench.hip -- cycle costs of the primitives the 128x128 diagonal block is built from, one wave on one SIMD
// (s_memtime stamps around unrolled sequences; every result is kept live).  Grounds the per-pivot cycle budget in DESIGN.md.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/leaf_ubench tools/leaf_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long now() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long v = (long long)__builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return v;
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
#define FENCE() asm volatile("" ::: "memory")
#define KEEP(x) asm volatile("" : "+v"(x))


template <int J> __device__ __forceinline__ void fmac_nbc(double& d, double s0, double s1) {
  asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s0), "v"(s1), "n"(J));
}
template <int K> __device__ __forceinline__ double row_bcast_safe(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
__device__ __forceinline__ double mul_then_gap(double a, double b) {
  double r;
  asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int K, int J> struct LeafCol {
  static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16], double lka, double lkb) {
    fmac_nbc<J>(a[J], lka, lka); fmac_nbc<J>(b[J], lka, lkb); LeafCol<K, J + 1>::run(a, b, lka, lkb);
  }
};
template <int K> struct LeafCol<K, 16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double, double) {} };
template <int K> struct LeafPivot {
  static __device__ __forceinline__ void run(double (&a)[16], double (&b)[16]) {
    const double piv = row_bcast_safe<K>(a[K]);
    double rinv = __builtin_amdgcn_rsq(piv);
    { const double gg = piv * rinv; const double ee = __builtin_fma(-gg, rinv, 1.0); const double pp = __builtin_fma(ee, 0.375, 0.5); rinv = __builtin_fma(rinv, ee * pp, rinv); }
    const double lka = mul_then_gap(a[K], rinv);
    const double lkb = b[K] * rinv;
    a[K] = lka; b[K] = lkb;
    LeafCol<K, K + 1>::run(a, b, lka, lkb);
    LeafPivot<K + 1>::run(a, b);
  }
};
template <> struct LeafPivot<16> { static __device__ __forceinline__ void run(double (&)[16], double (&)[16]) {} };
template <int J> struct DppRun { static __device__ __forceinline__ void run(double (&a)[16], double lk, double m) { fmac_nbc<J>(a[J], lk, m); DppRun<J + 1>::run(a, lk, m); } };
template <> struct DppRun<16> { static __device__ __forceinline__ void run(double (&)[16], double, double) {} };

constexpr int NT = 28;  // number of tests
__global__ void __launch_bounds__(512) k(long long* out, double* sink, double seed, int nwaves_active) {
  __shared__ double lds[4096];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  long long res[NT];
  for (int i = 0; i < NT; ++i) res[i] = 0;
  double x = seed + lane * 1e-3, y = 1.0 + lane * 1e-4;
  if (wave == 0) {
    long long t0, t1;
    // 0: 64 dependent v_fma_f64
    { double a = x; KEEP(a); t0 = now();
#pragma unroll
      for (int i = 0; i < 64; ++i) { a = __builtin_fma(a, y, 1e-9); KEEP(a); }
      t1 = now(); res[0] = t1 - t0; x += a * 1e-30; }
    // 1: 64 independent v_fma_f64 (8 chains x 8)
    { double a[8]; for (int j = 0; j < 8; ++j) { a[j] = x + j; KEEP(a[j]); } t0 = now();
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = __builtin_fma(a[j], y, 1e-9); KEEP(a[j]); }
      t1 = now(); res[1] = t1 - t0; for (int j = 0; j < 8; ++j) x += a[j] * 1e-30; }
    // 2: 16 dependent v_rsq_f64
    { double a = x + 2.0; KEEP(a); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) { a = __builtin_amdgcn_rsq(a); KEEP(a); }
      t1 = now(); res[2] = t1 - t0; x += a * 1e-30; }
    // 3: 16 independent v_rsq_f64
    { double a[16]; for (int j = 0; j < 16; ++j) { a[j] = x + j + 2.0; KEEP(a[j]); } t0 = now();
#pragma unroll
      for (int j = 0; j < 16; ++j) { a[j] = __builtin_amdgcn_rsq(a[j]); KEEP(a[j]); }
      t1 = now(); res[3] = t1 - t0; for (int j = 0; j < 16; ++j) x += a[j] * 1e-30; }
    // 4: 16 x (readlane pair + fma), independent targets, one source register (the in-panel trailing update)
    { double a[16], lk = x; for (int j = 0; j < 16; ++j) { a[j] = x + j; KEEP(a[j]); } KEEP(lk); t0 = now();
#pragma unroll
      for (int j = 0; j < 16; ++j) { a[j] = __builtin_fma(-lk, readlane_d(lk, j), a[j]); KEEP(a[j]); }
      t1 = now(); res[4] = t1 - t0; for (int j = 0; j < 16; ++j) x += a[j] * 1e-30; }
    // 5: one pivot step of the current elimination: readlane(piv) -> rsq -> 4 refinement ops -> mul -> then next pivot's fma (dependent), x16
    { double a = x + 3.0; KEEP(a); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double piv = readlane_d(a, i);
        double r = __builtin_amdgcn_rsq(piv);
        const double g = piv * r; const double e = __builtin_fma(-g, r, 1.0); const double pp = __builtin_fma(e, 0.375, 0.5);
        r = __builtin_fma(r, e * pp, r);
        const double l = a * r;
        a = __builtin_fma(-l, readlane_d(l, (i + 1) & 63), a + 4.0); KEEP(a);
      }
      t1 = now(); res[5] = t1 - t0; x += a * 1e-30; }
    // 6: 16 dependent mfma f64 16x16x4 (accumulator chain)
    { d4 acc = {0, 0, 0, 0}; double a = x, b = y; KEEP(a); KEEP(b); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      double s = acc[0] + acc[1] + acc[2] + acc[3]; KEEP(s);
      t1 = now(); res[6] = t1 - t0; x += s * 1e-30; }
    // 7: 16 independent mfma (4 accumulators x 4)
    { d4 acc[4]; for (int j = 0; j < 4; ++j) acc[j] = d4{0, 0, 0, 0}; double a = x, b = y; KEEP(a); KEEP(b); t0 = now();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
      double s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3]; KEEP(s);
      t1 = now(); res[7] = t1 - t0; x += s * 1e-30; }
    // 8: mfma whose A operand is the previous result's register (the micro-panel chain: result -> fragment), x8
    { d4 acc = {x, y, x, y}; t0 = now();
#pragma unroll
      for (int i = 0; i < 8; ++i) { d4 z = {0, 0, 0, 0}; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], acc[1], z, 0, 0, 0); }
      double s = acc[0] + acc[1]; KEEP(s);
      t1 = now(); res[8] = t1 - t0; x += s * 1e-30; }
    // 9: LDS round trip: ds_write_b64 -> wave fence -> ds_read_b64 (dependent), x16
    { double a = x; KEEP(a); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        lds[lane + 64 * (i & 7)] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        a = lds[(lane ^ 1) + 64 * (i & 7)] + 1.0; KEEP(a);
      }
      t1 = now(); res[9] = t1 - t0; x += a * 1e-30; }
    // 10: 32 x v_cndmask_b32 (16 f64 selects), independent
    { double a[16]; for (int j = 0; j < 16; ++j) { a[j] = x + j; KEEP(a[j]); } const bool c = (lane & 3) == 1; t0 = now();
#pragma unroll
      for (int j = 0; j < 16; ++j) { a[j] = c ? y : a[j]; KEEP(a[j]); }
      t1 = now(); res[10] = t1 - t0; for (int j = 0; j < 16; ++j) x += a[j] * 1e-30; }
    // 11: 16 x ds_bpermute pair (f64 gather), dependent chain
    { double a = x; KEEP(a); const int idx = ((lane + 17) & 63) << 2; t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int lo = __double2loint(a), hi = __double2hiint(a);
        lo = __builtin_amdgcn_ds_bpermute(idx, lo); hi = __builtin_amdgcn_ds_bpermute(idx, hi);
        a = __hiloint2double(hi, lo) + 1.0; KEEP(a);
      }
      t1 = now(); res[11] = t1 - t0; x += a * 1e-30; }
    // 12: 16 x (v_permlane32_swap pair): f64 half-swap, dependent
    { double a = x, b = y; KEEP(a); KEEP(b); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(alo), "+v"(blo));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ahi), "+v"(bhi));
        a = __hiloint2double(ahi, alo); b = __hiloint2double(bhi, blo);
      }
      t1 = now(); res[12] = t1 - t0; x += (a + b) * 1e-30; }
    // 13: 16 x (v_permlane16_swap pair)
    { double a = x, b = y; KEEP(a); KEEP(b); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(alo), "+v"(blo));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(ahi), "+v"(bhi));
        a = __hiloint2double(ahi, alo); b = __hiloint2double(bhi, blo);
      }
      t1 = now(); res[13] = t1 - t0; x += (a + b) * 1e-30; }
    // 14: 16 dependent v_mul_f64 with an SGPR (uniform) operand from readlane: readlane pair -> mul -> readlane ...
    { double a = x + 1.0; KEEP(a); t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) { const double s = readlane_d(a, i); a = a * s + 0.5; KEEP(a); }
      t1 = now(); res[14] = t1 - t0; x += a * 1e-30; }
    // 15: 64 dependent v_fma_f32
    { float a = (float)x; KEEP(a); const float yf = (float)y; t0 = now();
#pragma unroll
      for (int i = 0; i < 64; ++i) { a = __builtin_fmaf(a, yf, 1e-9f); KEEP(a); }
      t1 = now(); res[15] = t1 - t0; x += a * 1e-30; }
    // 16: ds_read_b64 uniform-address broadcast x16 dependent on a prior ds_write (write once, then 16 independent reads)
    { double a = x; KEEP(a); lds[1024 + lane] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double r[16]; t0 = now();
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = lds[1024 + j];
      double s = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += r[j]; KEEP(s);
      t1 = now(); res[16] = t1 - t0; x += s * 1e-30; }
    // 17: empty stamp pair
    { t0 = now(); FENCE(); t1 = now(); res[17] = t1 - t0; }
    // 18: mfma -> dependent VALU read of the result -> mfma (result feeds a v_mul before the next mfma), x8
    { d4 acc = {x, y, x, y}; t0 = now();
#pragma unroll
      for (int i = 0; i < 8; ++i) { d4 z = {0, 0, 0, 0}; const double f = acc[0] * 1.000001; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, z, 0, 0, 0); }
      double s = acc[0] + acc[1]; KEEP(s);
      t1 = now(); res[18] = t1 - t0; x += s * 1e-30; }

    // 22: 64 x v_fmac_f64_dpp row_newbcast (16 independent targets x 4)
    { double a[16], lk = x; for (int j = 0; j < 16; ++j) { a[j] = x + j; KEEP(a[j]); } KEEP(lk); t0 = now();
      DppRun<0>::run(a, lk, y); DppRun<0>::run(a, lk, y); DppRun<0>::run(a, lk, y); DppRun<0>::run(a, lk, y);
      for (int j = 0; j < 16; ++j) KEEP(a[j]);
      t1 = now(); res[22] = t1 - t0; for (int j = 0; j < 16; ++j) x += a[j] * 1e-30; }
    // 23: one whole 16-column panel elimination (LeafPivot<0>) on an SPD-ish panel in registers
    { double a[16], b[16];
      for (int j = 0; j < 16; ++j) { a[j] = ((lane & 15) == j ? 20.0 : 0.0) + 1.0 / (1.0 + ((lane & 15) > j ? (lane & 15) - j : j - (lane & 15))); b[j] = 0.3 + 0.01 * j + 1e-3 * lane; KEEP(a[j]); KEEP(b[j]); }
      t0 = now();
      LeafPivot<0>::run(a, b);
      for (int j = 0; j < 16; ++j) { KEEP(a[j]); KEEP(b[j]); }
      t1 = now(); res[23] = t1 - t0; for (int j = 0; j < 16; ++j) x += (a[j] + b[j]) * 1e-30; }
  }
  // 19/20: s_barrier with all 8 waves arriving together, x16 (measured by wave 0)
  {
    __syncthreads();
    long long t0 = now();
#pragma unroll
    for (int i = 0; i < 16; ++i) __syncthreads();
    long long t1 = now();
    if (wave == 0) res[19] = t1 - t0;
  }
  // 21: wave 0 writes LDS, others spin on an LDS flag: cross-wave hand-off latency (wave 1 stamps when it sees the flag)
  {
    volatile int* flag = reinterpret_cast<volatile int*>(&lds[2048]);
    volatile long long* st = reinterpret_cast<volatile long long*>(&lds[2050]);
    if (t == 0) { flag[0] = 0; }
    __syncthreads();
    if (wave == 0) {
      __builtin_amdgcn_s_sleep(20);
      if (lane == 0) { st[0] = now(); flag[0] = 1; }
    } else if (wave == 1) {
      while (flag[0] == 0) {}
      if (lane == 0) st[1] = now();
    }
    __syncthreads();
    if (t == 0) res[21] = st[1] - st[0];
  }
  if (wave == 0 && lane == 0) for (int i = 0; i < NT; ++i) out[i] = res[i];
  sink[t] = x;
}

int main() {
  long long* out; double* sink;
  CK(hipMalloc(&out, sizeof(long long) * NT)); CK(hipMalloc(&sink, sizeof(double) * 512));
  long long h[NT];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, sink, 1.5, 8);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  const long long base = h[17];
  auto per = [&](int i, int n) { return (double)(h[i] - base) / n; };
  printf("stamp pair overhead: %lld cycles (subtracted below)\n", base);
  printf(" 0 v_fma_f64 dependent:                 %.1f cyc each\n", per(0, 64));
  printf(" 1 v_fma_f64 independent:               %.1f cyc each\n", per(1, 64));
  printf(" 2 v_rsq_f64 dependent:                 %.1f cyc each\n", per(2, 16));
  printf(" 3 v_rsq_f64 independent:               %.1f cyc each\n", per(3, 16));
  printf(" 4 readlane pair + fma (independent):   %.1f cyc per column update\n", per(4, 16));
  printf(" 5 pivot step (readlane,rsq,refine,mul,readlane,fma) dependent: %.1f cyc per pivot\n", per(5, 16));
  printf(" 6 mfma_f64_16x16x4 dependent acc:      %.1f cyc each\n", per(6, 16));
  printf(" 7 mfma_f64_16x16x4 independent (4 acc): %.1f cyc each\n", per(7, 16));
  printf(" 8 mfma result -> A/B operand of next:  %.1f cyc each\n", per(8, 8));
  printf(" 9 LDS write -> read round trip:        %.1f cyc each\n", per(9, 16));
  printf("10 f64 select (2 v_cndmask_b32):        %.1f cyc each\n", per(10, 16));
  printf("11 f64 ds_bpermute pair (dependent):    %.1f cyc each\n", per(11, 16));
  printf("12 f64 v_permlane32_swap pair:          %.1f cyc each\n", per(12, 16));
  printf("13 f64 v_permlane16_swap pair:          %.1f cyc each\n", per(13, 16));
  printf("14 readlane pair -> fma dependent:      %.1f cyc each\n", per(14, 16));
  printf("15 v_fma_f32 dependent:                 %.1f cyc each\n", per(15, 64));
  printf("16 16 uniform-address ds_read_b64 + sum: %.1f cyc total\n", per(16, 1));
  printf("22 v_fmac_f64_dpp row_newbcast (indep):  %.1f cyc each\n", per(22, 64));
  printf("23 LeafPivot<0> whole panel, one wave:  %.1f cyc total\n", per(23, 1));
  printf("18 mfma -> v_mul -> mfma:               %.1f cyc per pair\n", per(18, 8));
  printf("19 s_barrier, 8 waves in step:          %.1f cyc each\n", (double)h[19] / 16);
  printf("21 LDS flag hand-off wave0 -> wave1:    %lld cyc\n", h[21]);
  return 0;
}
