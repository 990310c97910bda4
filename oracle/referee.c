/* referee.c -- an extended-precision REFEREE for the ill-conditioned end of a fit.  TEST INFRASTRUCTURE ONLY.
 *
 * Where the optimiser ends (cond(K) ~ 1e11..1e12 for config M) the GPU engine and the LAPACK oracle (oracle/gpr_oracle.py) differ
 * by ~cond(K) eps, and neither is "the truth": both solve with a kernel matrix rounded to f64.  This file supplies the truth for
 * the quantities the reference's predict path exposes (predict.rs:18-37: mean = K* alpha, var = diag + 1e-5 - k*^T K^-1 k*,
 * with alpha = K^-1 y, lml.rs:54):
 *   - the kernel matrix K = c Matern(|x_i - x_j| / ell) + noise I (matern_kernel.rs:37-81, product_kernel.rs:37, lml.rs:44) is
 *     evaluated in IEEE binary128 (__float128: 113-bit significand, libquadmath's sqrtq / expq) from the f64 inputs taken as
 *     exact, and kept as double-double pairs (hi + lo, 106 bits);
 *   - linear systems are solved by mixed-precision iterative refinement (oracle/referee.py): the f64 LAPACK factor of hi(K) is the
 *     preconditioner, the residual b - K x is formed HERE in double-double arithmetic with x carried as a double-double vector,
 *     so the iteration converges to the solution of the extended-precision system (error ~ cond(K) 1e-30, not cond(K) 1e-16).
 * Nothing under hbetune_rs_amd/ or csrc/ uses this; only tests/ and bench.py's parity leg do.
 *
 * gcc -O2 -fPIC -shared -fopenmp -mfma -ffp-contract=off -o oracle/libreferee.so oracle/referee.c -lquadmath -lm
 */
#include <math.h>
#include <quadmath.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double hi, lo; } dd;

/* error-free transformations (-ffp-contract=off: the compiler must not fuse these) */
static inline dd two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  dd r = {s, (a - (s - bb)) + (b - bb)};
  return r;
}
static inline dd fast_two_sum(double a, double b) { /* |a| >= |b| */
  const double s = a + b;
  dd r = {s, b - (s - a)};
  return r;
}
static inline dd two_prod(double a, double b) {
  const double p = a * b;
#ifdef __FMA__
  dd r = {p, __builtin_fma(a, b, -p)};
#else
  /* Dekker / Veltkamp split */
  const double sp = 134217729.0; /* 2^27 + 1 */
  double t = sp * a, ah = t - (t - a), al = a - ah;
  t = sp * b;
  double bh = t - (t - b), bl = b - bh;
  dd r = {p, ((ah * bh - p) + ah * bl + al * bh) + al * bl};
#endif
  return r;
}
static inline dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi), t = two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = fast_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return fast_two_sum(s.hi, s.lo);
}
static inline dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return fast_two_sum(p.hi, p.lo);
}
static inline dd dd_neg(dd a) { dd r = {-a.hi, -a.lo}; return r; }
static inline dd from_q(__float128 q) {
  dd r;
  r.hi = (double)q;
  r.lo = (double)(q - (__float128)r.hi);
  return r;
}

/* one kernel entry in binary128: nu2 = 2 nu in {1, 3, 5}, 0 = squared exponential (the extension of oracle/gpr_oracle.py) */
static __float128 kernel_q(const double* xa, const double* xb, int d, const double* ell, double amp, int nu2) {
  __float128 s = 0;
  for (int k = 0; k < d; ++k) {
    const __float128 t = ((__float128)xa[k] - (__float128)xb[k]) / (__float128)ell[k];
    s += t * t;
  }
  const __float128 r = sqrtq(s);
  __float128 m;
  if (nu2 == 0) m = expq(-s / 2);
  else if (nu2 == 1) m = expq(-r);
  else if (nu2 == 3) { const __float128 k3 = r * sqrtq((__float128)3); m = (1 + k3) * expq(-k3); }
  else { const __float128 k5 = r * sqrtq((__float128)5); m = (1 + k5 + k5 * k5 / 3) * expq(-k5); }
  return (__float128)amp * m;
}

typedef struct {
  int n, d, nu2;
  double amp, noise;
  double* ell;
  double* X;       /* n x d */
  double* khi;     /* n x n, full symmetric */
  double* klo;
} referee;

void* referee_create(const double* X, int n, int d, double noise, double amp, const double* ell, int nu2) {
  referee* h = (referee*)calloc(1, sizeof(referee));
  if (!h) return NULL;
  h->n = n; h->d = d; h->nu2 = nu2; h->amp = amp; h->noise = noise;
  h->ell = (double*)malloc(sizeof(double) * d);
  h->X = (double*)malloc(sizeof(double) * (size_t)n * d);
  h->khi = (double*)malloc(sizeof(double) * (size_t)n * n);
  h->klo = (double*)malloc(sizeof(double) * (size_t)n * n);
  if (!h->ell || !h->X || !h->khi || !h->klo) return NULL;
  memcpy(h->ell, ell, sizeof(double) * d);
  memcpy(h->X, X, sizeof(double) * (size_t)n * d);
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      __float128 q = kernel_q(X + (size_t)i * d, X + (size_t)j * d, d, ell, amp, nu2);
      if (i == j) q += (__float128)noise;
      const dd v = from_q(q);
      h->khi[(size_t)i * n + j] = v.hi; h->klo[(size_t)i * n + j] = v.lo;
      h->khi[(size_t)j * n + i] = v.hi; h->klo[(size_t)j * n + i] = v.lo;
    }
  return h;
}

void referee_destroy(void* hv) {
  referee* h = (referee*)hv;
  if (!h) return;
  free(h->ell); free(h->X); free(h->khi); free(h->klo); free(h);
}

/* hi(K): what an f64 LAPACK factorisation (the preconditioner) sees */
void referee_khi(void* hv, double* out) {
  referee* h = (referee*)hv;
  memcpy(out, h->khi, sizeof(double) * (size_t)h->n * h->n);
}

/* R[:, q] = B[:, q] - K X[:, q] in double-double, rounded to f64 at the end.  All arrays are [n][nrhs] row-major; blo may be NULL. */
void referee_residual(void* hv, int nrhs, const double* bhi, const double* blo, const double* xhi, const double* xlo, double* r) {
  referee* h = (referee*)hv;
  const int n = h->n;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const double* kh = h->khi + (size_t)i * n;
    const double* kl = h->klo + (size_t)i * n;
    for (int q = 0; q < nrhs; ++q) {
      dd acc = {bhi[(size_t)i * nrhs + q], blo ? blo[(size_t)i * nrhs + q] : 0.0};
      for (int j = 0; j < n; ++j) {
        const dd kij = {kh[j], kl[j]}, xj = {xhi[(size_t)j * nrhs + q], xlo[(size_t)j * nrhs + q]};
        acc = dd_add(acc, dd_neg(dd_mul(kij, xj)));
      }
      r[(size_t)i * nrhs + q] = acc.hi + acc.lo;
    }
  }
}

/* cross-kernel rows k*(xs_q) (predict.rs:18) in binary128, as double-double pairs: out[n][m] row-major (column q = candidate q) */
void referee_kstar(void* hv, const double* Xs, int m, double* khi, double* klo) {
  referee* h = (referee*)hv;
  const int n = h->n, d = h->d;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i)
    for (int q = 0; q < m; ++q) {
      const dd v = from_q(kernel_q(Xs + (size_t)q * d, h->X + (size_t)i * d, d, h->ell, h->amp, h->nu2));
      khi[(size_t)i * m + q] = v.hi; klo[(size_t)i * m + q] = v.lo;
    }
}

/* out[q] = sum_i A[i][q] * B[i][q] in double-double (columns of two [n][m] double-double arrays), rounded to f64 */
void referee_coldot(int n, int m, const double* ahi, const double* alo, const double* bhi, const double* blo, double* out_hi, double* out_lo) {
  for (int q = 0; q < m; ++q) {
    dd acc = {0.0, 0.0};
    for (int i = 0; i < n; ++i) {
      const dd a = {ahi[(size_t)i * m + q], alo[(size_t)i * m + q]}, b = {bhi[(size_t)i * m + q], blo[(size_t)i * m + q]};
      acc = dd_add(acc, dd_mul(a, b));
    }
    out_hi[q] = acc.hi; out_lo[q] = acc.lo;
  }
}

/* x (double-double, [n][m]) += d (f64): the refinement's update, error-free in the hi/lo pair */
void referee_axpy(int count, double* xhi, double* xlo, const double* dlt) {
  for (int i = 0; i < count; ++i) {
    const dd x = {xhi[i], xlo[i]}, dv = {dlt[i], 0.0};
    const dd s = dd_add(x, dv);
    xhi[i] = s.hi; xlo[i] = s.lo;
  }
}

/* log det K in extended precision: double-double Cholesky (Cholesky-Crout, column by column: every entry is a dot product of two
 * contiguous row prefixes), then 2 sum log L_jj with the logarithms in binary128.  n^3/3 double-double multiply-adds.
 * Returns 0, or 1 + j when pivot j is not positive.  out = {hi, lo}.  lml.rs:57-59 needs sum_i log L_ii = logdet / 2. */
int referee_logdet(void* hv, double* out) {
  referee* h = (referee*)hv;
  const int n = h->n;
  dd* L = (dd*)malloc(sizeof(dd) * (size_t)n * n);
  if (!L) return -1;
  int bad = 0;
  for (int j = 0; j < n && !bad; ++j) {
    /* diagonal */
    dd s = {h->khi[(size_t)j * n + j], h->klo[(size_t)j * n + j]};
    const dd* rj = L + (size_t)j * n;
    for (int k = 0; k < j; ++k) s = dd_add(s, dd_neg(dd_mul(rj[k], rj[k])));
    if (!(s.hi > 0)) { bad = 1 + j; break; }
    const __float128 piv = sqrtq((__float128)s.hi + (__float128)s.lo);
    const dd ljj = from_q(piv), inv = from_q(1 / piv);
    L[(size_t)j * n + j] = ljj;
#pragma omp parallel for schedule(static)
    for (int i = j + 1; i < n; ++i) {
      dd t = {h->khi[(size_t)i * n + j], h->klo[(size_t)i * n + j]};
      const dd* ri = L + (size_t)i * n;
      for (int k = 0; k < j; ++k) t = dd_add(t, dd_neg(dd_mul(ri[k], rj[k])));
      L[(size_t)i * n + j] = dd_mul(t, inv);
    }
  }
  if (!bad) {
    __float128 acc = 0;
    for (int j = 0; j < n; ++j) acc += logq((__float128)L[(size_t)j * n + j].hi + (__float128)L[(size_t)j * n + j].lo);
    const dd v = from_q(2 * acc);
    out[0] = v.hi; out[1] = v.lo;
  }
  free(L);
  return bad;
}

/* Gradient of the lml in binary128: g_j = 1/2 sum_ik (alpha_i alpha_k - Kinv_ik) dK_ik/dtheta_j (lml.rs:62-70), theta order
 * [noise, amplitude, ell_1..ell_d]; noise gradient = noise * I (lml.rs:41), amplitude: c * Matern (constant_kernel.rs:31-38 x
 * product_kernel.rs:56-67), length scales: matern_kernel.rs:88-131.  alpha and the FULL symmetric K^-1 come in as double-double
 * pairs ([n] and [n][n]); out_hi / out_lo: d + 2 entries. */
void referee_gradient(void* hv, const double* ahi, const double* alo, const double* vhi, const double* vlo, double* out_hi, double* out_lo) {
  referee* h = (referee*)hv;
  const int n = h->n, d = h->d, p = d + 2, nu2 = h->nu2;
  __float128* tot = (__float128*)calloc((size_t)p, sizeof(__float128));
#pragma omp parallel
  {
    __float128* acc = (__float128*)calloc((size_t)p, sizeof(__float128));
    __float128* dk = (__float128*)malloc(sizeof(__float128) * (size_t)d);
#pragma omp for schedule(dynamic, 4)
    for (int i = 0; i < n; ++i) {
      const __float128 ai = (__float128)ahi[i] + (__float128)alo[i];
      for (int j = 0; j < n; ++j) {
        const __float128 aj = (__float128)ahi[j] + (__float128)alo[j];
        const __float128 w = ai * aj - ((__float128)vhi[(size_t)i * n + j] + (__float128)vlo[(size_t)i * n + j]);
        __float128 s = 0;
        for (int k = 0; k < d; ++k) {
          const __float128 t = ((__float128)h->X[(size_t)i * d + k] - (__float128)h->X[(size_t)j * d + k]) / (__float128)h->ell[k];
          dk[k] = t * t;
          s += dk[k];
        }
        const __float128 r = sqrtq(s);
        __float128 km, gr;
        if (nu2 == 0) { km = expq(-s / 2); gr = km; }
        else if (nu2 == 1) { km = expq(-r); gr = (r > 0) ? km / r : 0; }
        else if (nu2 == 3) { const __float128 t3 = r * sqrtq((__float128)3), e = expq(-t3); km = (1 + t3) * e; gr = 3 * e; }
        else { const __float128 t5 = r * sqrtq((__float128)5), e = expq(-t5); km = (1 + t5 + t5 * t5 / 3) * e; gr = ((__float128)5 / 3) * (1 + t5) * e; }
        if (i == j) acc[0] += w * (__float128)h->noise;
        acc[1] += w * ((__float128)h->amp * km);
        const __float128 cg = w * (__float128)h->amp * gr;
        for (int k = 0; k < d; ++k) acc[2 + k] += cg * dk[k];
      }
    }
#pragma omp critical
    for (int q = 0; q < p; ++q) tot[q] += acc[q];
    free(acc); free(dk);
  }
  for (int q = 0; q < p; ++q) {
    const dd v = from_q(tot[q] / 2);
    out_hi[q] = v.hi; out_lo[q] = v.lo;
  }
  free(tot);
}
