/*
 * gpr_oracle.c — plain-C CPU restatement of hbetune's src/gpr path.  TEST INFRASTRUCTURE ONLY.
 *
 * Second, LAPACK-free oracle next to oracle/gpr_oracle.py: every loop is written out, so parity of the GPU engine does
 * not rest on any BLAS.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product library (libhbegp.so) never links or calls it.
 *
 * Pinning: same as gpr_oracle.py (reference known-answer tables + scikit-learn golden vectors under tests/golden/);
 * tests/test_oracle_golden.py checks this file against both.  The reference's Cholesky / solve / inverse are LAPACK
 * ?potrf / ?potrs / ?potri (ndarray-linalg 0.12.0 -> lapacke 0.2.0 -> openblas-src 0.7.0, not vendored); the routines
 * below restate LAPACK's published unblocked algorithms (dpotf2, dtrtri/dtrti2, dlauum/dlauu2, forward/back substitution).
 *
 * All citations are file:line in the reference tree.  f64 only (the f32 path is covered by gpr_oracle.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, n) ((size_t)(i) * (size_t)(n) + (size_t)(j))

/* matern_kernel.rs:262-283 (cdist) on length-scaled inputs (matern_kernel.rs:50-60) + map (matern_kernel.rs:65-80). */
static double matern_from_scaled(const double* xa, const double* xb, int d, int nu2) {
  double accum = 0.0;
  for (int i = 0; i < d; ++i) {
    double df = xa[i] - xb[i];
    accum += df * df; /* powi(2) */
  }
  double dist = sqrt(accum);
  if (nu2 == 0) return exp(-0.5 * accum); /* squared exponential (nu = infinity): extension, not in the reference */
  if (nu2 == 1) return exp(-dist);
  if (nu2 == 3) {
    double k = dist * sqrt(3.0);
    return (k + 1.0) * exp(-k);
  }
  double k = dist * sqrt(5.0);
  return (1.0 + k + k * k / 3.0) * exp(-k);
}

/* product_kernel.rs:36-38 with constant_kernel.rs:24-29:  K = c * Matern(x1, x2).  out is n1 x n2 row-major. */
void oracle_kernel(const double* x1, int n1, const double* x2, int n2, int d, int nu2, double amp, const double* ell,
                   double* out) {
  double* s1 = (double*)malloc(sizeof(double) * (size_t)n1 * d);
  double* s2 = (double*)malloc(sizeof(double) * (size_t)n2 * d);
  for (int i = 0; i < n1; ++i)
    for (int k = 0; k < d; ++k) s1[IDX(i, k, d)] = x1[IDX(i, k, d)] / ell[k];
  for (int i = 0; i < n2; ++i)
    for (int k = 0; k < d; ++k) s2[IDX(i, k, d)] = x2[IDX(i, k, d)] / ell[k];
  for (int i = 0; i < n1; ++i)
    for (int j = 0; j < n2; ++j) out[IDX(i, j, n2)] = amp * matern_from_scaled(s1 + (size_t)i * d, s2 + (size_t)j * d, d, nu2);
  free(s1);
  free(s2);
}

/* LAPACK dpotf2 (lower), row-major in place.  Returns 0, or j+1 when the leading minor of order j+1 is not positive. */
static int potf2_lower(double* a, int n) {
  for (int j = 0; j < n; ++j) {
    double ajj = a[IDX(j, j, n)];
    for (int k = 0; k < j; ++k) ajj -= a[IDX(j, k, n)] * a[IDX(j, k, n)];
    if (!(ajj > 0.0)) return j + 1; /* also catches NaN, as dpotf2's disnan test does */
    ajj = sqrt(ajj);
    a[IDX(j, j, n)] = ajj;
    for (int i = j + 1; i < n; ++i) {
      double s = a[IDX(i, j, n)];
      for (int k = 0; k < j; ++k) s -= a[IDX(i, k, n)] * a[IDX(j, k, n)];
      a[IDX(i, j, n)] = s / ajj;
    }
  }
  return 0;
}

/* dpotrs: solve L L^T x = b. */
static void potrs_lower(const double* l, int n, const double* b, double* x) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= l[IDX(i, k, n)] * x[k];
    x[i] = s / l[IDX(i, i, n)];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= l[IDX(k, i, n)] * x[k];
    x[i] = s / l[IDX(i, i, n)];
  }
}

/* dpotri = dtrtri (inverse of L) followed by dlauum (L^-T L^-1); returns the full symmetric matrix like invc(). */
static void potri_lower(const double* l, int n, double* inv) {
  double* x = (double*)calloc((size_t)n * n, sizeof(double));
  for (int j = 0; j < n; ++j) { /* column j of X = L^-1 by forward substitution */
    x[IDX(j, j, n)] = 1.0 / l[IDX(j, j, n)];
    for (int i = j + 1; i < n; ++i) {
      double s = 0.0;
      for (int k = j; k < i; ++k) s -= l[IDX(i, k, n)] * x[IDX(k, j, n)];
      x[IDX(i, j, n)] = s / l[IDX(i, i, n)];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = i; k < n; ++k) s += x[IDX(k, i, n)] * x[IDX(k, j, n)];
      inv[IDX(i, j, n)] = s;
      inv[IDX(j, i, n)] = s;
    }
  free(x);
}

/*
 * lml.rs:29-79  lml_with_gradient.  theta-gradient order [noise, amplitude, ell_1..ell_d] (lml.rs:67-68).
 * Materialises the n x n x (d+1) kernel-gradient tensor exactly like the reference (matern_kernel.rs:83-135,
 * product_kernel.rs:40-70).  Returns 0, or 1 when the factorisation fails (lml.rs:47-50).
 * Outputs (any may be NULL): lml, grad[d+2], alpha[n], kinv[n*n], kmat[n*n], ldiag[n].
 */
int oracle_lml_with_gradient(const double* x, const double* y, int n, int d, int nu2, double noise, double amp,
                             const double* ell, double* lml_out, double* grad, double* alpha_out, double* kinv_out,
                             double* kmat_out, double* ldiag_out) {
  const size_t nn = (size_t)n * n;
  double* kmat = (double*)malloc(sizeof(double) * nn);
  double* dk = (double*)malloc(sizeof(double) * nn * (size_t)(d + 1)); /* [i][j][param], param 0 = amplitude */
  oracle_kernel(x, n, x, n, d, nu2, amp, ell, kmat); /* kernel.theta_grad -> kernel (matern_kernel.rs:84) */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double* g = dk + IDX(i, j, n) * (size_t)(d + 1);
      double dsum = 0.0;
      for (int k = 0; k < d; ++k) { /* matern_kernel.rs:88-98 */
        double df = x[IDX(i, k, d)] - x[IDX(j, k, d)];
        g[1 + k] = df * df / (ell[k] * ell[k]);
        dsum += g[1 + k];
      }
      double km = kmat[IDX(i, j, n)] / amp; /* Matern value (k2) */
      for (int k = 0; k < d; ++k) {
        double gm;
        if (nu2 == 1) { /* matern_kernel.rs:102-111 */
          gm = km * g[1 + k] / sqrt(dsum);
          if (!isfinite(gm)) gm = 0.0;
        } else if (nu2 == 3) { /* :112-118 */
          gm = g[1 + k] * exp(-sqrt(dsum * 3.0)) * 3.0;
        } else if (nu2 == 0) { /* squared exponential: dK/dlog(ell_k) = K d_k */
          gm = km * g[1 + k];
        } else { /* :119-131 */
          double tmp = sqrt(dsum * 5.0);
          gm = exp(-tmp) * (tmp + 1.0) * g[1 + k] * (5.0 / 3.0);
        }
        g[1 + k] = gm * amp; /* product_kernel.rs:58  gradient2 * kernel1 */
      }
      g[0] = amp * km; /* product_kernel.rs:57  gradient1 * kernel2, gradient1 = c (constant_kernel.rs:31-38) */
    }
  for (int i = 0; i < n; ++i) kmat[IDX(i, i, n)] += noise; /* lml.rs:44 */
  if (kmat_out) memcpy(kmat_out, kmat, sizeof(double) * nn);

  double* chol = (double*)malloc(sizeof(double) * nn);
  memcpy(chol, kmat, sizeof(double) * nn);
  int info = potf2_lower(chol, n); /* lml.rs:47 */
  if (info != 0) {
    free(kmat); free(dk); free(chol);
    return 1; /* lml.rs:48-50 */
  }
  double* alpha = (double*)malloc(sizeof(double) * (size_t)n);
  potrs_lower(chol, n, y, alpha); /* lml.rs:54 */
  double ya = 0.0, logdet = 0.0;
  for (int i = 0; i < n; ++i) {
    ya += y[i] * alpha[i];
    logdet += log(chol[IDX(i, i, n)]);
  }
  if (lml_out) *lml_out = -0.5 * ya - logdet - (double)n / 2.0 * log(2.0 * M_PI); /* lml.rs:57-59 */
  double* kinv = (double*)malloc(sizeof(double) * nn);
  potri_lower(chol, n, kinv);
  if (grad) {
    for (int p = 0; p < d + 2; ++p) grad[p] = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double tmp = alpha[i] * alpha[j] - kinv[IDX(i, j, n)]; /* lml.rs:62 */
        if (i == j) grad[0] += tmp * noise;                    /* noise gradient = eye * noise (lml.rs:41) */
        const double* g = dk + IDX(i, j, n) * (size_t)(d + 1);
        for (int p = 0; p < d + 1; ++p) grad[1 + p] += tmp * g[p];
      }
    for (int p = 0; p < d + 2; ++p) grad[p] *= 0.5; /* lml.rs:67-70 */
  }
  if (alpha_out) memcpy(alpha_out, alpha, sizeof(double) * (size_t)n);
  if (kinv_out) memcpy(kinv_out, kinv, sizeof(double) * nn);
  if (ldiag_out)
    for (int i = 0; i < n; ++i) ldiag_out[i] = chol[IDX(i, i, n)];
  free(kmat); free(dk); free(chol); free(alpha); free(kinv);
  return 0;
}

/* predict.rs:7-52.  var may be NULL.  Returns the number of variances below -sqrt(1e-5) before clamping (:39-48). */
int oracle_predict(const double* xs, int m, const double* x_train, int n, int d, int nu2, double amp, const double* ell,
                   const double* alpha, const double* kinv, double* mean, double* var) {
  double* kt = (double*)malloc(sizeof(double) * (size_t)m * n);
  oracle_kernel(xs, m, x_train, n, d, nu2, amp, ell, kt); /* :18 */
  int warn = 0;
  const double min_noise = 1e-5;
  for (int k = 0; k < m; ++k) {
    double mu = 0.0;
    for (int j = 0; j < n; ++j) mu += kt[IDX(k, j, n)] * alpha[j]; /* :19 */
    mean[k] = mu;
    if (var) {
      double q = 0.0;
      for (int j = 0; j < n; ++j) { /* (k_trans . k_inv)[k][j] * k_trans[k][j]  (:30-37) */
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += kt[IDX(k, i, n)] * kinv[IDX(i, j, n)];
        q += s * kt[IDX(k, j, n)];
      }
      double v = amp + min_noise - q; /* diag = c (product_kernel.rs:72-74) */
      if (v < -sqrt(min_noise)) ++warn;
      if (v < 0.0) v = 0.0; /* :104-127 */
      var[k] = v;
    }
  }
  free(kt);
  return warn;
}
