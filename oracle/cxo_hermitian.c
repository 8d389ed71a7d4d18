/* TEST INFRASTRUCTURE ONLY (see conex_oracle.h): CPU restatement of the reference's Hermitian
 * PSD cone over R / C / H, plane by plane exactly as the reference computes it.
 *
 * Follows  conex/jordan_matrix_algebra.cc:57-210 (Identity, ConjugateTranspose, Multiply with
 *          its sign / index tables, Add, ScalarMultiply, JordanMultiply, QuadraticRepresentation,
 *          TraceInnerProduct), :379-452 (inner_product<d>, the 4-argument ApproximateEigenvalues),
 *          conex/exponential_map.cc:15-43 (DoExponentialMap: degree-2 Taylor, two squarings),
 *          conex/hermitian_psd.cc:10-91, 171-230 and hermitian_psd.h:103-115.
 *
 * A hyper-complex matrix is d real planes (d in {1,2,4,8}); plane p of an r x c matrix starts at
 * p*r*c, column-major.  Octonions (d = 8, order <= 3) use the same tables; their cone follows the
 * reference's separate, heuristic rules (hermitian_psd.cc:108-168, restated in cxo_program.c).
 *
 * The reference draws the Lanczos start vector with T::Random (libc rand(), unseeded): that is
 * unpinned, so the restatement and the product share the stateless generator cxo_hc_random()
 * below; only properties are pinned by the reference's tests (hermitian_psd_test.cc:25-66).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cxo_internal.h"

/* jordan_matrix_algebra.cc:103-124, upper-left 4x4 corner; target plane indx(i,j) = i ^ j */
/* the sign table M of jordan_matrix_algebra.cc:104-111 (the algebra of dimension d uses its top-left
 * d x d corner; the product index is i ^ j: the table indx of :113-120) */
static const int kSign[8][8] = {{1, 1, 1, 1, 1, 1, 1, 1},     {1, -1, -1, 1, -1, 1, 1, -1},
                                {1, 1, -1, -1, -1, -1, 1, 1}, {1, -1, 1, -1, -1, 1, -1, 1},
                                {1, 1, 1, 1, -1, -1, -1, -1}, {1, -1, 1, -1, 1, -1, 1, -1},
                                {1, -1, -1, 1, 1, -1, -1, 1}, {1, 1, -1, -1, 1, 1, -1, -1}};

double cxo_hc_random(uint64_t id, uint64_t call, uint64_t idx) {
  uint64_t z = 0x243F6A8885A308D3ull + id * 0x9E3779B97F4A7C15ull + call * 0xD1B54A32D192ED03ull +
               idx * 0x8CB92BA72F3D8DD7ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0; /* uniform in [-1, 1) */
}

static void rmm(int m, int n, int k, const double* A, const double* B, double* C) {
  for (int j = 0; j < n; j++) {
    double* c = C + (size_t)j * m;
    for (int i = 0; i < m; i++) c[i] = 0;
    for (int p = 0; p < k; p++) {
      double b = B[(size_t)j * k + p];
      const double* a = A + (size_t)p * m;
      for (int i = 0; i < m; i++) c[i] += a[i] * b;
    }
  }
}

/* Z (r x c) = X (r x k) * Y (k x c); Z must not alias X or Y.  :101-138 */
void cxo_hc_multiply(int d, int r, int k, int c, const double* X, const double* Y, double* Z) {
  size_t rc = (size_t)r * c, rk = (size_t)r * k, kc = (size_t)k * c;
  double* t = (double*)malloc(sizeof(double) * rc);
  memset(Z, 0, sizeof(double) * rc * d);
  for (int i = 0; i < d; i++)
    for (int j = 0; j < d; j++) {
      double* z = Z + (size_t)(i ^ j) * rc;
      rmm(r, c, k, X + (size_t)i * rk, Y + (size_t)j * kc, t);
      if (kSign[i][j] >= 1)
        for (size_t q = 0; q < rc; q++) z[q] += t[q];
      else
        for (size_t q = 0; q < rc; q++) z[q] -= t[q];
    }
  free(t);
}

/* :89-99; X is r x c, Z is c x r */
void cxo_hc_conj_transpose(int d, int r, int c, const double* X, double* Z) {
  size_t rc = (size_t)r * c;
  for (int p = 0; p < d; p++)
    for (int j = 0; j < c; j++)
      for (int i = 0; i < r; i++) {
        double v = X[p * rc + (size_t)j * r + i];
        Z[p * rc + (size_t)i * c + j] = p == 0 ? v : -v;
      }
}

/* :204-210: per plane cwiseProduct().colwise().sum().sum() */
double cxo_hc_trace_inner_product(int d, int n, const double* X, const double* Y) {
  size_t nn = (size_t)n * n;
  double ip = 0;
  for (int p = 0; p < d; p++) {
    double tot = 0;
    for (int j = 0; j < n; j++) {
      double cs = 0;
      for (int i = 0; i < n; i++) cs += X[p * nn + (size_t)j * n + i] * Y[p * nn + (size_t)j * n + i];
      tot += cs;
    }
    ip += tot;
  }
  return ip;
}

static void hc_jordan(int d, int n, const double* x, const double* y, double* z) { /* :164-169 */
  size_t sz = (size_t)n * n * d;
  double* a = (double*)malloc(sizeof(double) * sz);
  double* b = (double*)malloc(sizeof(double) * sz);
  cxo_hc_multiply(d, n, n, n, x, y, a);
  cxo_hc_multiply(d, n, n, n, y, x, b);
  for (size_t q = 0; q < sz; q++) z[q] = (a[q] + b[q]) * .5;
  free(a);
  free(b);
}

/* :171-177: Q(x) y = 2 x o (x o y) - (x o x) o y */
void cxo_hc_quadratic_representation(int d, int n, const double* x, const double* y, double* out) {
  size_t sz = (size_t)n * n * d;
  double* t1 = (double*)malloc(sizeof(double) * sz);
  double* t2 = (double*)malloc(sizeof(double) * sz);
  double* t3 = (double*)malloc(sizeof(double) * sz);
  hc_jordan(d, n, x, y, t1);
  hc_jordan(d, n, x, t1, t2);
  for (size_t q = 0; q < sz; q++) t2[q] = t2[q] * 2; /* X1 */
  hc_jordan(d, n, x, x, t1);
  hc_jordan(d, n, t1, y, t3);
  for (size_t q = 0; q < sz; q++) out[q] = t2[q] + t3[q] * -1; /* Add(X1, ScalarMultiply(.., -1)) */
  free(t1);
  free(t2);
  free(t3);
}

/* DoGeodesicUpdateScaled exponential_map.cc:131-144:
 *   herm(c^2 w + 2 c k Q(w) s + k^2 Q(w) (Q(s) w)),  c = 1.5, k = 0.5
 * (the octonion cone's TakeStep, hermitian_psd.cc:116-127; out must not alias w or s) */
void cxo_hc_geodesic_update_scaled(int d, int n, const double* w, const double* s, double* out) {
  size_t sz = (size_t)n * n * d;
  const double c = 1.5, k = 1.0 / 2.0;
  double* q1 = (double*)malloc(sizeof(double) * sz);
  double* q2 = (double*)malloc(sizeof(double) * sz);
  double* q3 = (double*)malloc(sizeof(double) * sz);
  cxo_hc_quadratic_representation(d, n, w, s, q1);
  cxo_hc_quadratic_representation(d, n, s, w, q2);
  cxo_hc_quadratic_representation(d, n, w, q2, q3);
  for (size_t q = 0; q < sz; q++) q1[q] = (w[q] * (c * c) + q1[q] * (2 * k * c)) + q3[q] * (k * k);
  cxo_hc_conj_transpose(d, n, n, q1, q2);
  for (size_t q = 0; q < sz; q++) out[q] = (q1[q] + q2[q]) * .5;
  free(q1);
  free(q2);
  free(q3);
}

/* inner_product<d>(V, U) = (V.col(0)^* U.col(1)).at(0)(0,0)  :379-384.  V, U are n x 2. */
static double hc_ip(int d, int n, const double* V, const double* U) {
  /* plane 0 of Multiply(ConjugateTranspose(v0), u1): contributions i == j in loop order */
  double z = 0;
  for (int i = 0; i < d; i++) {
    const double* v0 = V + (size_t)i * 2 * n;
    const double* u1 = U + (size_t)i * 2 * n + n;
    double dot = 0;
    for (int k = 0; k < n; k++) dot += (i == 0 ? v0[k] : -v0[k]) * u1[k];
    if (kSign[i][i] >= 1)
      z += dot;
    else
      z -= dot;
  }
  return z;
}

static void hc_apply_cols(int d, int n, const double* WS, const double* WSt, const double* V, double* U) {
  /* U.col(0) = WS V.col(0); U.col(1) = WS^* V.col(1) */
  size_t n2 = (size_t)2 * n;
  double* v = (double*)malloc(sizeof(double) * (size_t)n * d);
  double* u = (double*)malloc(sizeof(double) * (size_t)n * d);
  for (int col = 0; col < 2; col++) {
    for (int p = 0; p < d; p++) memcpy(v + (size_t)p * n, V + p * n2 + (size_t)col * n, sizeof(double) * n);
    cxo_hc_multiply(d, n, n, 1, col == 0 ? WS : WSt, v, u);
    for (int p = 0; p < d; p++) memcpy(U + p * n2 + (size_t)col * n, u + (size_t)p * n, sizeof(double) * n);
  }
  free(v);
  free(u);
}

/* ApproximateEigenvalues(WS, W, r, num_iter)  :386-452.  r: d planes of n.  Returns count. */
int cxo_hc_approximate_eigenvalues(int d, int n, const double* WS, const double* W, const double* r,
                                   int num_iter, double* eigs) {
  size_t n2 = (size_t)2 * n, sz = n2 * d, nn = (size_t)n * n;
  double* V = (double*)calloc(sz, sizeof(double));
  double* U = (double*)calloc(sz, sizeof(double));
  double* Vprev = (double*)calloc(sz, sizeof(double));
  double* WSt = (double*)malloc(sizeof(double) * nn * d);
  double* alpha = (double*)calloc((size_t)num_iter + 1, sizeof(double));
  double* beta = (double*)calloc((size_t)num_iter + 1, sizeof(double));
  double* wr = (double*)malloc(sizeof(double) * (size_t)n * d);
  cxo_hc_conj_transpose(d, n, n, WS, WSt);
  cxo_hc_multiply(d, n, n, 1, W, r, wr);
  for (int p = 0; p < d; p++) {
    memcpy(V + p * n2, wr + (size_t)p * n, sizeof(double) * n);
    memcpy(V + p * n2 + n, r + (size_t)p * n, sizeof(double) * n);
  }
  double sc = 1.0 / sqrt(hc_ip(d, n, V, V));
  for (size_t q = 0; q < sz; q++) V[q] = V[q] * sc;
  memcpy(Vprev, V, sizeof(double) * sz);
  hc_apply_cols(d, n, WS, WSt, V, U);
  double scaling = hc_ip(d, n, U, U);
  alpha[0] = hc_ip(d, n, V, U);
  for (size_t q = 0; q < sz; q++) U[q] = U[q] + V[q] * -alpha[0];
  int cnt = 0;
  for (int j = 1; j < num_iter; j++) {
    beta[j - 1] = hc_ip(d, n, U, U);
    if (beta[j - 1] < 1e-5 * scaling) break;
    beta[j - 1] = sqrt(beta[j - 1]);
    memcpy(Vprev, V, sizeof(double) * sz);
    double ib = 1.0 / beta[j - 1];
    for (size_t q = 0; q < sz; q++) V[q] = U[q] * ib;
    hc_apply_cols(d, n, WS, WSt, V, U);
    alpha[j] = hc_ip(d, n, V, U);
    for (size_t q = 0; q < sz; q++) U[q] = U[q] + V[q] * -alpha[j];
    for (size_t q = 0; q < sz; q++) U[q] = U[q] + Vprev[q] * -beta[j - 1];
    cnt++;
  }
  int ne = cxo_tridiagonal_eigenvalues(cnt + 1, alpha, beta, eigs);
  free(V);
  free(U);
  free(Vprev);
  free(WSt);
  free(alpha);
  free(beta);
  free(wr);
  return ne;
}

/* DoExponentialMap exponential_map.cc:15-43: y = (I + x/4 + x^2/32)^4 */
void cxo_hc_exponential_map(int d, int n, const double* x, double* y) {
  size_t nn = (size_t)n * n, sz = nn * d;
  double* xpow = (double*)malloc(sizeof(double) * sz);
  double* t = (double*)malloc(sizeof(double) * sz);
  for (size_t q = 0; q < sz; q++) xpow[q] = x[q] * 1.0 / 4.0; /* xinput * 1.0 / pow(2, squarings) */
  memcpy(y, xpow, sizeof(double) * sz);
  for (int i = 0; i < n; i++) y[(size_t)i * n + i] += 1;
  cxo_hc_multiply(d, n, n, n, x, xpow, t);
  for (size_t q = 0; q < sz; q++) t[q] *= 1.0 / 2 * 1.0 / 4.0; /* rescale(1/i * 1/2^squarings) */
  for (size_t q = 0; q < sz; q++) y[q] = y[q] + t[q];
  cxo_hc_multiply(d, n, n, n, y, y, xpow);
  cxo_hc_multiply(d, n, n, n, xpow, xpow, y);
  free(xpow);
  free(t);
}
