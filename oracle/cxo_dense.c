/*
 * conex oracle (TEST INFRASTRUCTURE ONLY) -- small dense routines.
 *
 * Restates the Eigen-backed pieces of:
 *   conex/exponential_map_pade.cc:10-32     [3/3] Pade expm, PartialPivLU solve
 *   conex/approximate_eigenvalues.cc:37-126 JacobiSolver (test-only reference path)
 *   conex/approximate_eigenvalues.cc:147-171 symmetric Lanczos
 *   conex/approximate_eigenvalues.cc:173-239 AsymmetricLanczos
 * Eigen's SelfAdjointEigenSolver::computeFromTridiagonal is replaced by an
 * implicit-shift QL iteration (same eigenvalues to rounding).
 * All matrices are column-major.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cxo_internal.h"

/* C(m x n) = A(m x k) * B(k x n) */
static void gemm(int m, int n, int k, const double* A, const double* B, double* C) {
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

/* Solve A X = B in place via LU with partial pivoting (first max wins). A,B n x n / n x nrhs. */
static void lu_solve(int n, int nrhs, double* A, double* B) {
  for (int k = 0; k < n; k++) {
    int piv = k;
    double best = fabs(A[(size_t)k * n + k]);
    for (int i = k + 1; i < n; i++) {
      double v = fabs(A[(size_t)k * n + i]);
      if (v > best) {
        best = v;
        piv = i;
      }
    }
    if (piv != k) {
      for (int j = 0; j < n; j++) {
        double t = A[(size_t)j * n + k];
        A[(size_t)j * n + k] = A[(size_t)j * n + piv];
        A[(size_t)j * n + piv] = t;
      }
      for (int j = 0; j < nrhs; j++) {
        double t = B[(size_t)j * n + k];
        B[(size_t)j * n + k] = B[(size_t)j * n + piv];
        B[(size_t)j * n + piv] = t;
      }
    }
    double d = A[(size_t)k * n + k];
    for (int i = k + 1; i < n; i++) A[(size_t)k * n + i] /= d;
    for (int j = k + 1; j < n; j++) {
      double akj = A[(size_t)j * n + k];
      for (int i = k + 1; i < n; i++) A[(size_t)j * n + i] -= A[(size_t)k * n + i] * akj;
    }
  }
  for (int c = 0; c < nrhs; c++) {
    double* b = B + (size_t)c * n;
    for (int j = 0; j < n; j++) { /* L unit lower */
      double bj = b[j];
      for (int i = j + 1; i < n; i++) b[i] -= A[(size_t)j * n + i] * bj;
    }
    for (int j = n - 1; j >= 0; j--) { /* U */
      b[j] /= A[(size_t)j * n + j];
      double bj = b[j];
      for (int i = 0; i < j; i++) b[i] -= A[(size_t)j * n + i] * bj;
    }
  }
}

/* exponential_map_pade.cc:10-32 */
void cxo_pade_expm(int n, const double* arg, double* result) {
  size_t nn = (size_t)n * n;
  double* V = (double*)malloc(sizeof(double) * nn);   /* even powers */
  double* U = (double*)malloc(sizeof(double) * nn);   /* odd powers */
  double* tmp = (double*)malloc(sizeof(double) * nn);
  double* numer = (double*)malloc(sizeof(double) * nn);
  double* denom = (double*)malloc(sizeof(double) * nn);
  gemm(n, n, n, arg, arg, V);                         /* A^2 */
  memcpy(tmp, V, sizeof(double) * nn);                /* tmp = 1*A^2 + 60 I */
  for (int i = 0; i < n; i++) tmp[(size_t)i * n + i] += 60.0;
  gemm(n, n, n, arg, tmp, U);                         /* U = A * tmp */
  for (size_t i = 0; i < nn; i++) V[i] *= 12.0;       /* V = 12 A^2 + 120 I */
  for (int i = 0; i < n; i++) V[(size_t)i * n + i] += 120.0;
  for (size_t i = 0; i < nn; i++) {
    numer[i] = U[i] + V[i];
    denom[i] = -U[i] + V[i];
  }
  lu_solve(n, n, denom, numer);
  memcpy(result, numer, sizeof(double) * nn);
  free(V);
  free(U);
  free(tmp);
  free(numer);
  free(denom);
}

/* Implicit QL with Wilkinson shift on a symmetric tridiagonal; eigenvalues only, ascending. */
int cxo_tridiagonal_eigenvalues(int n, const double* diag, const double* off, double* out) {
  if (n <= 0) return 0;
  double* d = out;
  double* e = (double*)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; i++) d[i] = diag[i];
  for (int i = 0; i < n - 1; i++) e[i] = off[i];
  e[n - 1] = 0;
  for (int l = 0; l < n; l++) {
    int iter = 0;
    int m;
    do {
      for (m = l; m < n - 1; m++) {
        double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.2204460492503131e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 200) break;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; i--) {
          double f = s * e[i];
          double b = c * e[i];
          r = hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  /* sort ascending */
  for (int i = 1; i < n; i++) {
    double v = d[i];
    int j = i - 1;
    while (j >= 0 && d[j] > v) {
      d[j + 1] = d[j];
      j--;
    }
    d[j + 1] = v;
  }
  free(e);
  return n;
}

/* inner_product approximate_eigenvalues.cc:173-176: V.col(0).dot(U.col(1)) */
static double ip01(int n, const double* V, const double* U) {
  double s = 0;
  for (int i = 0; i < n; i++) s += V[i] * U[n + i];
  return s;
}

/* AsymmetricLanczos approximate_eigenvalues.cc:178-239 */
int cxo_asymmetric_lanczos(int n, const double* WS, const double* W, const double* r,
                           int num_iter, double* eigs) {
  if (n == 1) { /* ApproximateEigenvalues :248-250 */
    eigs[0] = WS[0];
    return 1;
  }
  if (num_iter < 1) return 0;
  double* V = (double*)calloc((size_t)2 * n, sizeof(double));
  double* U = (double*)calloc((size_t)2 * n, sizeof(double));
  double* Vprev = (double*)calloc((size_t)2 * n, sizeof(double));
  double* alpha = (double*)calloc((size_t)num_iter, sizeof(double));
  double* beta = (double*)calloc((size_t)(num_iter > 1 ? num_iter - 1 : 1), sizeof(double));
  /* V.col(1) = r; V.col(0) = W r */
  for (int i = 0; i < n; i++) V[n + i] = r[i];
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int k = 0; k < n; k++) s += W[(size_t)k * n + i] * r[k];
    V[i] = s;
  }
  double nrm = sqrt(ip01(n, V, V));
  for (int i = 0; i < 2 * n; i++) V[i] /= nrm;
  memcpy(Vprev, V, sizeof(double) * 2 * (size_t)n);
  /* U.col(0) = WS V.col(0); U.col(1) = WS^T V.col(1) */
  for (int i = 0; i < n; i++) {
    double s0 = 0, s1 = 0;
    for (int k = 0; k < n; k++) {
      s0 += WS[(size_t)k * n + i] * V[k];
      s1 += WS[(size_t)i * n + k] * V[n + k];
    }
    U[i] = s0;
    U[n + i] = s1;
  }
  alpha[0] = ip01(n, V, U);
  for (int i = 0; i < 2 * n; i++) U[i] -= alpha[0] * V[i];
  int cnt = 0;
  for (int j = 1; j < num_iter; j++) {
    beta[j - 1] = ip01(n, U, U);
    if (beta[j - 1] < 1e-6) break;
    beta[j - 1] = sqrt(beta[j - 1]);
    memcpy(Vprev, V, sizeof(double) * 2 * (size_t)n);
    for (int i = 0; i < 2 * n; i++) V[i] = U[i] / beta[j - 1];
    for (int i = 0; i < n; i++) {
      double s0 = 0, s1 = 0;
      for (int k = 0; k < n; k++) {
        s0 += WS[(size_t)k * n + i] * V[k];
        s1 += WS[(size_t)i * n + k] * V[n + k];
      }
      U[i] = s0;
      U[n + i] = s1;
    }
    alpha[j] = ip01(n, V, U);
    for (int i = 0; i < 2 * n; i++) U[i] = U[i] - alpha[j] * V[i] - beta[j - 1] * Vprev[i];
    cnt++;
  }
  int ne = cxo_tridiagonal_eigenvalues(cnt + 1, alpha, beta, eigs);
  free(V);
  free(U);
  free(Vprev);
  free(alpha);
  free(beta);
  return ne;
}

/* symmetric Lanczos approximate_eigenvalues.cc:147-171 */
int cxo_symmetric_lanczos(int n, const double* A, const double* r0, int num_iter, double* eigs) {
  double* Vm = (double*)calloc((size_t)n * num_iter, sizeof(double));
  double* alpha = (double*)calloc((size_t)num_iter, sizeof(double));
  double* beta = (double*)calloc((size_t)num_iter, sizeof(double));
  double* wprev = (double*)calloc((size_t)n, sizeof(double));
  double* Av = (double*)calloc((size_t)n, sizeof(double));
  double nr = 0;
  for (int i = 0; i < n; i++) nr += r0[i] * r0[i];
  nr = sqrt(nr);
  for (int i = 0; i < n; i++) Vm[i] = r0[i] / nr;
  gemm(n, 1, n, A, Vm, Av);
  double a0 = 0;
  for (int i = 0; i < n; i++) a0 += Vm[i] * Av[i];
  alpha[0] = a0;
  for (int i = 0; i < n; i++) wprev[i] = Av[i] - a0 * Vm[i];
  for (int j = 1; j < num_iter; j++) {
    double* v = Vm + (size_t)j * n;
    double b = 0;
    for (int i = 0; i < n; i++) b += wprev[i] * wprev[i];
    b = sqrt(b);
    beta[j - 1] = b;
    for (int i = 0; i < n; i++) v[i] = wprev[i] / b;
    gemm(n, 1, n, A, v, Av);
    double a = 0;
    for (int i = 0; i < n; i++) a += v[i] * Av[i];
    alpha[j] = a;
    for (int i = 0; i < n; i++) wprev[i] = Av[i] - a * v[i] - b * Vm[(size_t)(j - 1) * n + i];
  }
  int ne = cxo_tridiagonal_eigenvalues(num_iter, alpha, beta, eigs);
  free(Vm);
  free(alpha);
  free(beta);
  free(wprev);
  free(Av);
  return ne;
}

/* JacobiSolver approximate_eigenvalues.cc:37-126 (monomial-basis orthogonal polynomials) */
typedef struct {
  int dim;   /* matrix order */
  int n;     /* number of polynomials */
  const double* W;
  const double* r0;
  double* powers; /* (n+1) matrices dim x dim */
} jacobi_t;

static void jac_eval_poly(const jacobi_t* J, int len, const double* p, double* out) {
  size_t nn = (size_t)J->dim * J->dim;
  for (size_t i = 0; i < nn; i++) out[i] = p[0] * J->powers[i];
  for (int k = 1; k < len; k++)
    for (size_t i = 0; i < nn; i++) out[i] += p[k] * J->powers[(size_t)k * nn + i];
}

/* (EvalPoly(p)^T r0) . (EvalPoly(q) W r0) */
static double jac_inner(const jacobi_t* J, int lp, const double* p, int lq, const double* q) {
  int d = J->dim;
  size_t nn = (size_t)d * d;
  double* P = (double*)malloc(sizeof(double) * nn);
  double* Q = (double*)malloc(sizeof(double) * nn);
  double* a = (double*)calloc((size_t)d, sizeof(double));
  double* wr = (double*)calloc((size_t)d, sizeof(double));
  double* b = (double*)calloc((size_t)d, sizeof(double));
  jac_eval_poly(J, lp, p, P);
  jac_eval_poly(J, lq, q, Q);
  for (int i = 0; i < d; i++) { /* a = P^T r0 */
    double s = 0;
    for (int k = 0; k < d; k++) s += P[(size_t)i * d + k] * J->r0[k];
    a[i] = s;
  }
  gemm(d, 1, d, J->W, J->r0, wr);
  gemm(d, 1, d, Q, wr, b);
  double s = 0;
  for (int i = 0; i < d; i++) s += a[i] * b[i];
  free(P);
  free(Q);
  free(a);
  free(wr);
  free(b);
  return s;
}

int cxo_jacobi_eigenvalues(int dim, const double* A, const double* W, const double* r0, int n,
                           double* eigs) {
  jacobi_t J;
  J.dim = dim;
  J.n = n;
  J.W = W;
  J.r0 = r0;
  size_t nn = (size_t)dim * dim;
  J.powers = (double*)calloc(nn * (size_t)(n + 1), sizeof(double));
  for (int i = 0; i < dim; i++) J.powers[(size_t)i * dim + i] = 1.0;
  for (int i = 1; i <= n; i++) gemm(dim, dim, dim, A, J.powers + (size_t)(i - 1) * nn, J.powers + (size_t)i * nn);

  double* alpha_v = (double*)calloc((size_t)n, sizeof(double));
  double* beta_v = (double*)calloc((size_t)n, sizeof(double));
  double* v = (double*)calloc((size_t)(n + 2) * (size_t)(n + 1), sizeof(double)); /* v[j] length n (+1) */
  int ld = n + 1;
  double* one = (double*)calloc((size_t)ld, sizeof(double));
  one[0] = 1;
  double beta = sqrt(jac_inner(&J, n, one, n, one));
  for (int i = 0; i < n; i++) v[(size_t)1 * ld + i] = one[i] / beta;
  double* Avj = (double*)calloc((size_t)ld, sizeof(double));
  double* vhat = (double*)calloc((size_t)ld, sizeof(double));
  for (int j = 1; j < n; j++) {
    double* vj = v + (size_t)j * ld;
    double* vjm = v + (size_t)(j - 1) * ld;
    Avj[0] = 0;
    for (int i = 1; i < n; i++) Avj[i] = vj[i - 1];
    double alpha = jac_inner(&J, n, Avj, n, vj);
    for (int i = 0; i < n; i++) vhat[i] = Avj[i] - alpha * vj[i] - beta * vjm[i];
    beta = sqrt(jac_inner(&J, n, vhat, n, vhat));
    for (int i = 0; i < n; i++) v[(size_t)(j + 1) * ld + i] = vhat[i] / beta;
    beta_v[j - 1] = beta;
    alpha_v[j - 1] = alpha;
  }
  double* vn = v + (size_t)n * ld;
  Avj[0] = 0;
  for (int i = 1; i <= n; i++) Avj[i] = vn[i - 1];
  alpha_v[n - 1] = jac_inner(&J, n + 1, Avj, n, vn);
  int ne = cxo_tridiagonal_eigenvalues(n, alpha_v, beta_v, eigs);
  free(J.powers);
  free(alpha_v);
  free(beta_v);
  free(v);
  free(one);
  free(Avj);
  free(vhat);
  return ne;
}
