/*
 * conex oracle (TEST INFRASTRUCTURE ONLY) -- supernodal storage, block
 * Cholesky and block triangular solves.
 *
 * Restates:
 *   conex/triangular_matrix_workspace.cc:14-33   LookupAddress
 *   conex/triangular_matrix_workspace.cc:37-121  TriangularMatrixWorkspace ctor
 *   conex/triangular_matrix_workspace.cc:123-159 Initialize, S_S
 *   conex/block_triangular_operations.cc:114-219 block solves + BlockCholeskyInPlace
 *   conex/supernodal_solver.cc:117-137, 264-273  Get / ToDense
 * The reference keeps double* tables; here they are slab offsets.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "cxo_internal.h"

static long pad4(long n) { /* memory_utils.h:4-12 */
  long r = n % 4;
  return r ? n + 4 - r : n;
}

/* LookupAddress triangular_matrix_workspace.cc:14-33; returns slab offset or -1 */
static long lookup_address(const cxo_workspace* o, int r, int c) {
  int node = o->var_to_sn[c];
  int node_r = o->var_to_sn[r];
  int j = o->var_to_pos[c];
  int ns = o->supernode_size[node];
  if (node == node_r) {
    int i = o->var_to_pos[r];
    return o->diag_off[node] + (long)j * ns + i;
  }
  for (int cnt = 0; cnt < o->separators[node].n; cnt++)
    if (o->separators[node].d[cnt] == r) return o->offd_off[node] + (long)cnt * ns + j;
  return -1;
}

cxo_workspace* cxo_workspace_new(int K, const ivec* path, const int* supernode_size) {
  cxo_workspace* o = (cxo_workspace*)calloc(1, sizeof(cxo_workspace));
  o->K = K;
  o->supernode_size = (int*)malloc(sizeof(int) * (size_t)K);
  memcpy(o->supernode_size, supernode_size, sizeof(int) * (size_t)K);
  int N = 0;
  for (int i = 0; i < K; i++) N += supernode_size[i];
  o->N = N;
  o->var_to_sn = (int*)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
  o->var_to_pos = (int*)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
  o->snodes = ivs_new(K);
  o->separators = ivs_new(K);
  int var = 0;
  for (int cnt = 0; cnt < K; cnt++) {
    for (int i = 0; i < supernode_size[cnt]; i++) {
      iv_push(&o->snodes[cnt], path[cnt].d[i]);
      o->var_to_sn[var] = cnt;
      o->var_to_pos[var] = i;
      var++;
    }
  }
  int nci = K - 1 > 0 ? K - 1 : 0;
  o->col_int = ivs_new(nci);
  o->col_int_start = ivs_new(nci);
  o->pair_first = ivs_new(nci);
  o->pair_second = ivs_new(nci);
  /* build forward, then reverse the outer lists (ctor :114-120) */
  ivec* f_j = ivs_new(nci);
  ivec* f_start = ivs_new(nci);
  ivec* f_pf = ivs_new(nci);
  ivec* f_ps = ivs_new(nci);
  for (int cnt = 0; cnt < K; cnt++) {
    int sep_size = path[cnt].n - supernode_size[cnt];
    for (int i = 0; i < sep_size; i++) {
      int v = path[cnt].d[i + supernode_size[cnt]];
      iv_push(&o->separators[cnt], v);
      int sn = o->var_to_sn[v] - 1;
      if (sn < 0 || sn >= nci) continue; /* malformed; reference throws under !NDEBUG */
      if (f_j[sn].n == 0 || f_j[sn].d[f_j[sn].n - 1] != cnt) {
        iv_push(&f_j[sn], cnt);
        iv_push(&f_start[sn], f_pf[sn].n);
      }
      iv_push(&f_pf[sn], o->var_to_pos[v]);
      iv_push(&f_ps[sn], i);
    }
  }
  for (int sn = 0; sn < nci; sn++) {
    int L = f_j[sn].n;
    iv_push(&o->col_int_start[sn], 0);
    for (int k = L - 1; k >= 0; k--) {
      int b = f_start[sn].d[k];
      int e = (k + 1 < L) ? f_start[sn].d[k + 1] : f_pf[sn].n;
      iv_push(&o->col_int[sn], f_j[sn].d[k]);
      for (int q = b; q < e; q++) {
        iv_push(&o->pair_first[sn], f_pf[sn].d[q]);
        iv_push(&o->pair_second[sn], f_ps[sn].d[q]);
      }
      iv_push(&o->col_int_start[sn], o->pair_first[sn].n);
    }
  }
  ivs_free(f_j, nci);
  ivs_free(f_start, nci);
  ivs_free(f_pf, nci);
  ivs_free(f_ps, nci);

  /* Initialize :123-147 -- block offsets */
  o->diag_off = (long*)malloc(sizeof(long) * (size_t)K);
  o->offd_off = (long*)malloc(sizeof(long) * (size_t)K);
  long off = 0;
  int max_sep = 1;
  for (int j = 0; j < K; j++) {
    long ns = supernode_size[j];
    o->diag_off[j] = off;
    off += pad4(ns * ns);
    o->offd_off[j] = off;
    off += pad4(ns * o->separators[j].n);
    if (o->separators[j].n > max_sep) max_sep = o->separators[j].n;
  }
  o->slab_size = off;
  o->slab = (double*)calloc((size_t)(off > 0 ? off : 1), sizeof(double));
  o->temporaries = (double*)calloc((size_t)max_sep, sizeof(double));

  /* S_S :149-159 */
  o->ss_index = (long**)calloc((size_t)K, sizeof(long*));
  o->ss_count = (int*)calloc((size_t)K, sizeof(int));
  for (int c = 0; c < K; c++) {
    const ivec* s = &o->separators[c];
    int size = (s->n * s->n + s->n) / 2;
    o->ss_count[c] = size;
    o->ss_index[c] = (long*)malloc(sizeof(long) * (size_t)(size > 0 ? size : 1));
    int cnt = 0;
    for (int j = 0; j < s->n; j++)
      for (int i = j; i < s->n; i++) o->ss_index[c][cnt++] = lookup_address(o, s->d[i], s->d[j]);
  }
  return o;
}

void cxo_workspace_free(cxo_workspace* o) {
  if (o) free(o->transpositions);
  if (!o) return;
  int nci = o->K - 1 > 0 ? o->K - 1 : 0;
  free(o->supernode_size);
  ivs_free(o->snodes, o->K);
  ivs_free(o->separators, o->K);
  free(o->diag_off);
  free(o->offd_off);
  free(o->slab);
  free(o->var_to_sn);
  free(o->var_to_pos);
  ivs_free(o->col_int, nci);
  ivs_free(o->col_int_start, nci);
  ivs_free(o->pair_first, nci);
  ivs_free(o->pair_second, nci);
  for (int i = 0; i < o->K; i++) free(o->ss_index[i]);
  free(o->ss_index);
  free(o->ss_count);
  free(o->temporaries);
  free(o);
}

/* Eigen::LLT in-place, lower (Eigen/src/Cholesky/LLT.h llt_inplace<Lower>::unblocked):
 * for k: x = A(k,k) - |A(k,0:k)|^2; fail if x <= 0; A(k,k) = sqrt(x);
 *        A(k+1:,k) = (A(k+1:,k) - A(k+1:,0:k) A(k,0:k)^T) / A(k,k).
 * returns 1 on success. */
int cxo_llt_inplace(int n, double* a, int lda) {
  for (int k = 0; k < n; k++) {
    double x = a[(size_t)k * lda + k];
    for (int p = 0; p < k; p++) x -= a[(size_t)p * lda + k] * a[(size_t)p * lda + k];
    if (!(x > 0.0)) return 0;
    x = sqrt(x);
    a[(size_t)k * lda + k] = x;
    for (int p = 0; p < k; p++) {
      double akp = a[(size_t)p * lda + k];
      for (int i = k + 1; i < n; i++) a[(size_t)k * lda + i] -= a[(size_t)p * lda + i] * akp;
    }
    for (int i = k + 1; i < n; i++) a[(size_t)k * lda + i] /= x;
  }
  return 1;
}

/* y <- L^{-1} y (L lower n x n col-major) */
static void trsv_lower(int n, const double* L, double* y) {
  for (int j = 0; j < n; j++) {
    y[j] /= L[(size_t)j * n + j];
    double yj = y[j];
    for (int i = j + 1; i < n; i++) y[i] -= L[(size_t)j * n + i] * yj;
  }
}
/* y <- L^{-T} y */
static void trsv_lower_t(int n, const double* L, double* y) {
  for (int j = n - 1; j >= 0; j--) {
    double s = y[j];
    for (int i = j + 1; i < n; i++) s -= L[(size_t)j * n + i] * y[i];
    y[j] = s / L[(size_t)j * n + j];
  }
}

/* BlockCholeskyInPlace block_triangular_operations.cc:184-219 */
int cxo_block_cholesky(cxo_workspace* C) {
  for (int i = 0; i < C->K; i++) {
    int ns = C->supernode_size[i];
    int s = C->separators[i].n;
    double* D = C->slab + C->diag_off[i];
    double* B = C->slab + C->offd_off[i];
    if (ns > 0) {
      if (!cxo_llt_inplace(ns, D, ns)) return 0;
    }
    if (ns > 0 && s > 0) {
      /* off <- L^{-1} off, column by column */
      for (int c = 0; c < s; c++) trsv_lower(ns, D, B + (size_t)c * ns);
      int index = 0;
      const long* ss = C->ss_index[i];
      for (int k = 0; k < s; k++) {
        for (int j = k; j < s; j++) {
          double dot = 0;
          const double* ck = B + (size_t)k * ns;
          const double* cj = B + (size_t)j * ns;
          for (int r = 0; r < ns; r++) dot += ck[r] * cj[r];
          C->slab[ss[index++]] -= dot;
        }
      }
    }
  }
  return 1;
}

/* ApplyBlockInverseInPlace :160-182 */
void cxo_apply_block_inverse(const cxo_workspace* m, double* y) {
  int start = 0;
  for (int i = 0; i < m->K - 1; i++) {
    int ns = m->supernode_size[i];
    if (ns == 0) continue;
    const double* D = m->slab + m->diag_off[i];
    const double* B = m->slab + m->offd_off[i];
    double* yi = y + start;
    trsv_lower(ns, D, yi);
    int s = m->separators[i].n;
    if (s > 0) {
      for (int c = 0; c < s; c++) {
        double t = 0;
        for (int r = 0; r < ns; r++) t += B[(size_t)c * ns + r] * yi[r];
        m->temporaries[c] = t;
      }
      for (int c = 0; c < s; c++) y[m->separators[i].d[c]] -= m->temporaries[c];
    }
    start += ns;
  }
  int nl = m->supernode_size[m->K - 1];
  if (nl > 0) trsv_lower(nl, m->slab + m->diag_off[m->K - 1], y + (m->N - nl));
}

/* ApplyBlockInverseOfTransposeInPlace :114-151 */
void cxo_apply_block_inverse_of_transpose(const cxo_workspace* m, double* y) {
  int K = m->K;
  int* start = (int*)malloc(sizeof(int) * (size_t)(K + 1));
  start[0] = 0;
  for (int i = 0; i < K; i++) start[i + 1] = start[i] + m->supernode_size[i];
  for (int i = K - 2; i >= 0; i--) {
    int ns1 = m->supernode_size[i + 1];
    if (ns1 == 0) continue;
    double* yp = y + start[i + 1];
    trsv_lower_t(ns1, m->slab + m->diag_off[i + 1], yp);
    for (int jc = 0; jc < m->col_int[i].n; jc++) {
      int j = m->col_int[i].d[jc];
      int nsj = m->supernode_size[j];
      double* res = y + start[j];
      const double* B = m->slab + m->offd_off[j];
      for (int q = m->col_int_start[i].d[jc]; q < m->col_int_start[i].d[jc + 1]; q++) {
        int pf = m->pair_first[i].d[q];
        int ps = m->pair_second[i].d[q];
        double w = yp[pf];
        for (int r = 0; r < nsj; r++) res[r] -= B[(size_t)ps * nsj + r] * w;
      }
    }
  }
  if (m->supernode_size[0] > 0) trsv_lower_t(m->supernode_size[0], m->slab + m->diag_off[0], y);
  free(start);
}

/* ------------------------------------------------------------------ LDLT path
 * Eigen::RLDLT (conex/RLDLT.h:298-431): diagonal pivoting on the largest |diagonal| entry,
 * left-looking column update, pivots with |d| <= 1e-9 clamped to +-1e-9.  Lower triangle only. */
int cxo_rldlt_inplace(int n, double* a, int lda, int* tr) {
#define M_(i, j) a[(size_t)(j) * lda + (i)]
  const double reg = 1e-9;
  int ret = 1;
  if (n <= 1) {
    if (n == 1) {
      if (fabs(M_(0, 0)) < reg) M_(0, 0) = M_(0, 0) < 0 ? -reg : reg;
      tr[0] = 0;
    }
    return 1; /* the size <= 1 branch reports success even when it clamps (RLDLT.h:311-330) */
  }
  double* temp = (double*)malloc(sizeof(double) * (size_t)n);
  for (int k = 0; k < n; k++) {
    int big = k; /* maxCoeff of |diag| over the trailing part: first maximum */
    double best = fabs(M_(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(M_(i, i)) > best) {
        best = fabs(M_(i, i));
        big = i;
      }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; j++) { /* row(k).head(k) <-> row(big).head(k) */
        double t = M_(k, j);
        M_(k, j) = M_(big, j);
        M_(big, j) = t;
      }
      for (int i = big + 1; i < n; i++) { /* col(k).tail(s) <-> col(big).tail(s) */
        double t = M_(i, k);
        M_(i, k) = M_(i, big);
        M_(i, big) = t;
      }
      double t = M_(k, k);
      M_(k, k) = M_(big, big);
      M_(big, big) = t;
      for (int i = k + 1; i < big; i++) {
        double u = M_(i, k);
        M_(i, k) = M_(big, i);
        M_(big, i) = u;
      }
    }
    int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = M_(j, j) * M_(k, j); /* D(0:k) .* A10^T */
      double dot = 0;
      for (int j = 0; j < k; j++) dot += M_(k, j) * temp[j];
      M_(k, k) -= dot;
      for (int i = 0; i < rs; i++) { /* A21 -= A20 * temp */
        double acc = 0;
        for (int j = 0; j < k; j++) acc += M_(k + 1 + i, j) * temp[j];
        M_(k + 1 + i, k) -= acc;
      }
    }
    double akk = M_(k, k);
    if (!(fabs(akk) > 1e-9)) {
      ret = 0;
      M_(k, k) = M_(k, k) < 0 ? -(1e-9) : (1e-9);
      akk = M_(k, k);
    }
    for (int i = 0; i < rs; i++) M_(k + 1 + i, k) /= akk;
  }
  free(temp);
  return ret;
#undef M_
}

static void apply_transpositions(int n, const int* tr, double* y, int stride, int cols) {
  for (int k = 0; k < n; k++)
    if (tr[k] != k)
      for (int c = 0; c < cols; c++) {
        double t = y[(size_t)c * stride + k];
        y[(size_t)c * stride + k] = y[(size_t)c * stride + tr[k]];
        y[(size_t)c * stride + tr[k]] = t;
      }
}
static void apply_transpositions_t(int n, const int* tr, double* y) {
  for (int k = n - 1; k >= 0; k--)
    if (tr[k] != k) {
      double t = y[k];
      y[k] = y[tr[k]];
      y[tr[k]] = t;
    }
}
static void trsv_unit_lower(int n, const double* L, double* y) {
  for (int j = 0; j < n; j++) {
    double yj = y[j];
    for (int i = j + 1; i < n; i++) y[i] -= L[(size_t)j * n + i] * yj;
  }
}
static void trsv_unit_lower_t(int n, const double* L, double* y) {
  for (int j = n - 1; j >= 0; j--) {
    double acc = y[j];
    for (int i = j + 1; i < n; i++) acc -= L[(size_t)j * n + i] * y[i];
    y[j] = acc;
  }
}

/* BlockLDLTInPlace block_triangular_operations.cc:315-349 */
int cxo_block_ldlt(cxo_workspace* C) {
  int ok = 1, start = 0;
  if (!C->transpositions) C->transpositions = (int*)calloc((size_t)(C->N > 0 ? C->N : 1), sizeof(int));
  for (int i = 0; i < C->K; i++) {
    int ns = C->supernode_size[i];
    int s = C->separators[i].n;
    double* D = C->slab + C->diag_off[i];
    double* B = C->slab + C->offd_off[i];
    int* tr = C->transpositions + start;
    if (ns > 0 && !cxo_rldlt_inplace(ns, D, ns, tr)) ok = 0;
    if (ns > 0 && s > 0) {
      apply_transpositions(ns, tr, B, ns, s);                            /* off = P off */
      for (int c = 0; c < s; c++) trsv_unit_lower(ns, D, B + (size_t)c * ns); /* L^{-1} */
      for (int c = 0; c < s; c++)
        for (int r = 0; r < ns; r++) B[(size_t)c * ns + r] = (1.0 / D[(size_t)r * ns + r]) * B[(size_t)c * ns + r];
      int index = 0;
      const long* ss = C->ss_index[i];
      for (int k = 0; k < s; k++)
        for (int j = k; j < s; j++) {
          double dot = 0; /* temp.col(k).dot(off.col(j)), temp = D off */
          for (int r = 0; r < ns; r++)
            dot += (D[(size_t)r * ns + r] * B[(size_t)k * ns + r]) * B[(size_t)j * ns + r];
          C->slab[ss[index++]] -= dot;
        }
    }
    start += ns;
  }
  C->factored_ldlt = 1;
  C->regularized = !ok;
  return ok;
}

/* SolveInPlaceLDLT: inv(M D) then inv(M^T), M = P^T L */
void cxo_solve_ldlt(const cxo_workspace* m, double* y) {
  int K = m->K;
  int* start = (int*)malloc(sizeof(int) * (size_t)(K + 1));
  start[0] = 0;
  for (int i = 0; i < K; i++) start[i + 1] = start[i] + m->supernode_size[i];
  /* ApplyBlockInverseOfMD :265-299 */
  for (int i = 0; i < K; i++) {
    int ns = m->supernode_size[i];
    if (i > 0) {
      int nsp = m->supernode_size[i - 1], s = m->separators[i - 1].n;
      if (nsp > 0 && s > 0) {
        const double* B = m->slab + m->offd_off[i - 1];
        const double* yp = y + start[i - 1];
        for (int c = 0; c < s; c++) {
          double t = 0;
          for (int r = 0; r < nsp; r++) t += B[(size_t)c * nsp + r] * yp[r];
          m->temporaries[c] = t;
        }
        for (int c = 0; c < s; c++) y[m->separators[i - 1].d[c]] -= m->temporaries[c];
      }
    }
    if (ns > 0) {
      apply_transpositions(ns, m->transpositions + start[i], y + start[i], ns, 1);
      trsv_unit_lower(ns, m->slab + m->diag_off[i], y + start[i]);
    }
  }
  for (int i = 0; i < K; i++) {
    int ns = m->supernode_size[i];
    const double* D = m->slab + m->diag_off[i];
    for (int r = 0; r < ns; r++) y[start[i] + r] = (1.0 / D[(size_t)r * ns + r]) * y[start[i] + r];
  }
  /* ApplyBlockInverseOfMTranspose :222-263 */
  if (K > 0 && m->supernode_size[K - 1] > 0) {
    int nl = m->supernode_size[K - 1];
    trsv_unit_lower_t(nl, m->slab + m->diag_off[K - 1], y + start[K - 1]);
    apply_transpositions_t(nl, m->transpositions + start[K - 1], y + start[K - 1]);
  }
  for (int i = K - 2; i >= 0; i--) {
    double* yp = y + start[i + 1];
    for (int jc = 0; jc < m->col_int[i].n; jc++) {
      int j = m->col_int[i].d[jc];
      int nsj = m->supernode_size[j];
      double* res = y + start[j];
      const double* B = m->slab + m->offd_off[j];
      for (int q = m->col_int_start[i].d[jc]; q < m->col_int_start[i].d[jc + 1]; q++) {
        int pf = m->pair_first[i].d[q];
        int ps = m->pair_second[i].d[q];
        double w = yp[pf];
        for (int r = 0; r < nsj; r++) res[r] -= B[(size_t)ps * nsj + r] * w;
      }
    }
    int ns = m->supernode_size[i];
    if (ns > 0) {
      trsv_unit_lower_t(ns, m->slab + m->diag_off[i], y + start[i]);
      apply_transpositions_t(ns, m->transpositions + start[i], y + start[i]);
    }
  }
  free(start);
}

/* Get / ToDense supernodal_solver.cc:117-137, 264-273 */
void cxo_workspace_to_dense(const cxo_workspace* o, double* out) {
  int N = o->N;
  memset(out, 0, sizeof(double) * (size_t)N * (size_t)N);
  for (int node = 0; node < o->K; node++) {
    int ns = o->supernode_size[node];
    if (ns == 0) continue;
    int first = o->snodes[node].d[0];
    const double* D = o->slab + o->diag_off[node];
    const double* B = o->slab + o->offd_off[node];
    for (int oj = 0; oj < ns; oj++) {
      int j = first + oj;
      for (int oi = oj; oi < ns; oi++) out[(size_t)j * N + (first + oi)] = D[(size_t)oj * ns + oi];
      for (int k = 0; k < o->separators[node].n; k++) {
        int i = o->separators[node].d[k];
        if (i >= j) out[(size_t)j * N + i] = B[(size_t)k * ns + oj];
      }
    }
  }
}
