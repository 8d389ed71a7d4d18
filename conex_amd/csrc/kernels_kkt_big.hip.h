// Supernodes whose panel does not fit LDS (more than ~140 columns): blocked, HBM-resident
// factorization and solves, driven from the host one supernode at a time.  Same mathematics as
// the in-kernel paths (reference BlockCholeskyInPlace block_triangular_operations.cc:184-219 and
// the block solves :114-182), organised as a right-looking blocked Cholesky with 32-column
// panels: small LDS kernels factor / solve the 32 x 32 diagonal blocks, every O(n^3) update is a
// batched fp64 MFMA GEMM (kernels_gemm.hip.h):
//     L21  = A21 L11^-T            (big_trsm_rows)
//     A22 -= L21 L21^T             (GEMM NT, lower only)          -- the SYRK of north_star
//     off[k-block,:] = L11^-1 off[k-block,:]; off[below,:] -= L21 off[k-block,:]   (GEMM NN)
//     U = off^T off, t = off^T b   (GEMM TN), scattered to the consumer slots
// Storage is the slab itself: diag block ns x ns column-major, off block ns x s column-major.
#pragma once
#include "kernels_gemm.hip.h"
#include "kernels_kkt.hip.h"

namespace cxk {

constexpr int kBigNB = 32;

// Apply the published updates of descendants to the panel / right-hand side in HBM.
__global__ void __launch_bounds__(256) big_pull(FactorPlan P, SnRec R, double* __restrict__ slab,
                                                double* __restrict__ rhs, int with_matrix) {
  const int ns = R.ns;
  double* D = slab + R.diag_off;
  double* B = slab + R.offd_off;
  const int gsz = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (with_matrix)
    for (int t = R.tg_beg + gid; t < R.tg_end; t += gsz) {
      const int loc = P.tg_loc[t];
      double* dst = loc < ns * ns ? D + loc : B + (loc - ns * ns);
      double acc = *dst;
      const int q1 = P.tr_ptr[t + 1];
      for (int q = P.tr_ptr[t]; q < q1; q++) acc -= P.upd[P.tr_src[q]];
      *dst = acc;
    }
  if (rhs)
    for (int i = gid; i < ns; i += gsz) {
      double acc = rhs[R.start + i];
      const int q1 = P.fs_ptr[R.start + i + 1];
      for (int q = P.fs_ptr[R.start + i]; q < q1; q++) acc -= P.updb[P.fs_src[q]];
      rhs[R.start + i] = acc;
    }
}

// In-place Cholesky of the nb x nb diagonal block at (k0, k0); one workgroup.
__global__ void __launch_bounds__(256) big_diag(double* __restrict__ D, int ns, int k0, int nb,
                                                int* __restrict__ fail) {
  __shared__ double L[kBigNB * kBigNB];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int q = tid; q < nb * nb; q += blockDim.x) {
    const int i = q % nb, j = q / nb;
    L[i + j * nb] = D[(k0 + i) + (size_t)(k0 + j) * ns];
  }
  __syncthreads();
  for (int k = 0; k < nb; k++) {
    const double d = L[k + k * nb];
    double root, inv;
    SqrtAndInverse(d, root, inv);
    if (!(d > 0.0) && tid == 0) bad = 1;
    __syncthreads();
    for (int i = k + tid; i < nb; i += blockDim.x) L[i + k * nb] = (i == k) ? root : L[i + k * nb] * inv;
    __syncthreads();
    const int rows = nb - k - 1;
    for (int idx = tid; idx < rows * rows; idx += blockDim.x) {
      const int i = k + 1 + idx % rows, j = k + 1 + idx / rows;
      if (j <= i) L[i + j * nb] = fma(-L[i + k * nb], L[j + k * nb], L[i + j * nb]);
    }
    __syncthreads();
  }
  if (bad) {
    if (tid == 0) atomicExch(fail, 1);
    return;
  }
  for (int q = tid; q < nb * nb; q += blockDim.x) {
    const int i = q % nb, j = q / nb;
    if (i >= j) D[(k0 + i) + (size_t)(k0 + j) * ns] = L[i + j * nb];
  }
}

// Work items: rows r > k0 + nb - 1 of the panel ( x <- x L11^-T ), columns c of the off block
// ( off[k-block, c] <- L11^-1 off[k-block, c] ) and, last, the right-hand side block.
__global__ void __launch_bounds__(256) big_trsm(double* __restrict__ D, double* __restrict__ B,
                                                double* __restrict__ rhs, int ns, int s, int k0, int nb,
                                                int with_matrix) {
  __shared__ double L[kBigNB * kBigNB];
  __shared__ double dinv[kBigNB];
  for (int q = threadIdx.x; q < nb * nb; q += blockDim.x) {
    const int i = q % nb, j = q / nb;
    L[i + j * nb] = D[(k0 + i) + (size_t)(k0 + j) * ns];
  }
  __syncthreads();
  if (threadIdx.x < nb) dinv[threadIdx.x] = 1.0 / L[threadIdx.x + threadIdx.x * nb];
  __syncthreads();
  const int below = ns - k0 - nb;
  const int n_rows = with_matrix ? below : 0, n_cols = with_matrix ? s : 0;
  const int total = n_rows + n_cols + (rhs ? 1 : 0);
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < total; w += gridDim.x * blockDim.x) {
    double x[kBigNB];
    if (w < n_rows) {  // row of A21: x L11^T = a
      const int r = k0 + nb + w;
      for (int j = 0; j < nb; j++) {
        double acc = D[r + (size_t)(k0 + j) * ns];
        for (int i = 0; i < j; i++) acc = fma(-x[i], L[j + i * nb], acc);
        x[j] = acc * dinv[j];
      }
      for (int j = 0; j < nb; j++) D[r + (size_t)(k0 + j) * ns] = x[j];
    } else {           // column of the off block or the right-hand side: L11 y = b
      double* col = (w < n_rows + n_cols) ? B + (size_t)(w - n_rows) * ns + k0 : rhs + k0;
      for (int i = 0; i < nb; i++) {
        double acc = col[i];
        for (int j = 0; j < i; j++) acc = fma(-L[i + j * nb], x[j], acc);
        x[i] = acc * dinv[i];
      }
      for (int i = 0; i < nb; i++) col[i] = x[i];
    }
  }
}

// rhs block <- L11^-T rhs block (back substitution step); one workgroup, one thread solves.
__global__ void __launch_bounds__(64) big_rhs_block_t(const double* __restrict__ D, double* __restrict__ rhs,
                                                      int ns, int k0, int nb) {
  __shared__ double L[kBigNB * kBigNB];
  for (int q = threadIdx.x; q < nb * nb; q += blockDim.x) {
    const int i = q % nb, j = q / nb;
    L[i + j * nb] = D[(k0 + i) + (size_t)(k0 + j) * ns];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* b = rhs + k0;
    for (int k = nb - 1; k >= 0; k--) {
      double acc = b[k];
      for (int i = k + 1; i < nb; i++) acc = fma(-L[i + k * nb], b[i], acc);
      b[k] = acc * (1.0 / L[k + k * nb]);
    }
  }
}

// b_i -= sum_q off[i, c_q] y[row_q]  (separator terms of the back substitution)
__global__ void __launch_bounds__(256) big_backsep(FactorPlan P, SnRec R, const double* __restrict__ slab,
                                                   double* __restrict__ rhs) {
  const int ns = R.ns;
  const double* B = slab + R.offd_off;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    double acc = rhs[R.start + i];
    for (int q = R.bs_beg; q < R.bs_end; q++) acc -= B[i + (size_t)P.bs_c[q] * ns] * rhs[P.bs_row[q]];
    rhs[R.start + i] = acc;
  }
}

// Scatter U (s x s, from off^T off) and t (s, from off^T b) to the consumer slots.
__global__ void __launch_bounds__(256) big_publish(FactorPlan P, SnRec R, const double* __restrict__ U,
                                                   const double* __restrict__ t, int with_matrix, int with_rhs) {
  const int s = R.nsep;
  const int gsz = gridDim.x * blockDim.x, gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (with_matrix) {
    const int* dst = P.pub_dst + R.upd_off;
    const int npairs = s * (s + 1) / 2;
    for (int e = gid; e < npairs; e += gsz) {
      int k = 0, rem = e;
      while (rem >= s - k) {
        rem -= s - k;
        k++;
      }
      const int j = k + rem;
      P.upd[dst[e]] = U[j + (size_t)k * s];
    }
  }
  if (with_rhs) {
    const int* dst = P.pubb_dst + R.updb_off;
    for (int c = gid; c < s; c += gsz) P.updb[dst[c]] = t[c];
  }
}

inline GemmArgs BigGemm(int M, int N, int K, const double* A, int64_t lda, const double* B, int64_t ldb,
                        double* C, int64_t ldc, double alpha, double beta, int lower_only) {
  GemmArgs a{};
  a.M = M;
  a.N = N;
  a.K = K;
  a.A = A;
  a.lda = lda;
  a.B = B;
  a.ldb = ldb;
  a.C = C;
  a.ldc = ldc;
  a.inner = 1;
  a.alpha = alpha;
  a.beta = beta;
  a.lower_only = lower_only;
  a.splits = 1;
  return a;
}

// mode 0: factor (+ forward when rhs), mode 1: forward only, mode 2: backward.
// ws: at least s*s + s doubles.
inline hipError_t BigSupernodeSweep(const FactorPlan& P, const SnRec& R, int mode, double* slab, double* rhs,
                                    int* fail, double* ws, hipStream_t st) {
  const int ns = R.ns, s = R.nsep;
  double* D = slab + R.diag_off;
  double* B = slab + R.offd_off;
  double* b = rhs ? rhs + R.start : nullptr;
  hipError_t e;
  if (mode == 2) {
    big_backsep<<<(ns + 255) / 256, 256, 0, st>>>(P, R, slab, rhs);
    const int nblk = (ns + kBigNB - 1) / kBigNB;
    for (int kb = nblk - 1; kb >= 0; kb--) {
      const int k0 = kb * kBigNB, nb = std::min(kBigNB, ns - k0), below = ns - k0 - nb;
      if (below > 0) {  // b_blk -= L21^T b_below
        GemmArgs g = BigGemm(nb, 1, below, D + (k0 + nb) + (size_t)k0 * ns, ns, b + k0 + nb, below, b + k0, nb,
                             -1.0, 1.0, 0);
        if ((e = LaunchGemm(g, true, false, 1, st)) != hipSuccess) return e;
      }
      big_rhs_block_t<<<1, 64, 0, st>>>(D, b, ns, k0, nb);
    }
    return hipGetLastError();
  }
  const int with_matrix = mode == 0;
  if (R.tg_end > R.tg_beg || rhs) big_pull<<<64, 256, 0, st>>>(P, R, slab, rhs, with_matrix);
  for (int k0 = 0; k0 < ns; k0 += kBigNB) {
    const int nb = std::min(kBigNB, ns - k0), below = ns - k0 - nb;
    if (with_matrix) big_diag<<<1, 256, 0, st>>>(D, ns, k0, nb, fail);
    const int items = (with_matrix ? below + s : 0) + (rhs ? 1 : 0);
    if (items > 0) big_trsm<<<(items + 255) / 256, 256, 0, st>>>(D, B, b, ns, s, k0, nb, with_matrix);
    if (below > 0) {
      const double* L21 = D + (k0 + nb) + (size_t)k0 * ns;
      if (with_matrix) {
        GemmArgs g1 = BigGemm(below, below, nb, L21, ns, L21, ns, D + (k0 + nb) + (size_t)(k0 + nb) * ns, ns,
                              -1.0, 1.0, 1);
        if ((e = LaunchGemm(g1, false, true, 1, st)) != hipSuccess) return e;
        if (s > 0) {
          GemmArgs g2 = BigGemm(below, s, nb, L21, ns, B + k0, ns, B + k0 + nb, ns, -1.0, 1.0, 0);
          if ((e = LaunchGemm(g2, false, false, 1, st)) != hipSuccess) return e;
        }
      }
      if (rhs) {
        GemmArgs g3 = BigGemm(below, 1, nb, L21, ns, b + k0, nb, b + k0 + nb, below, -1.0, 1.0, 0);
        if ((e = LaunchGemm(g3, false, false, 1, st)) != hipSuccess) return e;
      }
    }
  }
  if (s > 0) {
    double* U = ws;
    double* t = ws + (size_t)s * s;
    if (with_matrix) {
      GemmArgs g = BigGemm(s, s, ns, B, ns, B, ns, U, s, 1.0, 0.0, 0);
      if ((e = LaunchGemm(g, true, false, 1, st)) != hipSuccess) return e;
    }
    if (rhs) {
      GemmArgs g = BigGemm(s, 1, ns, B, ns, b, ns, t, s, 1.0, 0.0, 0);
      if ((e = LaunchGemm(g, true, false, 1, st)) != hipSuccess) return e;
    }
    big_publish<<<(s * (s + 1) / 2 + 255) / 256, 256, 0, st>>>(P, R, U, t, with_matrix, rhs != nullptr);
  }
  return hipGetLastError();
}

}  // namespace cxk
