// Batched fp64 GEMM on the matrix pipe (v_mfma_f64_16x16x4_f64) for the large-order paths:
//   * dense-LMI assembly at orders that do not fit LDS (P_i = A_i W, then the contraction
//     G = X^T Y over K = n^2 with split-K) -- reference dense_lmi_constraint.cc:72-103
//   * W (-S), the exponential-map products of TakeStep -- psd_constraint.cc:13-28, 45-84
//   * supernode trailing updates (SYRK / GEMM) of the blocked Cholesky --
//     block_triangular_operations.cc:184-219
//
//   C[b] (M x N) = alpha * op(A[b]) (M x K) * op(B[b]) (K x N) + beta * C[b]
//
// Storage: column-major.  TA = false: A(m,k) at A[m + k*lda]; TA = true: A(m,k) at A[k + m*lda]
// (the stored matrix is K x M).  TB = false: B(k,n) at B[k + n*ldb]; TB = true: B(k,n) at
// B[n + k*ldb].  Batch index b = blockIdx.z maps to (b / inner, b % inner) with one stride per
// level and operand (stride 0 shares an operand, e.g. W across the matrices of a constraint).
// Split-K: blockIdx.y selects a K range; partial results go to C + split * sCs (beta ignored)
// and are summed in a fixed order by gemm_reduce_splits -- no atomics, reproducible.
//
// Tiling: 64 x 64 x 16 per 256-thread workgroup, four wavefronts in a 2 x 2 grid, 2 x 2 MFMA
// tiles (32 x 32) per wavefront.  Global loads of step k+1 are issued into registers before the
// MFMAs of step k.  LDS images are padded so that the per-lane operand reads (A: 16 rows x 4
// k-values, B: 4 k-values x 16 columns) are bank-conflict free.  The result tile is staged
// through LDS so that C and the optional transposed copy Ct are both written coalesced.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cxk {

struct GemmArgs {
  int M, N, K;
  const double* A;
  int64_t lda, sA1, sA2;
  const double* B;
  int64_t ldb, sB1, sB2;
  double* C;
  int64_t ldc, sC1, sC2;
  double* Ct;  // optional: Ct(n, m) = C(m, n), leading dimension ldct, batch strides sT1/sT2
  int64_t ldct, sT1, sT2;
  int ctb;      // > 0: the columns of C are blocks of ctb columns; block k of Ct starts k * sTb further on
  int64_t sTb;  // (a wide GEMM against stacked matrices whose transposes are stored one after the other)
  int inner;   // batch b -> (b / inner, b % inner)
  double alpha, beta;
  int lower_only;  // write only m >= n, skip tiles above the diagonal (SYRK-shaped updates)
  int splits;      // split-K factor (gridDim.y)
  int64_t sCs;     // stride between split partials
};

constexpr int kGemmBK = 16;  // K granularity of a split (LaunchGemmSplitK deals whole steps of this many)

// Kernels and launchers live in gemm_mfma.hip (own translation unit).
// With splits > 1 `g.C` of the GEMM proper must point at the partial buffer -- see LaunchGemmSplitK.
hipError_t LaunchGemm(const GemmArgs& g, bool ta, bool tb, int batch, hipStream_t stream);
// Split-K GEMM: partials into `part` (same batch strides as C, split stride g.sCs), then the
// ordered reduction into C.
hipError_t LaunchGemmSplitK(GemmArgs g, bool ta, bool tb, int batch, double* part, hipStream_t stream);

}  // namespace cxk
