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
#include "device_utils.h"

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

constexpr int kGemmBM = 64, kGemmBN = 64, kGemmBK = 16;
constexpr int kGemmLdM = 80;  // [k][m] image: 64 + 16 -> rows k, k+1 fall in disjoint bank halves
constexpr int kGemmLdK = 17;  // [m][k] image: odd stride
constexpr int kGemmLdsDoubles = 64 * 65;  // result staging (>= the two operand images)
static_assert(2 * kGemmBK * kGemmLdM <= kGemmLdsDoubles && 2 * 64 * kGemmLdK <= kGemmLdsDoubles, "");

typedef double gemm_d4 __attribute__((ext_vector_type(4)));

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_f64_mfma(GemmArgs g) {
  __shared__ double lds[kGemmLdsDoubles];
  double* sA = lds;
  double* sB = lds + (TA ? 64 * kGemmLdK : kGemmBK * kGemmLdM);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (g.M + kGemmBM - 1) / kGemmBM;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int m_base = tm * kGemmBM, n_base = tn * kGemmBN;
  if (g.lower_only && m_base + kGemmBM - 1 < n_base) return;  // uniform per workgroup
  const int b1 = blockIdx.z / g.inner, b2 = blockIdx.z % g.inner;
  const double* A = g.A + b1 * g.sA1 + b2 * g.sA2;
  const double* B = g.B + b1 * g.sB1 + b2 * g.sB2;
  // K range of this split, in whole BK steps
  const int ksteps = (g.K + kGemmBK - 1) / kGemmBK;
  const int per = (ksteps + g.splits - 1) / g.splits;
  const int ks0 = blockIdx.y * per, ks1 = min(ksteps, ks0 + per);

  // staging maps: element e = tid + 256 u, u < 4, of a 64 x 16 operand tile
  double ra[4], rb[4];
  auto load_tiles = [&](int k_base) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = tid + 256 * u;
      {
        const int mm = TA ? (e >> 4) : (e & 63), kk = TA ? (e & 15) : (e >> 6);
        const int m = m_base + mm, k = k_base + kk;
        const bool ok = m < g.M && k < g.K;
        ra[u] = ok ? (TA ? A[k + (int64_t)m * g.lda] : A[m + (int64_t)k * g.lda]) : 0.0;
      }
      {
        const int nn = TB ? (e & 63) : (e >> 4), kk = TB ? (e >> 6) : (e & 15);
        const int n = n_base + nn, k = k_base + kk;
        const bool ok = n < g.N && k < g.K;
        rb[u] = ok ? (TB ? B[n + (int64_t)k * g.ldb] : B[k + (int64_t)n * g.ldb]) : 0.0;
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = tid + 256 * u;
      if (TA)
        sA[(e >> 4) * kGemmLdK + (e & 15)] = ra[u];
      else
        sA[(e >> 6) * kGemmLdM + (e & 63)] = ra[u];
      if (TB)
        sB[(e >> 6) * kGemmLdM + (e & 63)] = rb[u];
      else
        sB[(e >> 4) * kGemmLdK + (e & 15)] = rb[u];
    }
  };

  gemm_d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = gemm_d4{0.0, 0.0, 0.0, 0.0};
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int l15 = lane & 15, kq = lane >> 4;

  if (ks0 < ks1) load_tiles(ks0 * kGemmBK);
  for (int ks = ks0; ks < ks1; ks++) {
    __syncthreads();  // previous step's MFMA operand reads are done
    store_tiles();
    __syncthreads();
    if (ks + 1 < ks1) load_tiles((ks + 1) * kGemmBK);
#pragma unroll
    for (int sub = 0; sub < kGemmBK / 4; sub++) {
      const int k = sub * 4 + kq;
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int m = wm + 16 * i + l15;
        a[i] = TA ? sA[m * kGemmLdK + k] : sA[k * kGemmLdM + m];
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int n = wn + 16 * j + l15;
        b[j] = TB ? sB[k * kGemmLdM + n] : sB[n * kGemmLdK + k];
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // result tile -> LDS (row m, column n at m + 65 n), then coalesced global writes
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        lds[(wm + 16 * i + kq + 4 * e) + 65 * (wn + 16 * j + l15)] = acc[i][j][e];
  __syncthreads();
  double* C = g.C + b1 * g.sC1 + b2 * g.sC2 + (g.splits > 1 ? blockIdx.y * g.sCs : 0);
  const bool partial = g.splits > 1;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int mm = e & 63, nn = e >> 6;
    const int m = m_base + mm, n = n_base + nn;
    if (m < g.M && n < g.N && (!g.lower_only || m >= n)) {
      double v = lds[mm + 65 * nn];
      double* dst = C + m + (int64_t)n * g.ldc;
      if (partial)
        *dst = v;
      else
        *dst = (g.beta == 0.0) ? g.alpha * v : g.alpha * v + g.beta * *dst;
    }
  }
  if (g.Ct && !partial) {
    double* Ct = g.Ct + b1 * g.sT1 + b2 * g.sT2;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int nn = e & 63, mm = e >> 6;
      const int m = m_base + mm, n = n_base + nn;
      if (m < g.M && n < g.N)
        Ct[n + (int64_t)m * g.ldct + (g.ctb > 0 ? (int64_t)(n / g.ctb) * g.sTb : 0)] = g.alpha * lds[mm + 65 * nn];
    }
  }
}

// C = alpha * sum_s partial[s] + beta * C over the split partials.  One wavefront per output
// element: lane l adds partials l, l+64, ... in order, then a fixed butterfly -- the summation
// order depends only on `splits`, so results are reproducible run to run.
__global__ void __launch_bounds__(256) gemm_reduce_splits(GemmArgs g, const double* __restrict__ part) {
  const int b1 = blockIdx.z / g.inner, b2 = blockIdx.z % g.inner;
  const double* P = part + b1 * g.sC1 + b2 * g.sC2;
  double* C = g.C + b1 * g.sC1 + b2 * g.sC2;
  const int64_t total = (int64_t)g.M * g.N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t e = (int64_t)blockIdx.x * 4 + wave; e < total; e += (int64_t)gridDim.x * 4) {
    const int m = (int)(e % g.M), n = (int)(e / g.M);
    if (g.lower_only && m < n) continue;
    double acc = 0.0;
    for (int s = lane; s < g.splits; s += 64) acc += P[s * g.sCs + m + (int64_t)n * g.ldc];
    acc = WaveSum(acc);
    if (lane == 0) {
      double* dst = C + m + (int64_t)n * g.ldc;
      *dst = (g.beta == 0.0) ? g.alpha * acc : g.alpha * acc + g.beta * *dst;
    }
  }
}

// Launch helper.  With splits > 1 `g.C` of the GEMM proper must point at the partial buffer
// (splits * sCs doubles per batch level as laid out by the caller) -- see LaunchGemmSplitK.
inline hipError_t LaunchGemm(const GemmArgs& g, bool ta, bool tb, int batch, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return hipSuccess;
  const int tiles = ((g.M + kGemmBM - 1) / kGemmBM) * ((g.N + kGemmBN - 1) / kGemmBN);
  dim3 grid(tiles, g.splits > 1 ? g.splits : 1, batch);
  if (!ta && !tb)
    gemm_f64_mfma<false, false><<<grid, 256, 0, stream>>>(g);
  else if (ta && !tb)
    gemm_f64_mfma<true, false><<<grid, 256, 0, stream>>>(g);
  else if (!ta && tb)
    gemm_f64_mfma<false, true><<<grid, 256, 0, stream>>>(g);
  else
    gemm_f64_mfma<true, true><<<grid, 256, 0, stream>>>(g);
  return hipGetLastError();
}

// Split-K GEMM: partials into `part` (same batch strides as C, split stride g.sCs), then the
// ordered reduction into C.
inline hipError_t LaunchGemmSplitK(GemmArgs g, bool ta, bool tb, int batch, double* part,
                                   hipStream_t stream) {
  if (g.splits <= 1) return LaunchGemm(g, ta, tb, batch, stream);
  GemmArgs p = g;
  p.C = part;
  p.Ct = nullptr;
  hipError_t e = LaunchGemm(p, ta, tb, batch, stream);
  if (e != hipSuccess) return e;
  const int64_t total = (int64_t)g.M * g.N;
  dim3 grid((unsigned)std::min<int64_t>((total + 3) / 4, 4096), 1, batch);
  gemm_reduce_splits<<<grid, 256, 0, stream>>>(g, part);
  return hipGetLastError();
}

}  // namespace cxk
