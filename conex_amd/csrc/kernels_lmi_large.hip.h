// Dense-LMI kernels for orders whose working set does not fit LDS (n > 60): every n x n product
// runs on the fp64 matrix pipe through the batched GEMM of kernels_gemm.hip.h, with the matrices
// resident in HBM; the remaining steps are streaming or one-workgroup-per-constraint kernels.
//
// Reference semantics (same as kernels_lmi.hip.h, which holds the LDS-resident forms):
//   ConstructSchurComplementSystem(DenseLMIConstraint*)  dense_lmi_constraint.cc:72-103
//   PrepareStep / GetWeightedSlackEigenvalues            psd_constraint.cc:45-84, 97-128
//   TakeStep / GeodesicUpdate (Pade[3/3] + PartialPivLU) psd_constraint.cc:13-28, 86-90
//   AffineUpdate                                         psd_constraint.cc:33-43
//   AsymmetricLanczos                                    approximate_eigenvalues.cc:178-239
//
// Assembly:  P_i = A_i W and its transpose (one batched NN GEMM with a transposed second
// output), then  G(i,j) = tr(P_i P_j) = sum_k P_i[k] * P_j^T[k]  as ONE (m+1) x n^2 x (m+1)
// GEMM per constraint (TN, split-K with an ordered reduction) -- the second product W (A_i W) of
// the reference is never formed.  Row m of that product is AQc, its corner <c,Qc>.
#pragma once
#include "kernels_gemm.hip.h"
#include "kernels_lmi.hip.h"

namespace cxk {

struct LmiLargeWs {
  double* P;     // count x (m+1) x n^2   P_i = A_i W   (index m: C W)
  double* PT;    // count x (m+1) x n^2   transposes
  double* Gf;    // count x (m+1)^2       contraction result (lower triangle)
  double* part;  // splits x count x (m+1)^2 split-K partials
  double* tmp;   // count x 8 x n^2       step temporaries (aliases P: never live together)
  int* piv;      // count x n             pivot rows of the Pade LU
  int splits;
  // Hermitian cones over C / H (herm_d = 2 / 4), folded form: fold = n / herm_d > 0, Aleft = the first
  // `fold` columns of every [A_1 .. A_m C], side by side (count x n x fold (m+1), leading dimension n)
  int fold;
  const double* Aleft;
};

// ---- assembly ---------------------------------------------------------------------------
// Split-K reduction of Gf, its scatter into the constraint's Schur block and the traces
// AW(i) = tr(P_i), <w,c> = tr(P_C) in ONE launch (they were three: gemm_reduce_splits,
// lmi_large_finalize, lmi_large_traces -- each a launch and two memory hops for a 51 x 51 result at
// C2).  One wavefront per output element, summation orders unchanged: lane l adds partials
// l, l+64, ... then the fixed butterfly (gemm_reduce_splits); traces as in lmi_large_traces.
__global__ void __launch_bounds__(256)
lmi_large_reduce_finalize(LmiGroup g, Arena ar, LmiLargeWs ws, const double* __restrict__ src, int splits,
                          int64_t sCs) {
  const int n = g.n, m = g.m, m1 = m + 1, nn = n * n;
  const int mem = blockIdx.y, id = g.ids[mem];
  const int lane = threadIdx.x & 63;
  int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  // (folded Hermitian form: the contraction already is tr / herm_d, see LmiLargeSchurFolded)
  const double osc = (g.herm_d > 1 && !ws.fold) ? 1.0 / g.herm_d : 1.0;  // exact (power of two)
  if (splits <= 1) {
    // nothing to reduce: a THREAD per output element (waves [0, ew)), then a wave per trace.  (The
    // wave-per-element form below would return the same bits -- one addend and 63 zeros -- at
    // 64 times the wavefronts: 95 us of dispatch at 1000 constraints of order 28, m = 28.)
    const int ew = (m1 * m1 + 63) >> 6;
    if (w < ew) {
      const int e = w * 64 + lane, i = e % m1, j = e / m1;
      if (e >= m1 * m1 || i < j) return;
      const double acc = src[(size_t)mem * m1 * m1 + e] * osc;
      if (i < m)
        ar.G[ar.g_off[id] + i + (size_t)j * m] = acc;
      else if (j < m)
        ar.AQcc[ar.r_off[id] + j] = acc;
      else
        ar.sc[2 * id + 1] = acc;
      return;
    }
    w += m1 * m1 - ew;  // the trace branch below
  }
  if (w < m1 * m1) {
    const int i = w % m1, j = w / m1;
    if (i < j) return;
    const double* P = src + (size_t)mem * m1 * m1 + i + (size_t)j * m1;
    double acc = 0.0;
    for (int q = lane; q < (splits > 1 ? splits : 1); q += 64) acc += P[q * sCs];
    acc = WaveSum(acc) * osc;
    if (lane == 0) {
      if (i < m)
        ar.G[ar.g_off[id] + i + (size_t)j * m] = acc;
      else if (j < m)
        ar.AQcc[ar.r_off[id] + j] = acc;
      else
        ar.sc[2 * id + 1] = acc;
    }
  } else if (w < m1 * m1 + m1) {
    const int i = w - m1 * m1;
    // (folded: P_i is its top `fold` rows, fold x n with leading dimension fold; tr / herm_d = the trace
    //  of its first block)
    const int nr = ws.fold ? ws.fold : n;
    const double* Pi = ws.P + (size_t)mem * m1 * nn + (size_t)i * nr * n;
    double t = 0;
    for (int r = lane; r < nr; r += 64) t += Pi[r + (size_t)r * nr];
    t = WaveSum(t) * osc;
    if (lane == 0) {
      if (i < m)
        ar.AWc[ar.r_off[id] + i] = t;
      else
        ar.sc[2 * id] = t;
    }
  }
}


// ---- step kernels -------------------------------------------------------------------------
// S = sum_i y_i A_i - k C   (dense_lmi_constraint.cc:8-27); grid (blocks over n^2, count)
__global__ void __launch_bounds__(256) lmi_large_slack(LmiGroup g, StepArgs sa, double* __restrict__ S) {
  const int n = g.n, m = g.m, nn = n * n;
  const int mem = blockIdx.y, id = g.ids[mem];
  extern __shared__ double sy[];
  for (int q = threadIdx.x; q < m; q += blockDim.x) sy[q] = sa.y[sa.cl_perm[sa.cl_ptr[id] + q]];
  __syncthreads();
  const double* A = g.A + (size_t)mem * g.a_stride;
  const double* Cm = g.C + (size_t)mem * nn;
  if (g.sp_pptr) {  // sparse group: nonzeros of each position, variable index ascending
    const int* pp = g.sp_pptr + (size_t)mem * nn;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x) {
      double s = 0;
      for (int e = pp[q]; e < pp[q + 1]; e++) s += sy[g.sp_pvar[e]] * g.sp_pval[e];
      S[(size_t)mem * nn + q] = s - sa.c_weight * Cm[q];
    }
    return;
  }
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x) {
    double s = 0;
    for (int i = 0; i < m; i++) s += sy[i] * A[(size_t)i * nn + q];
    s -= sa.c_weight * Cm[q];
    S[(size_t)mem * nn + q] = s;
  }
}

// out = a * X + d * I  (per constraint; stride n^2)
__global__ void __launch_bounds__(256) lmi_large_axpd(int n, const double* __restrict__ X, double a, double d,
                                                      double* __restrict__ out) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x)
    out[base + q] = a * X[base + q] + ((q % n == q / n) ? d : 0.0);
}

// X = (WS + e I) * alpha, exactly as psd_constraint.cc:86-90 orders it (add, then scale if != 1)
__global__ void __launch_bounds__(256) lmi_large_step_arg(int n, const double* __restrict__ WS, double e, double alpha,
                                                          double* __restrict__ X) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x) {
    double x = WS[base + q];
    if (q % n == q / n) x += e;
    if (alpha != 1.0) x *= alpha;
    X[base + q] = x;
  }
}

// aug = [ V - U | V + U ],  V = 12 X2 + 120 I   (exponential_map_pade.cc:23-32); aug is n x 2n
__global__ void __launch_bounds__(256) lmi_large_pade_system(int n, const double* __restrict__ X2,
                                                             const double* __restrict__ U, double* __restrict__ aug) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x) {
    const double v = X2[base + q] * 12.0 + ((q % n == q / n) ? 120.0 : 0.0);
    const double u = U[base + q];
    aug[2 * base + q] = -u + v;
    aug[2 * base + nn + q] = u + v;
  }
}

// Y = (X/4 + I) + T * 0.125 with T = X (X/4): the degree-2 Taylor head of DoExponentialMap
// (exponential_map.cc:23-37); Xq = X * 1.0 / 4.0 is produced by lmi_large_quarter.
__global__ void __launch_bounds__(256) lmi_large_quarter(int n, const double* __restrict__ X, double* __restrict__ Xq) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x)
    Xq[base + q] = X[base + q] * 1.0 / 4.0;
}
__global__ void __launch_bounds__(256) lmi_large_taylor_head(int n, const double* __restrict__ Xq,
                                                             const double* __restrict__ T, double* __restrict__ Y) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x)
    Y[base + q] = (Xq[base + q] + ((q % n == q / n) ? 1.0 : 0.0)) + T[base + q] * 0.125;
}

// ---- blocked partial-pivot LU of the Pade system, batched over the constraints of a group ----
// aug (n x 2n per constraint, column-major, ld = n) = [ V - U | V + U ]; the solution E of
// (V - U) E = V + U overwrites the right half.  Eigen::PartialPivLU semantics (first largest
// |entry| of the column is the pivot; exponential_map_pade.cc:23-32): 32-column panels are
// factored with pivoting inside one workgroup (panel in LDS), the row swaps and the triangular
// solves run column-parallel, every O(n^3) update is the batched fp64 MFMA GEMM.
constexpr int kLuNB = 32;

// Panel factorization: rows k0..n-1, columns k0..k0+nb-1.  grid = constraints, one THREAD PER
// ROW with the row's nb <= 32 panel entries in registers (static indices: the column loop is
// unrolled).  Per column: block-wide arg-max of |a[j]| over rows >= j (first maximum), the pivot
// row and row j trade places through two LDS lines, the pivot row doubles as the broadcast
// operand of the rank-1 update.  Three barriers per column, no LDS-resident panel (rows <= 1024).
struct LuPanelShared {
  double line[2][kLuNB];
  double s_val[16];
  int s_idx[16];
  int s_piv;
};

// Column steps J .. NB-1 by compile-time recursion (every register index is an immediate).
template <int NB, int J>
struct LuPanelSteps {
  static __device__ __forceinline__ void run(double (&a)[NB], LuPanelShared& sh, int* piv, int k0, int nb,
                                             int i, bool has) {
    if constexpr (J < NB) {
      if (J < nb) {  // uniform
        const int lane = i & 63, wave = i >> 6, nw = blockDim.x >> 6;
        double best = (has && i >= J) ? fabs(a[J]) : -1.0;
        int bi = i;
        for (int off = 32; off > 0; off >>= 1) {
          const double ov = __shfl_xor(best, off, 64);
          const int oi = __shfl_xor(bi, off, 64);
          if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
          }
        }
        if (lane == 0) {
          sh.s_val[wave] = best;
          sh.s_idx[wave] = bi;
        }
        __syncthreads();
        if (i == 0) {
          double b = sh.s_val[0];
          int p = sh.s_idx[0];
          for (int w = 1; w < nw; w++)
            if (sh.s_val[w] > b || (sh.s_val[w] == b && sh.s_idx[w] < p)) {
              b = sh.s_val[w];
              p = sh.s_idx[w];
            }
          sh.s_piv = p;
          piv[k0 + J] = k0 + p;
        }
        __syncthreads();
        const int p = sh.s_piv;
        if (i == p) {
#pragma unroll
          for (int c = 0; c < NB; c++) sh.line[0][c] = a[c];
        }
        if (i == J && p != J) {
#pragma unroll
          for (int c = 0; c < NB; c++) sh.line[1][c] = a[c];
        }
        __syncthreads();
        if (i == J) {
#pragma unroll
          for (int c = 0; c < NB; c++) a[c] = sh.line[0][c];
        } else if (i == p) {
#pragma unroll
          for (int c = 0; c < NB; c++) a[c] = sh.line[1][c];
        }
        if (has && i > J) {
          const double l = a[J] / sh.line[0][J];
          a[J] = l;
#pragma unroll
          for (int c = J + 1; c < NB; c++) a[c] -= l * sh.line[0][c];
        }
      }
      LuPanelSteps<NB, J + 1>::run(a, sh, piv, k0, nb, i, has);
    }
  }
};

template <int NB>
__global__ void __launch_bounds__(1024) lu_panel(double* __restrict__ aug_all, int n, int k0, int nb,
                                                 int* __restrict__ piv_all) {
  __shared__ LuPanelShared sh;
  double* aug = aug_all + (size_t)blockIdx.x * 2 * n * n;
  int* piv = piv_all + (size_t)blockIdx.x * n;
  const int rows = n - k0, i = threadIdx.x;
  const bool has = i < rows;
  double a[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) a[c] = (has && c < nb) ? aug[(k0 + i) + (size_t)(k0 + c) * n] : 0.0;
  LuPanelSteps<NB, 0>::run(a, sh, piv, k0, nb, i, has);
  if (has) {
#pragma unroll
    for (int c = 0; c < NB; c++)
      if (c < nb) aug[(k0 + i) + (size_t)(k0 + c) * n] = a[c];
  }
}

// Columns outside the panel (0..k0-1 and k0+nb..2n-1): apply the panel's row swaps, then for the
// columns to the right the unit-lower solve x <- L11^-1 x.  grid = (column blocks, constraints).
// One thread per column; NB is the compile-time block size (registers with static indices; the
// ragged last block takes the NB = 0 run-time form).
template <int NB>
__global__ void __launch_bounds__(256) lu_swap_trsm(double* __restrict__ aug_all, int n, int k0, int nb_rt,
                                                    const int* __restrict__ piv_all) {
  __shared__ double L[kLuNB * kLuNB];
  __shared__ int sp[kLuNB];
  const int nb = NB > 0 ? NB : nb_rt;
  double* aug = aug_all + (size_t)blockIdx.y * 2 * n * n;
  const int* piv = piv_all + (size_t)blockIdx.y * n;
  for (int q = threadIdx.x; q < nb * nb; q += blockDim.x) {
    const int i = q % nb, j = q / nb;
    L[i + j * nb] = aug[(k0 + i) + (size_t)(k0 + j) * n];
  }
  if (threadIdx.x < nb) sp[threadIdx.x] = piv[k0 + threadIdx.x];
  __syncthreads();
  const int ncols = 2 * n - nb;
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < ncols; w += gridDim.x * blockDim.x) {
    const int c = w < k0 ? w : w + nb;
    double* col = aug + (size_t)c * n;
    for (int j = 0; j < nb; j++) {
      const int p = sp[j];
      if (p != k0 + j) {
        const double t = col[k0 + j];
        col[k0 + j] = col[p];
        col[p] = t;
      }
    }
    if (c >= k0 + nb) {
      if constexpr (NB > 0) {
        double x[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) x[i] = col[k0 + i];
#pragma unroll
        for (int i = 0; i < NB; i++) {
#pragma unroll
          for (int j = 0; j < i; j++) x[i] -= L[i + j * NB] * x[j];
        }
#pragma unroll
        for (int i = 0; i < NB; i++) col[k0 + i] = x[i];
      } else {
        for (int i = 0; i < nb; i++) {
          double acc = col[k0 + i];
          for (int j = 0; j < i; j++) acc -= L[i + j * nb] * col[k0 + j];
          col[k0 + i] = acc;
        }
      }
    }
  }
}

// Right-hand-side block rows k0..k0+nb-1 <- U11^-1 (those rows); one thread per RHS column.
template <int NB>
__global__ void __launch_bounds__(256) lu_backsolve_block(double* __restrict__ aug_all, int n, int k0, int nb_rt) {
  __shared__ double U[kLuNB * kLuNB];
  const int nb = NB > 0 ? NB : nb_rt;
  double* aug = aug_all + (size_t)blockIdx.y * 2 * n * n;
  for (int q = threadIdx.x; q < nb * nb; q += blockDim.x) {
    const int i = q % nb, j = q / nb;
    U[i + j * nb] = aug[(k0 + i) + (size_t)(k0 + j) * n];
  }
  __syncthreads();
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    double* col = aug + (size_t)(n + c) * n + k0;
    if constexpr (NB > 0) {
      double x[NB];
#pragma unroll
      for (int i = 0; i < NB; i++) x[i] = col[i];
#pragma unroll
      for (int i = NB - 1; i >= 0; i--) {
#pragma unroll
        for (int j = i + 1; j < NB; j++) x[i] -= U[i + j * NB] * x[j];
        x[i] = x[i] / U[i + i * NB];
      }
#pragma unroll
      for (int i = 0; i < NB; i++) col[i] = x[i];
    } else {
      for (int i = nb - 1; i >= 0; i--) {
        double acc = col[i];
        for (int j = i + 1; j < nb; j++) acc -= U[i + j * nb] * col[j];
        col[i] = acc / U[i + i * nb];
      }
    }
  }
}

inline hipError_t LmiLargeLuSolve(int n, int count, double* aug, int* piv, hipStream_t st) {
  if (n > 1024) return hipErrorNotSupported;  // one panel row per thread
  const int64_t an = 2 * (int64_t)n * n;
  hipError_t e;
  auto gemm = [&](int M, int N, int K, const double* A, const double* B, double* C) {
    GemmArgs g{};
    g.M = M;
    g.N = N;
    g.K = K;
    g.A = A;
    g.lda = n;
    g.sA1 = an;
    g.B = B;
    g.ldb = n;
    g.sB1 = an;
    g.C = C;
    g.ldc = n;
    g.sC1 = an;
    g.inner = 1;
    g.alpha = -1.0;
    g.beta = 1.0;
    g.splits = 1;
    return LaunchGemm(g, false, false, count, st);
  };
  for (int k0 = 0; k0 < n; k0 += kLuNB) {
    const int nb = std::min(kLuNB, n - k0), below = n - k0 - nb;
    {
      const int threads = std::min(1024, ((n - k0 + 63) / 64) * 64);
      lu_panel<kLuNB><<<count, threads, 0, st>>>(aug, n, k0, nb, piv);
    }
    if (nb == kLuNB)
      lu_swap_trsm<kLuNB><<<dim3((2 * n - nb + 63) / 64, count), 64, 0, st>>>(aug, n, k0, nb, piv);
    else
      lu_swap_trsm<0><<<dim3((2 * n - nb + 63) / 64, count), 64, 0, st>>>(aug, n, k0, nb, piv);
    if (below > 0) {  // A22 -= L21 U12 over all columns to the right (incl. the right-hand sides)
      if ((e = gemm(below, 2 * n - k0 - nb, nb, aug + (k0 + nb) + (size_t)k0 * n,
                    aug + k0 + (size_t)(k0 + nb) * n, aug + (k0 + nb) + (size_t)(k0 + nb) * n)) != hipSuccess)
        return e;
    }
  }
  const int nblk = (n + kLuNB - 1) / kLuNB;
  for (int kb = nblk - 1; kb >= 0; kb--) {
    const int k0 = kb * kLuNB, nb = std::min(kLuNB, n - k0);
    if (nb == kLuNB)
      lu_backsolve_block<kLuNB><<<dim3((n + 63) / 64, count), 64, 0, st>>>(aug, n, k0, nb);
    else
      lu_backsolve_block<0><<<dim3((n + 63) / 64, count), 64, 0, st>>>(aug, n, k0, nb);
    if (k0 > 0) {  // RHS[0:k0, :] -= U[0:k0, k0:k0+nb] E_blk
      if ((e = gemm(k0, n, nb, aug + (size_t)k0 * n, aug + k0 + (size_t)n * n, aug + (size_t)n * n)) != hipSuccess)
        return e;
    }
  }
  return hipGetLastError();
}

// W = (T + T^T) / 2
__global__ void __launch_bounds__(256) lmi_large_symmetrize(int n, const double* __restrict__ T, double* __restrict__ W) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x) {
    const int a = q % n, b = q / n;
    W[base + q] = (T[base + q] + T[base + b + (size_t)a * n]) * 0.5;
  }
}

// AffineUpdate tail: W = W (1 + e) + T   (psd_constraint.cc:33-43)
__global__ void __launch_bounds__(256) lmi_large_affine(int n, double e, const double* __restrict__ T, double* __restrict__ W) {
  const int nn = n * n;
  const size_t base = (size_t)blockIdx.y * nn;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nn; q += gridDim.x * blockDim.x) {
    const double w = (e == 0) ? W[base + q] : W[base + q] * (1 + e);
    W[base + q] = w + T[base + q];
  }
}

// Block-wide two-sided Lanczos on WS (HBM) with V = [W r, r]; all vectors in LDS.
// y0 = M x (thread per row, coalesced), y1 = M^T x (wave per column).
__device__ inline void BlockGemvBoth(int n, const double* __restrict__ M, const double* x0, const double* x1,
                                     double* y0, double* y1) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  // y0 = M x0: thread per row, eight independent loads in flight (the matrix comes from L2)
  for (int i = tid; i < n; i += blockDim.x) {
    double s = 0;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = M[i + (size_t)(k + u) * n];
#pragma unroll
      for (int u = 0; u < 8; u++) s = fma(v[u], x0[k + u], s);
    }
    for (; k < n; k++) s = fma(M[i + (size_t)k * n], x0[k], s);
    y0[i] = s;
  }
  // y1 = M^T x1: wave per column, coalesced along the column
  for (int i = wave; i < n; i += nw) {
    double s = 0;
    for (int k = lane; k < n; k += 64) s = fma(M[k + (size_t)i * n], x1[k], s);
    s = WaveSum(s);
    if (lane == 0) y1[i] = s;
  }
}

// mode 0: PrepareStep tail; mode 1: GetWeightedSlackEigenvalues tail.  WS and S in HBM.
template <int MODE>
__global__ void __launch_bounds__(512) lmi_large_spectrum(LmiGroup g, StepArgs sa, const double* __restrict__ WS_all,
                                                          const double* __restrict__ S_all) {
  extern __shared__ double lds[];
  const int n = g.n, nn = n * n;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* WS = WS_all + (size_t)mem * nn;
  const double* S = S_all + (size_t)mem * nn;
  const double* W = g.W + (size_t)mem * nn;
  double* V0 = lds;
  double* V1 = V0 + n;
  double* U0 = V1 + n;
  double* U1 = U0 + n;
  double* P0 = U1 + n;
  double* P1 = P0 + n;
  double* rr = P1 + n;
  const bool herm = g.herm_d != 0;
  const int num_iter = herm ? (n / g.herm_d) / 2 + 1 : n / 2;
  double* alpha = rr + n;
  double* beta = alpha + num_iter + 1;
  double* red = beta + num_iter + 1;  // 16
  __shared__ int s_index;
  const int tid = threadIdx.x;
  if (tid == 0) {
    int idx = 0;
    for (int i = 1; i < n; i++)
      if (WS[i + (size_t)i * n] > WS[idx + (size_t)idx * n]) idx = i;
    s_index = idx;
  }
  __syncthreads();
  // PrepareStep aliases minus_s and WS (psd_constraint.cc:60-69): its start vector is a column of
  // WS; GetWeightedSlackEigenvalues starts from the column of minus_s.
  const double* r = (MODE == 0 ? WS : S) + (size_t)s_index * n;
  for (int i = tid; i < n; i += blockDim.x) {
    const double ri = herm ? HcRandom(id, sa.call, i) : r[i];  // T::Random(n,1) for Hermitian
    rr[i] = ri;
    V1[i] = ri;
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    double s = 0;
    for (int k = 0; k < n; k++) s = fma(W[i + (size_t)k * n], rr[k], s);
    V0[i] = s;
  }
  __syncthreads();
  double ip = 0;
  for (int i = tid; i < n; i += blockDim.x) ip = fma(V0[i], V1[i], ip);
  const double nrm = sqrt(BlockSum(ip, red));
  const double inrm = 1.0 / nrm;
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    V0[i] = herm ? V0[i] * inrm : V0[i] / nrm;
    V1[i] = herm ? V1[i] * inrm : V1[i] / nrm;
    P0[i] = V0[i];
    P1[i] = V1[i];
  }
  __syncthreads();
  int cnt = 0;
  double beta_prev = 0, scaling = 0;
  for (int j = 0; j < num_iter; j++) {
    if (j > 0) {
      double b2 = 0;
      for (int i = tid; i < n; i += blockDim.x) b2 = fma(U0[i], U1[i], b2);
      b2 = BlockSum(b2, red);
      __syncthreads();
      if (herm ? (b2 < 1e-5 * scaling) : (b2 < 1e-6)) break;
      beta_prev = sqrt(b2);
      if (tid == 0) beta[j - 1] = beta_prev;
      const double ib = 1.0 / beta_prev;
      for (int i = tid; i < n; i += blockDim.x) {
        P0[i] = V0[i];
        P1[i] = V1[i];
        V0[i] = herm ? U0[i] * ib : U0[i] / beta_prev;
        V1[i] = herm ? U1[i] * ib : U1[i] / beta_prev;
      }
      cnt++;
      __syncthreads();
    }
    BlockGemvBoth(n, WS, V0, V1, U0, U1);
    __syncthreads();
    if (j == 0 && herm) {
      double sc = 0;
      for (int i = tid; i < n; i += blockDim.x) sc = fma(U0[i], U1[i], sc);
      scaling = BlockSum(sc, red);
      __syncthreads();
    }
    double a = 0;
    for (int i = tid; i < n; i += blockDim.x) a = fma(V0[i], U1[i], a);
    a = BlockSum(a, red);
    __syncthreads();
    if (tid == 0) alpha[j] = a;
    for (int i = tid; i < n; i += blockDim.x) {
      double u0 = U0[i] - a * V0[i], u1 = U1[i] - a * V1[i];
      if (j > 0) {
        u0 -= beta_prev * P0[i];
        u1 -= beta_prev * P1[i];
      }
      U0[i] = u0;
      U1[i] = u1;
    }
    __syncthreads();
  }
  __syncthreads();  // alpha / beta complete
  if (tid < 64) TridiagMinMaxWave(cnt + 1, alpha, beta, &red[8], &red[9]);
  // tr(WS WS) and tr(WS)
  double t2 = 0, t1 = 0;
  for (int q = tid; q < nn; q += blockDim.x) {
    const int a = q % n, b = q / n;
    t2 = fma(WS[q], WS[b + (size_t)a * n], t2);
    if (a == b) t1 += WS[q];
  }
  t2 = BlockSum(t2, red);
  __syncthreads();
  t1 = BlockSum(t1, red);
  __syncthreads();
  if (tid == 0) {
    double mn = red[8], mx = red[9];
    if (!sa.no_clamp) ClampToSpectrumBound(n, t1, t2, &mn, &mx);
    if (g.herm_d > 1) {
      t2 /= g.herm_d;
      t1 /= g.herm_d;
    }
    const int rank = herm ? n / g.herm_d : n;
    if (MODE == 0) {
      const double l1 = fabs(sa.e_weight + mn), l2 = fabs(sa.e_weight + mx);
      sa.info[2 * id] = t2 + 2 * t1 + rank;
      sa.info[2 * id + 1] = l1 < l2 ? l2 : l1;
    } else {
      sa.info[4 * id] = -mx;
      sa.info[4 * id + 1] = -mn;
      sa.info[4 * id + 2] = t2;
      sa.info[4 * id + 3] = -t1;
    }
  }
}

inline size_t LmiLargeSpectrumLds(int n) { return sizeof(double) * (size_t)(7 * n + 2 * (n / 2 + 2) + 16); }

// ---- host-side drivers --------------------------------------------------------------------
inline GemmArgs SquareGemm(int n, const double* A, int64_t sA, const double* B, int64_t sB, double* C,
                           int64_t sC) {
  GemmArgs a{};
  a.M = a.N = a.K = n;
  a.A = A;
  a.lda = n;
  a.sA1 = sA;
  a.B = B;
  a.ldb = n;
  a.sB1 = sB;
  a.C = C;
  a.ldc = n;
  a.sC1 = sC;
  a.inner = 1;
  a.alpha = 1.0;
  a.beta = 0.0;
  a.splits = 1;
  return a;
}

// Hermitian cones over C / H through their real representation L(X) of order n = d n0 (block (k, j) =
// +- X_{k ^ j}; block row 0 = (X_0, -X_1, .., -X_{d-1}), block column 0 = (X_0, X_1, ..)^T): products of
// such matrices have the same form, so a product is known from its top n0 rows, and
//     tr(P_x P_y) = d sum_C s_C tr(T_x[:, C] T_y[:, C]),   T = top n0 rows, s = (+, -, .., -),
// because block (C, 0) of P_y is -(block (0, C)) for C >= 1.  So only  T_i^T = W A_i[:, 0 .. n0)  is
// formed (ONE GEMM against the first n0 columns of all matrices side by side: 1 / d of the
// multiply-adds, full 64-row tiles), and the contraction runs over n0 n entries instead of n^2
// against the partner image  Z_y[r, C n0 + k] = s_C T_y^T[C n0 + r, k]  -- the reference computes
// plane by plane (jordan_matrix_algebra.cc:101-138); this is the same saving on the matrix pipe.
__global__ void __launch_bounds__(256)
lmi_large_herm_partner(int n, int n0, int m1, int64_t stride, const double* __restrict__ PTl, double* __restrict__ Z) {
  const int mem = blockIdx.y;
  const double* src = PTl + (size_t)mem * stride;
  double* dst = Z + (size_t)mem * stride;
  const int per = n0 * n, total = per * m1;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const int i = q / per, e = q - i * per, r = e % n0, col = e / n0, C = col / n0, k = col - C * n0;
    const double v = src[(size_t)(i * n0 + k) * n + C * n0 + r];
    dst[(size_t)i * per + e] = C == 0 ? v : -v;
  }
}

inline hipError_t LmiLargeReduceFinalizeLaunch(const LmiGroup& g, const Arena& ar, const LmiLargeWs& ws, const GemmArgs& a,
                                               hipStream_t st) {
  const int m1 = g.m + 1;
  const bool split = a.splits > 1;
  const int waves = (split ? m1 * m1 : (m1 * m1 + 63) / 64) + m1;
  lmi_large_reduce_finalize<<<dim3((waves + 3) / 4, g.count), 256, 0, st>>>(g, ar, ws, split ? ws.part : ws.Gf, a.splits,
                                                                          a.sCs);
  return hipGetLastError();
}

inline hipError_t LmiLargeSchurFolded(const LmiGroup& g, const Arena& ar, const LmiLargeWs& ws, hipStream_t st) {
  const int n = g.n, m1 = g.m + 1, n0 = ws.fold;
  const int64_t nn = (int64_t)n * n, per = (int64_t)n0 * n, stride = m1 * nn;
  double* PTl = ws.PT;               // T_i^T side by side: n x (n0 m1), leading dimension n
  double* Z = ws.PT + stride / 2;    // partner images, n0 x n each (per m1 <= stride / 2: herm_d >= 2)
  hipError_t e;
  {
    GemmArgs a{};
    a.M = n;
    a.N = n0 * m1;
    a.K = n;
    a.A = g.W;
    a.lda = n;
    a.sA1 = nn;
    a.B = ws.Aleft;
    a.ldb = n;
    a.sB1 = per * m1;
    a.C = PTl;
    a.ldc = n;
    a.sC1 = stride;
    a.Ct = ws.P;  // T_i (n0 x n, leading dimension n0), matrix i at i * per
    a.ldct = n0;
    a.sT1 = stride;
    a.ctb = n0;
    a.sTb = per - n0;
    a.inner = 1;
    a.alpha = 1.0;
    a.beta = 0.0;
    a.splits = 1;
    if ((e = LaunchGemm(a, false, false, g.count, st)) != hipSuccess) return e;
  }
  {
    const int blocks = (int)std::min<int64_t>((per * m1 + 255) / 256, 256);
    lmi_large_herm_partner<<<dim3(blocks, g.count), 256, 0, st>>>(n, n0, m1, stride, PTl, Z);
  }
  GemmArgs a{};
  a.M = a.N = m1;
  a.K = (int)per;
  a.A = ws.P;
  a.lda = per;
  a.sA1 = stride;
  a.B = Z;
  a.ldb = per;
  a.sB1 = stride;
  a.C = ws.Gf;
  a.ldc = m1;
  a.sC1 = (int64_t)m1 * m1;
  a.inner = 1;
  a.alpha = 1.0;
  a.beta = 0.0;
  a.lower_only = 1;
  a.splits = std::max(1, std::min(ws.splits, (int)((per + kGemmBK - 1) / kGemmBK)));
  a.sCs = (int64_t)g.count * m1 * m1;
  if (a.splits > 1) a.C = ws.part;
  if ((e = LaunchGemm(a, true, false, g.count, st)) != hipSuccess) return e;
  return LmiLargeReduceFinalizeLaunch(g, ar, ws, a, st);
}

inline hipError_t LmiLargeSchur(const LmiGroup& g, const Arena& ar, const LmiLargeWs& ws, hipStream_t st) {
  if (ws.fold) return LmiLargeSchurFolded(g, ar, ws, st);
  const int n = g.n, m = g.m, m1 = m + 1;
  const int64_t nn = (int64_t)n * n;
  hipError_t e;
  {  // PT[c,i] = W[c] A[c,i] (= P[c,i]^T, both factors symmetric), P = its transposed copy: ONE
     // n x (n (m + 1)) GEMM per constraint against the matrices side by side, [A_1 ... A_m C] (the
     // host stores C behind the A_i: contiguous, leading dimension n), instead of m + 1 GEMMs of n^3
     // -- at n = 200 the 64-wide tiles are 78 % full instead of 61 %, and the affine term no longer
     // costs a launch of its own (a lone 200^3 product: 16 workgroups, 17 us of latency).  Same
     // products in the same k order as A_i W: same bits.
    GemmArgs a{};
    a.M = n;
    a.N = n * m1;
    a.K = n;
    a.A = g.W;
    a.lda = n;
    a.sA1 = nn;
    a.B = g.A;
    a.ldb = n;
    a.sB1 = g.a_stride;
    a.C = ws.PT;
    a.ldc = n;
    a.sC1 = m1 * nn;
    a.Ct = ws.P;
    a.ldct = n;
    a.sT1 = m1 * nn;
    a.ctb = n;
    a.sTb = nn - n;
    a.inner = 1;
    a.alpha = 1.0;
    a.beta = 0.0;
    a.splits = 1;
    if ((e = LaunchGemm(a, false, false, g.count, st)) != hipSuccess) return e;
  }
  {  // Gf[c] = X^T Y, X = P[c] (n^2 x m1), Y = PT[c]
    GemmArgs a{};
    a.M = a.N = m1;
    a.K = (int)nn;
    a.A = ws.P;
    a.lda = nn;
    a.sA1 = m1 * nn;
    a.B = ws.PT;
    a.ldb = nn;
    a.sB1 = m1 * nn;
    a.C = ws.Gf;
    a.ldc = m1;
    a.sC1 = (int64_t)m1 * m1;
    a.inner = 1;
    a.alpha = 1.0;
    a.beta = 0.0;
    a.lower_only = 1;
    a.splits = ws.splits;
    a.sCs = (int64_t)g.count * m1 * m1;
    const bool split = a.splits > 1;
    if (split) a.C = ws.part;  // partials; reduced by lmi_large_reduce_finalize
    if ((e = LaunchGemm(a, true, false, g.count, st)) != hipSuccess) return e;
    const int waves = (split ? m1 * m1 : (m1 * m1 + 63) / 64) + m1;
    lmi_large_reduce_finalize<<<dim3((waves + 3) / 4, g.count), 256, 0, st>>>(g, ar, ws, split ? ws.part : ws.Gf,
                                                                            a.splits, a.sCs);
  }
  return hipGetLastError();
}

inline dim3 ElemGrid(int n, int count) {
  const int blocks = (n * n + 255) / 256;
  return dim3(blocks < 256 ? blocks : 256, count);
}

// mode 0: PrepareStep (stores WS in g.T1; affine branch updates W), mode 1: eigenvalue query
inline hipError_t LmiLargePrepare(const LmiGroup& g, const StepArgs& sa, const LmiLargeWs& ws, int mode,
                                  hipStream_t st) {
  const int n = g.n;
  const int64_t nn = (int64_t)n * n;
  double* S = ws.tmp;                        // count x nn
  double* WS = (mode == 0) ? g.T1 : ws.tmp + (size_t)g.count * nn;
  double* T = ws.tmp + 2 * (size_t)g.count * nn;
  hipError_t e;
  lmi_large_slack<<<ElemGrid(n, g.count), 256, sizeof(double) * g.m, st>>>(g, sa, S);
  GemmArgs a = SquareGemm(n, g.W, nn, S, nn, WS, nn);
  if ((e = LaunchGemm(a, false, false, g.count, st)) != hipSuccess) return e;
  if (mode == 0 && sa.affine) {
    GemmArgs b = SquareGemm(n, WS, nn, g.W, nn, T, nn);
    if ((e = LaunchGemm(b, false, false, g.count, st)) != hipSuccess) return e;
    lmi_large_affine<<<ElemGrid(n, g.count), 256, 0, st>>>(n, sa.e_weight, T, g.W);
    return hipGetLastError();
  }
  if (mode == 0)
    lmi_large_spectrum<0><<<g.count, 512, LmiLargeSpectrumLds(n), st>>>(g, sa, WS, S);
  else
    lmi_large_spectrum<1><<<g.count, 512, LmiLargeSpectrumLds(n), st>>>(g, sa, WS, S);
  return hipGetLastError();
}

// W <- sym( pade33( (WS + e I) alpha ) W )
inline hipError_t LmiLargeTakeStep(const LmiGroup& g, const StepArgs& sa, const LmiLargeWs& ws, hipStream_t st) {
  const int n = g.n, count = g.count;
  const int64_t nn = (int64_t)n * n;
  double* X = ws.tmp;
  double* X2 = X + (size_t)count * nn;
  double* T4 = X2 + (size_t)count * nn;
  double* U = T4 + (size_t)count * nn;
  double* aug = U + (size_t)count * nn;  // count x 2 nn
  double* EW = aug + 2 * (size_t)count * nn;
  hipError_t e;
  const dim3 eg = ElemGrid(n, count);
  lmi_large_step_arg<<<eg, 256, 0, st>>>(n, g.T1, sa.e_weight, sa.step_size, X);
  if (g.herm_d) {
    // E = ((I + X/4 + X^2/32)^2)^2 (exponential_map.cc:15-43), W <- sym(E W)
    double* Xq = X2;
    double* T = T4;
    double* Y = U;
    lmi_large_quarter<<<eg, 256, 0, st>>>(n, X, Xq);
    GemmArgs t0 = SquareGemm(n, X, nn, Xq, nn, T, nn);
    if ((e = LaunchGemm(t0, false, false, count, st)) != hipSuccess) return e;
    lmi_large_taylor_head<<<eg, 256, 0, st>>>(n, Xq, T, Y);
    GemmArgs t1 = SquareGemm(n, Y, nn, Y, nn, T, nn);
    if ((e = LaunchGemm(t1, false, false, count, st)) != hipSuccess) return e;
    GemmArgs t2 = SquareGemm(n, T, nn, T, nn, Y, nn);
    if ((e = LaunchGemm(t2, false, false, count, st)) != hipSuccess) return e;
    GemmArgs t3 = SquareGemm(n, Y, nn, g.W, nn, EW, nn);
    if ((e = LaunchGemm(t3, false, false, count, st)) != hipSuccess) return e;
    lmi_large_symmetrize<<<eg, 256, 0, st>>>(n, EW, g.W);
    return hipGetLastError();
  }
  GemmArgs a = SquareGemm(n, X, nn, X, nn, X2, nn);
  if ((e = LaunchGemm(a, false, false, count, st)) != hipSuccess) return e;
  lmi_large_axpd<<<eg, 256, 0, st>>>(n, X2, 1.0, 60.0, T4);
  GemmArgs b = SquareGemm(n, X, nn, T4, nn, U, nn);
  if ((e = LaunchGemm(b, false, false, count, st)) != hipSuccess) return e;
  lmi_large_pade_system<<<eg, 256, 0, st>>>(n, X2, U, aug);
  if ((e = LmiLargeLuSolve(n, count, aug, ws.piv, st)) != hipSuccess) return e;
  GemmArgs c = SquareGemm(n, aug + nn, 2 * nn, g.W, nn, EW, nn);
  if ((e = LaunchGemm(c, false, false, count, st)) != hipSuccess) return e;
  lmi_large_symmetrize<<<eg, 256, 0, st>>>(n, EW, g.W);
  return hipGetLastError();
}

}  // namespace cxk
