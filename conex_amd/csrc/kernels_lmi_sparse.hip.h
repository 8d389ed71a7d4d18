// Sparse-LMI Schur assembly.
//
// The C-ABI builds matrix inequalities entry by entry (CONEX_UpdateLinearOperator,
// hermitian_psd.cc:249-275), so the A_i of real programs hold a handful of nonzeros while the
// reference -- and the dense kernels here -- stream and multiply full n x n matrices.  For a
// constraint whose matrices are sparse enough (see LmiSparsePays) the same quantities
//
//   G(i,j) = tr(W A_i W A_j) = sum_{(r,c,a) in A_i} sum_{(p,q,b) in A_j} a b W[c,p] W[q,r]
//   AW(i)  = tr(A_i W)       = sum_{(r,c,a) in A_i} a W[c,r]
//   AQc(i) = tr(A_i W C W)   = sum_{(r,c,a) in A_i} a X[c,r],   X = W C W (dense, C is dense)
//   <w,c>  = sum C o W,  <c,Qc> = sum C o X
//
// (dense_lmi_constraint.cc:72-103, hermitian_psd.cc:171-230) are evaluated from the nonzeros:
// O((sum nnz)^2) instead of O(n^3 m + n^2 m^2), and the m n^2 stream of A disappears from HBM
// (the dense A is never uploaded).  Values equal the dense path's up to summation order
// (tolerance parity, tests/test_gpu_sparse.py); every sum has a fixed order: bit-reproducible.
//
// Entry lists hold BOTH triangles of every A_i (the trace inner products above run over the
// full matrix).  Layout per group: entries of (member, i) at [eptr[mem*m+i], eptr[mem*m+i+1]),
// erc = row | col << 16, eval = value.  The slack of PrepareStep uses a second, position-major
// copy (kernels_lmi.hip.h / kernels_lmi_large.hip.h: sp_pptr / sp_pvar / sp_pval) so that every
// entry of sum_i y_i A_i is accumulated by one thread in the reference's order of i.
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

// Sparse evaluation costs about 2 (sum nnz)^2 multiply-adds at a fraction of the dense kernels'
// efficiency; it is chosen when that is at least 2x fewer operations than the dense formula.
inline bool LmiSparsePays(int n, int m, double nnz_a) {
  const double dense = 4.0 * n * n * n * (m + 1.0) + (double)n * n * m * m;
  return 4.0 * nnz_a * nnz_a <= dense;
}

__device__ __forceinline__ void PairFromIndex(long long t, int* i, int* j) {
  // t = i (i + 1) / 2 + j, 0 <= j <= i
  long long ii = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ii * (ii + 1) / 2 > t) ii--;
  while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
  *i = (int)ii;
  *j = (int)(t - ii * (ii + 1) / 2);
}

// grid (count, chunks).  SMALL: W (and X = W C W, computed here by chunk 0) live in LDS;
// otherwise W is read from HBM/L2 and X comes from two GEMM launches (Xg: count x n^2).
// WAVE_PER_PAIR: a wavefront splits the nnz_i x nnz_j terms of one pair (heavier matrices);
// otherwise one thread sums one pair.
template <bool SMALL, bool WAVE_PER_PAIR>
__global__ void __launch_bounds__(256) lmi_schur_sparse(LmiGroup g, Arena ar, const double* __restrict__ Xg) {
  extern __shared__ double lds[];
  const int n = g.n, m = g.m, nn = n * n;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* Cm = g.C + (size_t)mem * nn;
  const double* Wg = g.W + (size_t)mem * nn;
  const int* eptr = g.sp_eptr + (size_t)mem * m;
  const int* erc = g.sp_erc;
  const double* eval = g.sp_eval;
  double* G = ar.G + ar.g_off[id];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const double osc = g.herm_d > 1 ? 1.0 / g.herm_d : 1.0;
  const bool first = blockIdx.y == 0;
  const double* W = Wg;
  const double* X = Xg ? Xg + (size_t)mem * nn : nullptr;
  if (SMALL) {
    double* sW = lds;
    for (int q = threadIdx.x; q < nn; q += blockDim.x) sW[q] = Wg[q];
    W = sW;
    if (first) {
      double* sC = sW + nn;
      double* sP = sC + nn;
      double* sX = sP + nn;
      for (int q = threadIdx.x; q < nn; q += blockDim.x) sC[q] = Cm[q];
      __syncthreads();
      LdsGemm(n, sC, sW, sP);  // C W
      __syncthreads();
      LdsGemm(n, sW, sP, sX);  // W C W
      X = sX;
    }
    __syncthreads();
  }
  if (first) {
    // residual vectors: one wavefront per variable
    for (int i = wave; i < m; i += nwaves) {
      double aw = 0, aq = 0;
      for (int e = eptr[i] + lane; e < eptr[i + 1]; e += 64) {
        const int rc = erc[e], r = rc & 0xffff, c = rc >> 16;
        const double a = eval[e];
        aw = fma(a, W[c + (size_t)r * n], aw);
        aq = fma(a, X[c + (size_t)r * n], aq);
      }
      aw = WaveSum(aw);
      aq = WaveSum(aq);
      if (lane == 0) {
        ar.AWc[ar.r_off[id] + i] = aw * osc;
        ar.AQcc[ar.r_off[id] + i] = aq * osc;
      }
    }
    if (wave == 0) {
      double wc = 0, cq = 0;
      for (int q = lane; q < nn; q += 64) {
        const double c = Cm[q];
        wc = fma(c, W[q], wc);
        cq = fma(c, X[q], cq);
      }
      wc = WaveSum(wc);
      cq = WaveSum(cq);
      if (lane == 0) {
        ar.sc[2 * id] = wc * osc;
        ar.sc[2 * id + 1] = cq * osc;
      }
    }
  }
  const long long pairs = (long long)m * (m + 1) / 2;
  const long long per = (pairs + gridDim.y - 1) / gridDim.y;
  const long long t0 = per * blockIdx.y, t1 = (t0 + per < pairs) ? t0 + per : pairs;
  if (WAVE_PER_PAIR) {
    for (long long t = t0 + wave; t < t1; t += nwaves) {
      int i, j;
      PairFromIndex(t, &i, &j);
      const int bi = eptr[i], ni = eptr[i + 1] - bi, bj = eptr[j], nj = eptr[j + 1] - bj;
      double s = 0;
      const long long terms = (long long)ni * nj;
      for (long long q = lane; q < terms; q += 64) {
        const int ei = bi + (int)(q / nj), ej = bj + (int)(q % nj);
        const int rc = erc[ei], r = rc & 0xffff, c = rc >> 16;
        const int pq = erc[ej], p = pq & 0xffff, qq = pq >> 16;
        s = fma(eval[ei] * eval[ej], W[c + (size_t)p * n] * W[qq + (size_t)r * n], s);
      }
      s = WaveSum(s);
      if (lane == 0) G[i + (size_t)j * m] = s * osc;
    }
  } else {
    for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
      int i, j;
      PairFromIndex(t, &i, &j);
      const int bi = eptr[i], ei1 = eptr[i + 1], bj = eptr[j], ej1 = eptr[j + 1];
      double s = 0;
      for (int ei = bi; ei < ei1; ei++) {
        const int rc = erc[ei], r = rc & 0xffff, c = rc >> 16;
        const double a = eval[ei];
        const double* Wc = W + c;               // W[c, p] = W[c + p n]
        const double* Wr = W + (size_t)r * n;   // W[q, r] = W[q + r n]
        for (int ej = bj; ej < ej1; ej++) {
          const int pq = erc[ej], p = pq & 0xffff, qq = pq >> 16;
          s = fma(a * eval[ej], Wc[(size_t)p * n] * Wr[qq], s);
        }
      }
      G[i + (size_t)j * m] = s * osc;
    }
  }
}

inline size_t LmiSparseLds(int n) { return sizeof(double) * 4 * (size_t)n * n; }

}  // namespace cxk
