// Sparse-LMI Schur assembly.
//
// The C-ABI builds matrix inequalities entry by entry (CONEX_UpdateLinearOperator,
// hermitian_psd.cc:249-275), so the A_i of real programs hold a handful of nonzeros while the
// reference -- and the dense kernels here -- stream and multiply full n x n matrices.  For a
// constraint whose matrices are sparse enough (see LmiSparsePays) the same quantities
//
//   G(i,j) = tr(W A_i W A_j) = sum_{(r,c,a) in A_i} sum_{(p,q,b) in A_j} a b W[c,p] W[q,r]
//   AW(i)  = tr(A_i W)       = sum_{(r,c,a) in A_i} a W[c,r]
//   AQc(i) = G(i, C),  <c,Qc> = G(C, C),  <w,c> = AW(C)       (C = matrix number m)
//
// (dense_lmi_constraint.cc:72-103, hermitian_psd.cc:171-230) are evaluated from the nonzeros:
// O((sum nnz)^2) instead of O(n^3 m + n^2 m^2), and the m n^2 stream of A disappears from HBM
// (the dense A is never uploaded).  A dense affine term C (more than 64 nonzeros) does not join
// the pair sums: X = W C W is formed densely instead (LDS for small orders, two GEMM launches
// otherwise) and AQc(i) = sum a X[c,r], <c,Qc> = sum C o X, <w,c> = sum C o W.
// Values equal the dense path's up to summation order (tolerance parity,
// tests/test_gpu_sparse.py); every sum has a fixed order: bit-reproducible.
//
// Entry lists hold BOTH triangles of every matrix (the trace inner products above run over the
// full matrix).  Layout per group, m1 = m + 1 lists per member: entries of (member, i) at
// [eptr[mem*m1+i], eptr[mem*m1+i+1]), erc = row | col << 16, eval = value; list m is C (empty
// when C takes the dense route).  The slack of PrepareStep uses a second, position-major copy
// (kernels_lmi.hip.h / kernels_lmi_large.hip.h: sp_pptr / sp_pvar / sp_pval) so that every
// entry of sum_i y_i A_i is accumulated by one thread in the reference's order of i.
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

// The pair sums cost about (sum nnz)^2 / 2 terms; a term is four scattered reads (entry pair, two
// W elements).  Measured term rates on MI355X against the dense kernels' flop rates give the
// break-even ratios below (dense flops per sparse term): ~80 where W sits in LDS and the dense
// side is the fused kernel, ~40 for the other LDS-resident shapes, ~300 beyond LDS (W read from
// L2/HBM, dense side on the MFMA GEMM pipeline at 25-40 TFLOP/s).
inline bool LmiSparsePays(int n, int m, double nnz_a, bool lds_resident, bool fused) {
  const double dense = 4.0 * n * n * n * (m + 1.0) + (double)n * n * m * m;
  const double per_term = lds_resident ? (fused ? 80.0 : 40.0) : 300.0;
  return 0.5 * nnz_a * nnz_a * per_term <= dense;
}

#ifdef CXK_DEBUG_STAMPS
__device__ long long g_sparse_stamp[8];
#define SPSTAMP(i) do { if (blockIdx.x == 500 && blockIdx.y == 0 && threadIdx.x == 0) g_sparse_stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPSTAMP(i) do { } while (0)
#endif

struct SparseLaunch {
  const double* Xg;    // count x n^2: W C W from the GEMM path (large orders with a dense C), else null
  const double* part;  // count x npart x 2: partial sums of <w,c>, <c,Qc> (lmi_dense_c_scalars)
  int npart;
  int emax;            // entries reserved in LDS per workgroup (STAGE)
  int cdense;          // C takes the dense route
};

// Dense C beyond LDS-resident orders: partial sums of <w,c> = sum C o W and <c,Qc> = sum C o X over
// slices of the n^2 positions; grid (npart, count).  The sparse kernel adds them in block order.
__global__ void __launch_bounds__(256) lmi_dense_c_scalars(LmiGroup g, const double* __restrict__ Xg,
                                                           double* __restrict__ part) {
  __shared__ double red[16];
  const int nn = g.n * g.n, mem = blockIdx.y;
  const double* Cm = g.C + (size_t)mem * nn;
  const double* Wg = g.W + (size_t)mem * nn;
  const double* X = Xg + (size_t)mem * nn;
  const int per = (nn + gridDim.x - 1) / gridDim.x;
  const int q0 = per * blockIdx.x, q1 = q0 + per < nn ? q0 + per : nn;
  double wc = 0, cq = 0;
  for (int q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
    const double c = Cm[q];
    wc = fma(c, Wg[q], wc);
    cq = fma(c, X[q], cq);
  }
  wc = BlockSum(wc, red);
  cq = BlockSum(cq, red + 8);
  if (threadIdx.x == 0) {
    part[((size_t)mem * gridDim.x + blockIdx.x) * 2] = wc;
    part[((size_t)mem * gridDim.x + blockIdx.x) * 2 + 1] = cq;
  }
}

// Sum over a group of LPP consecutive lanes (LPP a power of two <= 64) in a fixed butterfly order;
// every lane of the group receives the total.
template <int LPP>
__device__ __forceinline__ double GroupSum(double v) {
  if constexpr (LPP == 4) {  // the same butterfly (xor 2, xor 1) on DPP quad permutations: no LDS crossbar
    v += DppMove<0x4E>(v);   // quad_perm [2,3,0,1]
    v += DppMove<0xB1>(v);   // quad_perm [1,0,3,2]
    return v;
  } else if constexpr (LPP == 2) {
    return v + DppMove<0xB1>(v);
  } else {
#pragma unroll
    for (int d = LPP >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
  }
}

// Sum of the ni x nj terms  a b W[c,p] W[q,r]  of one pair ((r,c,a) from the first list, (p,q,b)
// from the second) shared by LP consecutive lanes.  The longer list goes across the lanes (lane
// `sub` takes its entries sub, sub + LP, ..), every lane walks the whole shorter list, four entries
// at a time: the lane's own entry fixes a column and a row of W (two base addresses), a term is
// then one packed index, one value, two W elements and three multiply-adds -- ~13 instructions
// where a term-by-term walk over the ni x nj rectangle took ~30 (the kernel is bound by instruction
// issue at the sparse-C4 shape: DESIGN 4.8).  sum_i a_i (sum_j b_j w w): fixed order, bit-reproducible;
// every lane of the group returns the total.
template <int LP>
__device__ __forceinline__ double PairSum(const int* erc, const double* eval, const double* W, int n, int bi,
                                          int ni, int bj, int nj, int sub) {
  if (nj > ni) {  // (uniform over the group)
    int t = bi;
    bi = bj;
    bj = t;
    t = ni;
    ni = nj;
    nj = t;
  }
  double s = 0;
  for (int ei = sub; ei < ni; ei += LP) {
    const int rc = erc[bi + ei], r = rc & 0xffff, c = rc >> 16;
    const double a = eval[bi + ei];
    const double* Wc = W + c;                           // W[c + p n]
    const double* Wr = W + __umul24((unsigned)r, (unsigned)n);  // W[q + r n]  (orders < 2^12: 24-bit products, full rate)
    double t = 0;
    int ej = 0;
    for (; ej + 4 <= nj; ej += 4) {
      int pq[4];
      double b[4], w0[4], w1[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        pq[u] = erc[bj + ej + u];
        b[u] = eval[bj + ej + u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        w0[u] = Wc[__umul24((unsigned)(pq[u] & 0xffff), (unsigned)n)];
        w1[u] = Wr[pq[u] >> 16];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) t = fma(b[u], w0[u] * w1[u], t);
    }
    for (; ej < nj; ej++) {
      const int pq = erc[bj + ej];
      t = fma(eval[bj + ej], Wc[__umul24((unsigned)(pq & 0xffff), (unsigned)n)] * Wr[pq >> 16], t);
    }
    s = fma(a, t, s);
  }
  return GroupSum<LP>(s);
}

// grid (count, chunks of the pair range).
// SMALL: W (and, for a dense C, X = W C W computed by chunk 0) live in LDS; otherwise W is read
// from HBM/L2.  LPP lanes share the nnz_i x nnz_j terms of one pair (1: a thread per pair, for
// the sparsest lists; 64: a wavefront per pair) -- a term is two dependent LDS hops (entry, then
// the W elements it names), so the terms of a lane are issued four at a time.  STAGE: the
// constraint's entry list is copied to LDS first: a sweep reads entries at unrelated addresses,
// which costs a cache line per lane from L1/L2 but only bank conflicts from LDS.
template <bool SMALL, int LPP, bool STAGE>
__global__ void __launch_bounds__(256) lmi_schur_sparse(LmiGroup g, Arena ar, SparseLaunch L) {
  extern __shared__ double lds[];
  const int n = g.n, m = g.m, m1 = m + 1, nn = n * n;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* Cm = g.C + (size_t)mem * nn;
  const double* Wg = g.W + (size_t)mem * nn;
  const int* gptr = g.sp_eptr + (size_t)mem * m1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const double osc = g.herm_d > 1 ? 1.0 / g.herm_d : 1.0;
  const bool first = blockIdx.y == 0;
  const bool cdense = L.cdense != 0;
  // LDS: [W | C P X (dense C)] (SMALL), entry values, entry row/col, list pointers
  double* sW = lds;
  double* s_val = lds + (SMALL ? (cdense ? 4 : 1) * (size_t)nn : 0);
  int* s_rc = reinterpret_cast<int*>(s_val + (STAGE ? L.emax : 0));
  int* s_ptr = s_rc + (STAGE ? L.emax : 0);
  SPSTAMP(0);
  const int e0 = gptr[0];
  for (int q = threadIdx.x; q <= m1; q += blockDim.x) s_ptr[q] = gptr[q] - (STAGE ? e0 : 0);
  const int* erc = g.sp_erc;
  const double* eval = g.sp_eval;
  if (STAGE) {
    const int cnt = gptr[m1] - e0;
    for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
      s_val[q] = g.sp_eval[e0 + q];
      s_rc[q] = g.sp_erc[e0 + q];
    }
    erc = s_rc;
    eval = s_val;
  }
  const double* W = Wg;
  if (SMALL) {
    for (int q = threadIdx.x; q < nn; q += blockDim.x) sW[q] = Wg[q];
    W = sW;
  }
  __syncthreads();
  SPSTAMP(1);
  // dense C: X = W C W (LDS product here for small orders, GEMM launches before this kernel otherwise)
  const double* X = (cdense && L.Xg) ? L.Xg + (size_t)mem * nn : nullptr;
  if (cdense && SMALL) {
    double* sC = sW + nn;
    double* sP = sC + nn;
    double* sX = sP + nn;
    for (int q = threadIdx.x; q < nn; q += blockDim.x) sC[q] = Cm[q];
    __syncthreads();
    LdsGemm(n, sC, sW, sP);  // C W
    __syncthreads();
    LdsGemm(n, sW, sP, sX);  // W C W
    __syncthreads();
    X = sX;
    if (first) {
      double wc = 0, cq = 0;
      for (int q = threadIdx.x; q < nn; q += blockDim.x) {
        const double c = sC[q];
        wc = fma(c, W[q], wc);
        cq = fma(c, X[q], cq);
      }
      double* red = reinterpret_cast<double*>(s_ptr + ((m1 + 3) & ~1));
      wc = BlockSum(wc, red);
      cq = BlockSum(cq, red + 8);
      if (threadIdx.x == 0) {
        ar.sc[2 * id] = wc * osc;
        ar.sc[2 * id + 1] = cq * osc;
      }
    }
  } else if (cdense && first && threadIdx.x == 0) {
    // <w,c>, <c,Qc>: partial sums of lmi_dense_c_scalars, added in block order
    double wc = 0, cq = 0;
    for (int k = 0; k < L.npart; k++) {
      wc += L.part[((size_t)mem * L.npart + k) * 2];
      cq += L.part[((size_t)mem * L.npart + k) * 2 + 1];
    }
    ar.sc[2 * id] = wc * osc;
    ar.sc[2 * id + 1] = cq * osc;
  }
  {
    // AW(i) = sum a W[c,r] (list m, a sparse C, gives <w,c>) and, for a dense C, AQc(i) = sum a X[c,r]:
    // LPV lanes per list, the lists dealt to all chunks
    const int lists = cdense ? m : m1;
    constexpr int LPV = LPP < 8 ? 8 : LPP;
    const int gpb = blockDim.x / LPV;
    for (int i = blockIdx.y * gpb + threadIdx.x / LPV; i < lists; i += gridDim.y * gpb) {
      double aw = 0, aq = 0;
      for (int e = s_ptr[i] + (threadIdx.x % LPV); e < s_ptr[i + 1]; e += LPV) {
        const int rc = erc[e], r = rc & 0xffff, c = rc >> 16;
        const double a = eval[e];
        aw = fma(a, W[c + (size_t)r * n], aw);
        if (cdense) aq = fma(a, X[c + (size_t)r * n], aq);
      }
      aw = GroupSum<LPV>(aw);
      if (cdense) aq = GroupSum<LPV>(aq);
      if (threadIdx.x % LPV == 0) {
        if (i < m) {
          ar.AWc[ar.r_off[id] + i] = aw * osc;
          if (cdense) ar.AQcc[ar.r_off[id] + i] = aq * osc;
        } else {
          ar.sc[2 * id] = aw * osc;
        }
      }
    }
  }
  SPSTAMP(2);
  // pair sums: LPP lanes per pair, the pair range dealt to the chunks; pair t -> (i, j) from the
  // group's table.  Row m (a sparse C: at most 64 nonzeros) rides in the same loop:
  // AQc(j) = G(C, A_j), <c,Qc> = G(C, C).
  double* G = ar.G + ar.g_off[id];
  {
    const long long pairs = cdense ? (long long)m * (m + 1) / 2 : (long long)m1 * (m1 + 1) / 2;
    const long long per = (pairs + gridDim.y - 1) / gridDim.y;
    const long long t0 = per * blockIdx.y, t1 = (t0 + per < pairs) ? t0 + per : pairs;
    const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP, ngrp = blockDim.x / LPP;
    long long t = t0 + grp;
    int ij_next = t < t1 ? g.sp_pairs[t] : 0;
    for (; t < t1; t += ngrp) {
      const int ij = ij_next, i = ij & 0xffff, j = ij >> 16;
      ij_next = t + ngrp < t1 ? g.sp_pairs[t + ngrp] : 0;  // (asked for a pass ahead: off the chain)
      const int bi = s_ptr[i], bj = s_ptr[j];
      const double s = PairSum<LPP>(erc, eval, W, n, bi, s_ptr[i + 1] - bi, bj, s_ptr[j + 1] - bj, sub);
      if (sub == 0) {
        if (i < m)
          G[i + (size_t)j * m] = s * osc;
        else if (j < m)
          ar.AQcc[ar.r_off[id] + j] = s * osc;
        else
          ar.sc[2 * id + 1] = s * osc;
      }
    }
  }
#ifdef CXK_DEBUG_STAMPS
  SPSTAMP(3);
  __syncthreads();
  SPSTAMP(4);
#endif
}

// bytes of LDS: matrices, staged entries, list pointers (+ block-reduction scratch)
inline size_t LmiSparseLds(int n, int m, bool small, bool cdense, int emax_staged) {
  size_t b = small ? sizeof(double) * (cdense ? 4 : 1) * (size_t)n * n : 0;
  b += (size_t)emax_staged * 12;
  b += sizeof(int) * (size_t)((m + 1 + 3) & ~1) + sizeof(double) * 16;
  return b;
}

}  // namespace cxk
