// Supernodes whose panel does not fit LDS (more than ~140 columns): blocked, HBM-resident
// factorization and solves, driven from the host one supernode at a time.  Same mathematics as
// the in-kernel paths (reference BlockCholeskyInPlace block_triangular_operations.cc:184-219 and
// the block solves :114-182), organised as a right-looking blocked Cholesky with 32-column
// panels: one kernel per panel factors the 32 x 32 diagonal block (every workgroup redundantly,
// row per lane in registers: no dependent launch) and solves its rows, every O(n^3) update is
// a batched fp64 MFMA GEMM (kernels_gemm.hip.h); the triangular solves with the factored block
// stream it once through ONE workgroup (big_solve_fwd / big_solve_bwd):
//     L21  = A21 L11^-T            (big_panel)
//     A22 -= L21 L21^T             (GEMM NT, lower only)          -- the SYRK of north_star
//     off[k-block,:] = L11^-1 off[k-block,:]; off[below,:] -= L21 off[k-block,:]   (GEMM NN)
//     U = off^T off, t = off^T b   (GEMM TN), scattered to the consumer slots
// Storage is the slab itself: diag block ns x ns column-major, off block ns x s column-major.
#pragma once
#include "big_chol.h"
#include "big_panel_solve.hip.h"
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

// Panel step of the blocked factorization; one WAVEFRONT per workgroup, one work item per lane.
// Every wavefront factors the nb x nb diagonal block at (k0, k0) itself -- row per lane, the
// register elimination of the small-supernode kernels (ElimSteps) -- so the triangular solves
// below it need no second, dependent launch and no LDS: L[j][i] reaches the FMAs as a scalar
// (v_readlane of column i at lane j).  Workgroup 0 stores the factor.  Work items: rows
// r >= k0 + nb of the panel ( x <- x L11^-T ) and columns c of the off block
// ( off[k-block, c] <- L11^-1 off[k-block, c] ), the 32 unknowns of an item in registers.
__global__ void __launch_bounds__(64) big_panel(double* __restrict__ D, double* __restrict__ B, int ns, int s,
                                                int k0, int nb, int* __restrict__ fail) {
  constexpr int NB = kBigNB;
  const int lane = threadIdx.x;
  const int l32 = lane & 31;  // lanes 32..63 mirror lanes 0..31 through the elimination: every
                              // 16-lane DPP row then holds half of a column of L (see the solves)
  double a[NB + 1];           // a[j] = L[l32][j] (rows >= nb: unit rows)
#ifdef CXK_DEBUG_STAMPS
#define BPSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && k0 == 64) g_cxk_stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BPSTAMP(i) do { } while (0)
#endif
  BPSTAMP(0);
  const bool row = l32 < nb;
  {
    const double* src = D + (k0 + (row ? l32 : 0)) + (size_t)k0 * ns;
#pragma unroll
    for (int j = 0; j < NB; j++) a[j] = (row && j <= l32) ? src[(size_t)(j < nb ? j : 0) * ns] : 0.0;
  }
  // this lane's work item, loaded while the elimination runs: element e at base[e * st]
  // (a row of A21 steps by ns, a column of the off block by 1)
  const int below = ns - k0 - nb;
  const int w = blockIdx.x * 64 + lane;
  const bool is_row = w < below, is_col = !is_row && w < below + s;
  double* base = is_row ? D + (k0 + nb + w) + (size_t)k0 * ns
                        : B + (size_t)(is_col ? w - below : 0) * ns + k0;
  const size_t st = is_row ? (size_t)ns : 1;
  double x[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) x[j] = ((is_row || is_col) && j < nb) ? base[j * st] : 0.0;
#pragma unroll
  for (int j = 0; j < NB; j++)
    if (j >= nb && l32 == j) a[j] = 1.0;  // padding pivots
  a[NB] = 0.0;
  bool bad = false;
  BPSTAMP(1);
  ElimSteps<NB, 0, 0>::run(a, l32, bad, nb);
  BPSTAMP(2);
  if (bad && lane == 0) atomicExch(fail, 1);
  if (blockIdx.x == 0 && row && lane < NB) {
    double* dst = D + (k0 + lane) + (size_t)k0 * ns;
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (j <= lane) dst[(size_t)j * ns] = a[j];
  }
  BPSTAMP(3);
  // Both kinds of item are forward substitutions with L11: x_j = (x_j - sum_{i<j} L[j][i] x_i) / L[j][j],
  // column by column: once x_i is final every later unknown takes its term (i ascending per unknown).
  // L[j][i] is lane j of column a[i]: with DPP rows 0/2 (1/3) mirrored into every row it reaches
  // the fma as a row_newbcast operand -- one instruction per term, as in the elimination itself.
  double diag = 1.0;
#pragma unroll
  for (int j = 0; j < NB; j++) diag = (l32 == j) ? a[j] : diag;
  const double dinv = 1.0 / diag;  // lane j: 1 / L[j][j]
  BigPanelSolve<NB, 0>::run(a, x, dinv);
  BPSTAMP(4);
  if (is_row || is_col) {
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (j < nb) base[j * st] = x[j];
  }
  BPSTAMP(5);
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads()
// also waits for every outstanding global load (vmcnt(0)), i.e. for the operands requested a phase
// ahead below -- which is the round trip the request was issued early to hide.
__device__ __forceinline__ void BigLdsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// b <- L^-1 b for the factored diagonal block of a big supernode: ONE workgroup streams L once,
// panel by panel (b lives in LDS), the mirror image of big_solve_bwd below.  Wavefront 0 solves the
// 32 x 32 blocks (row per lane in registers).  Of the update  b_below -= L21 y_k  only the 32 rows of
// the NEXT panel are on the dependent chain: wavefronts 1 .. 8 form them as 32 dot products (sixteen
// lanes per row) between two solves, and apply the rest of the update (one row per thread, 32
// multiply-adds) while wavefront 0 solves the next block.  L does not depend on b: operands are
// requested a panel ahead, the barriers order LDS traffic only.
constexpr int kBigFwdThreads = 64 + 32 * 16, kBigFwdRows = 32 * 16;
__global__ void __launch_bounds__(kBigFwdThreads) big_solve_fwd(const double* __restrict__ D,
                                                                double* __restrict__ b, int ns) {
  constexpr int NB = kBigNB;
  extern __shared__ double sb[];  // ns
  __shared__ double yk[NB];
  const int lane = threadIdx.x & 63;
  const bool solver = threadIdx.x < 64;
  const int ut = (int)threadIdx.x - 64, g = ut >> 4, sub = ut & 15;  // row g of the next panel, 16 lanes each
  for (int i = threadIdx.x; i < ns; i += blockDim.x) sb[i] = b[i];
  // solver: rg[j] = L[k0 + lane][k0 + j].  Others, for panel k's columns: rg[j] = L[k0 + 64 + ut][k0 + j]
  // (a row beyond the next panel), near[h] = L[k0 + 32 + g][k0 + sub + 16 h] (a row of the next panel)
  double rg[NB], near_cur[2], near_next[2];
  near_cur[0] = near_cur[1] = near_next[0] = near_next[1] = 0.0;
  auto request_block = [&](int k0) {
    if (k0 >= ns) return;
    const int nb = ns - k0 < NB ? ns - k0 : NB;
    const double* src = D + (k0 + (lane < nb ? lane : 0)) + (size_t)k0 * ns;
#pragma unroll
    for (int j = 0; j < NB; j++) rg[j] = src[(size_t)(j < nb ? j : 0) * ns];
  };
  // near_next <- the rows of the panel behind columns c0 .. c0 + 31 (used two phases 1 later)
  auto request_near = [&](int c0) {
    if (c0 + NB >= ns) return;
    const int below = ns - c0 - NB;
    const double* nrow = D + (c0 + NB + (g < below ? g : 0)) + (size_t)c0 * ns;
#pragma unroll
    for (int h = 0; h < 2; h++) near_next[h] = nrow[(size_t)(sub + 16 * h) * ns];
  };
  // rg <- this thread's row beyond the next panel, columns k0 .. k0 + 31 (used while the next block is solved)
  auto request_far = [&](int k0) {
    if (k0 + NB >= ns) return;
    const int below = ns - k0 - NB;
    const double* frow = D + (k0 + NB + (32 + ut < below ? 32 + ut : 0)) + (size_t)k0 * ns;
#pragma unroll
    for (int j = 0; j < NB; j++) rg[j] = frow[(size_t)j * ns];
  };
  if (solver)
    request_block(0);
  else
    request_near(0);
  BigLdsBarrier();
  for (int k0 = 0; k0 < ns; k0 += NB) {
    const int nb = ns - k0 < NB ? ns - k0 : NB;
    if (!solver && k0 > 0) {
      // the rows of THIS panel take the term of the panel solved last (everything older is in already)
      double acc = 0.0;
#pragma unroll
      for (int h = 0; h < 2; h++) acc = fma(near_cur[h], yk[sub + 16 * h], acc);
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
      if (sub == 0 && g < nb) sb[k0 + g] -= acc;
    }
    BigLdsBarrier();
    if (solver) {
      const bool row = lane < nb;
      double l[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) l[j] = (row && j <= lane && j < nb) ? rg[j] : (j == lane ? 1.0 : 0.0);
      double v = row ? sb[k0 + lane] : 0.0;
      double diag = 1.0;
#pragma unroll
      for (int j = 0; j < NB; j++) diag = (lane == j) ? l[j] : diag;
      const double dinv = 1.0 / diag;
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const double yj = ReadLane(v, j) * ReadLane(dinv, j);
        if (lane == j)
          v = yj;
        else if (lane > j)
          v = fma(-l[j], yj, v);
      }
      // (yk of the previous panel is still being read by the others in this phase: written after the barrier)
      BigLdsBarrier();
      if (row) {
        sb[k0 + lane] = v;
        yk[lane] = v;
      } else if (lane < NB) {
        yk[lane] = 0.0;
      }
      request_block(k0 + NB);
    } else {
      if (k0 > 0) {
        // meanwhile: the previous panel's term for the rows beyond this panel
        const int kp = k0 - NB, belowp = ns - kp - NB;
        if (32 + ut < belowp) {
          double acc = sb[kp + NB + 32 + ut];
#pragma unroll
          for (int j = 0; j < NB; j++) acc = fma(-rg[j], yk[j], acc);
          sb[kp + NB + 32 + ut] = acc;
        }
        for (int r = 32 + ut + kBigFwdRows; r < belowp; r += kBigFwdRows) {  // (very tall blocks: loaded here)
          const double* frow = D + (kp + NB + r) + (size_t)kp * ns;
          double acc = sb[kp + NB + r];
#pragma unroll 8
          for (int j = 0; j < NB; j++) acc = fma(-frow[(size_t)j * ns], yk[j], acc);
          sb[kp + NB + r] = acc;
        }
      }
      BigLdsBarrier();
#pragma unroll
      for (int h = 0; h < 2; h++) near_cur[h] = near_next[h];  // (requested a panel ago)
      request_near(k0 + NB);
      request_far(k0);  // this panel's columns: used once it is solved, behind the next solve
    }
    BigLdsBarrier();
  }
  for (int i = threadIdx.x; i < ns; i += blockDim.x) b[i] = sb[i];
}

// b <- L^-T b, panels from the last to the first.  Wavefront 0 solves the transposed 32 x 32 blocks
// (column per lane in registers); wavefronts 1 .. 8 form the 32 dot products L21^T x_below, sixteen
// lanes per column (16 consecutive rows per load instruction), reduced in a fixed order.
// Two things keep a panel at "its own arithmetic plus two barriers":
//  * L does not depend on b, so every operand is requested a whole panel ahead, and the barriers
//    order LDS traffic only (BigLdsBarrier: __syncthreads() would wait for those requests);
//  * of a panel's dot products only the terms of the 32 rows solved LAST are on the dependent chain:
//    the terms of all earlier-solved rows are summed while wavefront 0 solves the panel in between.
constexpr int kBigBwdThreads = 64 + 32 * 16, kBigBwdU = 28, kBigBwdRows = 32 + 16 * kBigBwdU;
__global__ void __launch_bounds__(kBigBwdThreads) big_solve_bwd(const double* __restrict__ D,
                                                                double* __restrict__ b, int ns) {
  constexpr int NB = kBigNB, U = kBigBwdU;
  extern __shared__ double sb[];  // ns
  __shared__ double part[NB];
  const int lane = threadIdx.x & 63;
  const bool solver = threadIdx.x < 64;
  const int j = ((int)threadIdx.x - 64) >> 4, sub = ((int)threadIdx.x - 64) & 15;  // column j of a panel, 16 lanes each
  for (int i = threadIdx.x; i < ns; i += blockDim.x) sb[i] = b[i];
  const int nblk = (ns + NB - 1) / NB;
  // solver: c[k] = L[k0 + k][k0 + lane].  Others, for the panel whose dot products they are forming:
  // w[.][h] = L[k0 + nb + sub + 16 h][k0 + j], h = 0, 1 (the 32 rows right below the panel: solved last),
  // v[u] = the rows 32 + sub + 16 u below it
  // one register array for both roles: c = rg; v[u] = rg[u], w of this panel rg[U + h], of the next rg[U + 2 + h]
  double rg[NB];
  static_assert(kBigBwdU + 4 == kBigNB, "v and w fill the array");
#define c rg
#define v rg
#define w_cur(h_) rg[U + (h_)]
#define w_next(h_) rg[U + 2 + (h_)]
  double old_part = 0.0;  // the dot product over the rows 32 .. below - 1 of the panel coming next
  auto request_block = [&](int kb) {
    if (kb < 0) return;
    const int k0 = kb * NB, nb = ns - k0 < NB ? ns - k0 : NB;
    const double* src = D + k0 + (size_t)(k0 + (lane < nb ? lane : 0)) * ns;
#pragma unroll
    for (int k = 0; k < NB; k++) c[k] = src[k < nb ? k : 0];
  };
  auto request_rows = [&](int kb) {  // (never the last panel: nb = 32)
    if (kb < 0) return;
    const int k0 = kb * NB, below = ns - k0 - NB;
    const double* colp = D + (k0 + NB) + (size_t)(k0 + j) * ns;
#pragma unroll
    for (int h = 0; h < 2; h++) w_next(h) = colp[sub + 16 * h < below ? sub + 16 * h : 0];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = colp[32 + sub + 16 * u < below ? 32 + sub + 16 * u : 0];
  };
  if (solver)
    request_block(nblk - 1);
  else
    request_rows(nblk - 2);
  BigLdsBarrier();
  for (int kb = nblk - 1; kb >= 0; kb--) {
    const int k0 = kb * NB, nb = ns - k0 < NB ? ns - k0 : NB, below = ns - k0 - nb;
    if (!solver) {
      // the panel's dot products: what was summed during the previous solve plus the rows solved by it
      double acc = old_part;
      if (below > 0) {
#pragma unroll
        for (int h = 0; h < 2; h++)
          if (sub + 16 * h < below) acc = fma(w_cur(h), sb[k0 + nb + sub + 16 * h], acc);
      }
#pragma unroll
      for (int h = 0; h < 2; h++) w_cur(h) = w_next(h);  // (requested a panel ago; the next request overwrites w_next)
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
      if (sub == 0) part[j] = acc;
    }
    BigLdsBarrier();
    if (solver) {
      const bool colv = lane < nb;
#pragma unroll
      for (int k = 0; k < NB; k++) c[k] = (colv && k >= lane && k < nb) ? c[k] : (k == lane ? 1.0 : 0.0);
      double x = colv ? sb[k0 + lane] - (below > 0 ? part[lane] : 0.0) : 0.0;
      double diag = 1.0;
#pragma unroll
      for (int k = 0; k < NB; k++) diag = (lane == k) ? c[k] : diag;
      const double dinv = 1.0 / diag;
#pragma unroll
      for (int k = NB - 1; k >= 0; k--) {
        const double xk = ReadLane(x, k) * ReadLane(dinv, k);
        if (lane == k)
          x = xk;
        else if (lane < k)
          x = fma(-c[k], xk, x);
      }
      if (colv) sb[k0 + lane] = x;
      request_block(kb - 1);
    } else if (kb >= 1) {
      // meanwhile, for the NEXT panel (kb - 1): its rows from 32 on, whose unknowns are all solved
      const int k1 = k0 - NB, below1 = ns - k1 - NB;
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < U; u++)
        if (32 + sub + 16 * u < below1) acc = fma(v[u], sb[k1 + NB + 32 + sub + 16 * u], acc);
      if (below1 > kBigBwdRows) {  // (very tall blocks: the rest is loaded here)
        const double* colp = D + (k1 + NB) + (size_t)(k1 + j) * ns;
        for (int r = kBigBwdRows + sub; r < below1; r += 16) acc = fma(colp[r], sb[k1 + NB + r], acc);
      }
      old_part = acc;
      request_rows(kb - 2);
    }
    BigLdsBarrier();
  }
#undef c
#undef v
#undef w_cur
#undef w_next
  for (int i = threadIdx.x; i < ns; i += blockDim.x) b[i] = sb[i];
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
// df_flags != nullptr: the factor sweep of a supernode that fits big_chol_dataflow (big_chol.h) takes
// that ONE launch for the panel loop and the forward substitution; *df_gen counts its launches.
inline hipError_t BigSupernodeSweep(const FactorPlan& P, const SnRec& R, int mode, double* slab, double* rhs,
                                    int* fail, double* ws, hipStream_t st, int* df_flags = nullptr,
                                    int* df_gen = nullptr) {
  const int ns = R.ns, s = R.nsep;
  double* D = slab + R.diag_off;
  double* B = slab + R.offd_off;
  double* b = rhs ? rhs + R.start : nullptr;
  hipError_t e;
  const size_t solve_lds = sizeof(double) * (size_t)ns;
  if (solve_lds > 150 * 1024) return hipErrorNotSupported;  // right-hand side block held in LDS
  static PerDeviceOnce once;
  e = once.run([] {
    const int lim = 150 * 1024;
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&big_solve_fwd),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    if (e2 != hipSuccess) return e2;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&big_solve_bwd),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  });
  if (e != hipSuccess) return e;
  if (mode == 2) {
    if (R.bs_end > R.bs_beg) big_backsep<<<(ns + 255) / 256, 256, 0, st>>>(P, R, slab, rhs);
    big_solve_bwd<<<1, kBigBwdThreads, solve_lds, st>>>(D, b, ns);
    return hipGetLastError();
  }
  const int with_matrix = mode == 0;
  if ((with_matrix && R.tg_end > R.tg_beg) || (rhs && R.mf > 0)) big_pull<<<64, 256, 0, st>>>(P, R, slab, rhs, with_matrix);
  const bool dataflow = with_matrix && df_flags && df_gen && BigCholSupports(ns, s);
  if (dataflow) {
    BigCholArgs ca;
    ca.D = D;
    ca.B = B;
    ca.b = b;
    ca.ns = ns;
    ca.s = s;
    ca.flags = df_flags;
    ca.gen = *df_gen = (*df_gen >= (1 << 30)) ? 1 : *df_gen + 1;
    ca.fail = fail;
    if ((e = LaunchBigChol(ca, st)) != hipSuccess) return e;
  } else if (with_matrix)
    for (int k0 = 0; k0 < ns; k0 += kBigNB) {
      const int nb = std::min(kBigNB, ns - k0), below = ns - k0 - nb;
      const int items = below + s;
      big_panel<<<std::max(1, (items + 63) / 64), 64, 0, st>>>(D, B, ns, s, k0, nb, fail);
      if (below > 0) {
        const double* L21 = D + (k0 + nb) + (size_t)k0 * ns;
        GemmArgs g1 = BigGemm(below, below, nb, L21, ns, L21, ns, D + (k0 + nb) + (size_t)(k0 + nb) * ns, ns,
                              -1.0, 1.0, 1);
        if ((e = LaunchGemm(g1, false, true, 1, st)) != hipSuccess) return e;
        if (s > 0) {
          GemmArgs g2 = BigGemm(below, s, nb, L21, ns, B + k0, ns, B + k0 + nb, ns, -1.0, 1.0, 0);
          if ((e = LaunchGemm(g2, false, false, 1, st)) != hipSuccess) return e;
        }
      }
    }
  if (rhs && !dataflow) big_solve_fwd<<<1, kBigFwdThreads, solve_lds, st>>>(D, b, ns);
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
