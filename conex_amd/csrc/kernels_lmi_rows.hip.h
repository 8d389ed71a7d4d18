// TakeStep for small dense LMIs, one WAVEFRONT per constraint, one matrix row per lane, everything
// in registers.  Compiled for register capacities N = 20 and 32 (the order n <= N is a run-time
// value: rows and columns beyond n are padding); a column spans up to two 16-lane DPP rows.  Reference: TakeStep -> GeodesicUpdate with the Pade [3/3] exponential
// (psd_constraint.cc:45-84, exponential_map_pade.cc:10-32):
//     X = (WS + e I) * step;  A2 = X X;  U = X (A2 + 60 I);  V = 12 A2 + 120 I
//     E = (V - U)^-1 (V + U);  W <- sym(E W)
//
// The matrix products are row-times-matrix sums  out[c] = sum_j row[j] M[j][c]  in which M[j][c] is
// lane j's register c: it reaches the fma as a row_newbcast DPP operand (the mirror of DPP row 0
// serves j < 16, the mirror of row 1 the rest), one instruction per term, j ascending -- the same
// fma chains as the LDS kernel (lmi_take_step_generic) up to the linear solve.  The solve is a
// Gauss-Jordan elimination with partial pivoting on [V - U | V + U]: the pivot row stays in its
// lane (an implicit permutation), its entries reach the other rows by v_readlane.  The workgroup
// kernel spends most of its 64 us at the C4 shape in barriers and in serial LDS chains (pivot
// search, back substitution); this one needs no barrier at all.  Mathematically the same update;
// the elimination order differs from an LU + triangular solves, so W agrees with the oracle to
// rounding (tests/test_gpu_parity.py, <= 1e-11).
#pragma once
#include "kernels_cone.hip.h"
#include "kernels_kkt.hip.h"
#include "kernels_lmi.hip.h"

namespace cxk {

// acc += w_bcast * v, W operand = lane J of each 16-lane row of `w` (DPP row_newbcast: the one DPP
// mode gfx90a+ allows on fp64).
template <int J>
__device__ __forceinline__ void FmaBcast(double& acc, double w, double v) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(w), "v"(v), "n"(J));
}

template <int N, int J>
struct RowDotSteps {  // acc += sum_{j >= J} M[j][c] row[j]
  static __device__ __forceinline__ void run(double& acc, double m0, double m1, const double (&row)[N]) {
    if constexpr (J < N) {
      if constexpr (J < 16)
        FmaBcast<J>(acc, m0, row[J]);
      else
        FmaBcast<J - 16>(acc, m1, row[J]);
      RowDotSteps<N, J + 1>::run(acc, m0, m1, row);
    }
  }
};

// The same for W columns at once, term j of every column before term j + 1 of any: each column's sum
// is one dependent chain (an fp64 fma can issue every 4 cycles, its result is back after ~8), so one
// column at a time leaves every other issue slot empty.  Same chains, same bits.
template <int N, int W, int J>
struct RowDotStepsWide {
  static __device__ __forceinline__ void run(double (&acc)[W], const double (&m0)[W], const double (&m1)[W],
                                             const double (&row)[N]) {
    if constexpr (J < N) {
#pragma unroll
      for (int q = 0; q < W; q++) {
        if constexpr (J < 16)
          FmaBcast<J>(acc[q], m0[q], row[J]);
        else
          FmaBcast<J - 16>(acc[q], m1[q], row[J]);
      }
      RowDotStepsWide<N, W, J + 1>::run(acc, m0, m1, row);
    }
  }
};

// out[c] = sum_j row[j] * M[j][c], M[j][c] = lane j's mat[c]; columns C .. N-1, kRowColumns at a time
constexpr int kRowColumns = 4;
template <int N, int C>
struct RowTimesMatrix {
  static __device__ __forceinline__ void run(const double (&row)[N], const double (&mat)[N], double (&out)[N]) {
    if constexpr (C + kRowColumns <= N) {
      constexpr int W = kRowColumns;
      double m0[W], m1[W], acc[W];
#pragma unroll
      for (int q = 0; q < W; q++) {
        const RowPair mp = Swap16(mat[C + q]);  // a: DPP rows 0/2 everywhere, b: rows 1/3
        m0[q] = mp.a;
        m1[q] = mp.b;
        acc[q] = 0.0;
      }
#pragma unroll
      for (int q = 0; q < W; q++) DppOperandFence(m0[q], m1[q], acc[q]);
      RowDotStepsWide<N, W, 0>::run(acc, m0, m1, row);  // rows / columns beyond the order hold zeros
#pragma unroll
      for (int q = 0; q < W; q++) out[C + q] = acc[q];
      RowTimesMatrix<N, C + W>::run(row, mat, out);
    } else if constexpr (C < N) {
      const RowPair mp = Swap16(mat[C]);
      double m0 = mp.a, m1 = mp.b, acc = 0.0;
      DppOperandFence(m0, m1, acc);
      RowDotSteps<N, 0>::run(acc, m0, m1, row);
      out[C] = acc;
      RowTimesMatrix<N, C + 1>::run(row, mat, out);
    }
  }
};

__device__ __forceinline__ double ReadLaneUniform(double v, int src) {  // src wave-uniform at run time
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

template <int N, int K, int C>
struct GjColumns {  // aug[c] -= f * pivot_row[c], c = C .. 2N-1
  static __device__ __forceinline__ void run(double (&aug)[2 * N], double f, int p) {
    if constexpr (C < 2 * N) {
      aug[C] = fma(-f, ReadLaneUniform(aug[C], p), aug[C]);
      GjColumns<N, K, C + 1>::run(aug, f, p);
    }
  }
};

// Gauss-Jordan steps K .. N-1 with partial pivoting among the rows not used as pivots yet.
template <int N, int K>
struct GjSteps {
  static __device__ __forceinline__ void run(double (&aug)[2 * N], int lane, int n, bool& done, double& mypiv,
                                             int& myk) {
    if constexpr (K < N) {
      if (K >= n) return;  // wave-uniform: padding columns
      const bool cand = lane < n && !done;
      const double mag = cand ? fabs(aug[K]) : -1.0;
      const double best = WaveMax(mag);
      unsigned long long bal = __ballot(cand && mag == best);
      if (bal == 0) bal = __ballot(cand);  // NaN column: any remaining row (the result is NaN anyway)
      const int p = __builtin_amdgcn_readfirstlane(__ffsll((long long)bal) - 1);
      const double pk = ReadLaneUniform(aug[K], p);
      const bool is_p = lane == p;
      const double f = is_p ? 0.0 : aug[K] / pk;  // every other row, pivots of earlier steps included
      if (is_p) {
        done = true;
        mypiv = aug[K];
        myk = K;
      }
      GjColumns<N, K, K + 1>::run(aug, f, p);
      GjSteps<N, K + 1>::run(aug, lane, n, done, mypiv, myk);
    }
  }
};

template <int N>
__global__ void __launch_bounds__(256) lmi_take_step_rows(LmiGroup g, StepArgs sa) {
  static_assert(N <= 32, "a column spans DPP rows 0 and 1");
  __shared__ double sT[4][N * N];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mem = blockIdx.x * 4 + wave;
  if (mem >= g.count) return;  // wave-uniform; no workgroup barrier below
  // (asked for here, looked at in front of the store: the flags' round trip stays off the chain)
  const int skip0 = sa.skip_if ? sa.skip_if[0] : 0, skip1 = sa.skip_if ? sa.skip_if[1] : 0;
  const int n = g.n, nn = n * n;
  double* Wg = g.W + (size_t)mem * nn;
  const double* T1 = g.T1 + (size_t)mem * nn;
  const bool row = lane < n;
  const int r = row ? lane : 0;
  const double step = StepSizeOf(sa);
  double x[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    double v = (row && j < n) ? T1[r + j * n] : 0.0;
    if (row && j == r) v += sa.e_weight;
    if (step != 1.0) v *= step;
    x[j] = v;
  }
  double aug[2 * N];
  {
    double a2[N], t[N], u[N];
    RowTimesMatrix<N, 0>::run(x, x, a2);  // A2 = X X
#pragma unroll
    for (int c = 0; c < N; c++) t[c] = a2[c] + ((row && c == r) ? 60.0 : 0.0);
    RowTimesMatrix<N, 0>::run(x, t, u);   // U = X (A2 + 60 I)
#pragma unroll
    for (int c = 0; c < N; c++) {
      const double v = a2[c] * 12.0 + ((row && c == r) ? 120.0 : 0.0);
      aug[c] = -u[c] + v;      // denominator
      aug[N + c] = u[c] + v;   // numerator
    }
  }
  bool done = false;
  double mypiv = 1.0;
  int myk = 0;
  GjSteps<N, 0>::run(aug, lane, n, done, mypiv, myk);
  // this lane now holds row myk of E = denominator^-1 numerator
  double e[N], w[N], ew[N];
#pragma unroll
  for (int c = 0; c < N; c++) e[c] = aug[N + c] / mypiv;
#pragma unroll
  for (int c = 0; c < N; c++) w[c] = (row && c < n) ? Wg[r + c * n] : 0.0;  // lane j: row j of W
  RowTimesMatrix<N, 0>::run(e, w, ew);  // row myk of E W
  double* T = sT[wave];
  if (row) {
#pragma unroll
    for (int c = 0; c < N; c++)
      if (c < n) T[myk + c * n] = ew[c];
  }
  WaveSync();
  const bool skipped = skip0 != 0 || (sa.skip_tag != 0 && skip1 == sa.skip_tag);  // StepSkipped
  if (row && !skipped) {
#pragma unroll
    for (int c = 0; c < N; c++)
      if (c < n) Wg[r + c * n] = (T[r + c * n] + T[c + r * n]) * 0.5;
  }
}

// The same for Hermitian cones (through their real representation of order N): the reference's
// Taylor-squaring exponential  E = ((I + X/4 + X^2/32)^2)^2  (DoExponentialMap,
// exponential_map.cc:15-43), then W <- sym(E W).  Five row-times-matrix products, no linear solve;
// every product is the same fma chain as in lmi_take_step_generic's Hermitian branch.
template <int N>
__global__ void __launch_bounds__(256) lmi_take_step_rows_taylor(LmiGroup g, StepArgs sa) {
  if (StepSkipped(sa)) return;  // (enqueued before the host saw the factorization fail: leave W alone)
  static_assert(N <= 32, "a column spans DPP rows 0 and 1");
  __shared__ double sT[4][N * N];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mem = blockIdx.x * 4 + wave;
  if (mem >= g.count) return;
  const int n = g.n, nn = n * n;
  double* Wg = g.W + (size_t)mem * nn;
  const double* T1 = g.T1 + (size_t)mem * nn;
  const bool row = lane < n;
  const int r = row ? lane : 0;
  const double step = StepSizeOf(sa);
  double x[N], v[N], t[N], y[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    double e = (row && j < n) ? T1[r + j * n] : 0.0;
    if (row && j == r) e += sa.e_weight;
    if (step != 1.0) e *= step;
    x[j] = e;
    v[j] = e * 1.0 / 4.0;
  }
  RowTimesMatrix<N, 0>::run(x, v, t);  // X (X/4)
#pragma unroll
  for (int c = 0; c < N; c++) y[c] = (v[c] + ((row && c == r) ? 1.0 : 0.0)) + t[c] * 0.125;
  RowTimesMatrix<N, 0>::run(y, y, t);  // Y^2
  RowTimesMatrix<N, 0>::run(t, t, y);  // Y^4 = E
#pragma unroll
  for (int c = 0; c < N; c++) v[c] = (row && c < n) ? Wg[r + c * n] : 0.0;  // lane j: row j of W
  RowTimesMatrix<N, 0>::run(y, v, t);  // E W
  double* T = sT[wave];
  if (row) {
#pragma unroll
    for (int c = 0; c < N; c++)
      if (c < n) T[r + c * n] = t[c];
  }
  WaveSync();
  if (row) {
#pragma unroll
    for (int c = 0; c < N; c++)
      if (c < n) Wg[r + c * n] = (T[r + c * n] + T[c + r * n]) * 0.5;
  }
}

// ---------------------------------------------------------------------------------------------
// PrepareStep / GetWeightedSlackEigenvalues for dense LMIs of order N, one wavefront per
// constraint (lmi_prepare_generic gives a constraint a 256-thread workgroup of which three
// wavefronts idle through the Lanczos recurrence).  Every sum below is the sum the workgroup
// kernel forms, in the same order -- the slack (i ascending), WS = W S (k ascending fma chains),
// the Lanczos recurrence of LanczosWave0 with the matrix rows in registers and the vectors one
// value per lane (matrix-vector products as row_newbcast DPP fma chains), the traces with the
// workgroup kernel's partition into 256 partial sums -- so the results are bit-identical.
template <int NR, int J>
struct LanczosDot {  // acc += sum_{k >= J} row[k] * vec_k
  static __device__ __forceinline__ void run(double& acc, double m0, double m1, const double (&row)[NR]) {
    if constexpr (J < NR) {
      if constexpr (J < 16)
        FmaBcast<J>(acc, m0, row[J]);
      else
        FmaBcast<J - 16>(acc, m1, row[J]);
      LanczosDot<NR, J + 1>::run(acc, m0, m1, row);
    }
  }
};

template <int NR>
__device__ __forceinline__ double LanczosMatVec(const double (&row)[NR], double v) {
  const RowPair mp = Swap16(v);  // a: DPP rows 0/2 everywhere (elements 0..15), b: rows 1/3 (16..31)
  double m0 = mp.a, m1 = mp.b, acc = 0.0;
  DppOperandFence(m0, m1, acc);
  LanczosDot<NR, 0>::run(acc, m0, m1, row);
  return acc;
}

// AsymmetricLanczos (approximate_eigenvalues.cc:178-239) as in LanczosWave0, non-Hermitian rules.
template <int N>
__device__ __forceinline__ void LanczosRows(const double (&ws)[N], const double (&wst)[N], const double (&w)[N],
                                            double rvec, int lane, int num_iter, double* ab, double* out,
                                            int n = N) {
  const bool act = lane < n;
  double* alpha = ab;
  double* beta = ab + num_iter + 1;
  // V.col(1) = r ; V.col(0) = W r
  double v1 = act ? rvec : 0.0;
  double v0 = LanczosMatVec<N>(w, v1);
  const double nrm = sqrt(WaveSum(act ? fma(v0, v1, 0.0) : 0.0));
  v0 = v0 / nrm;
  v1 = v1 / nrm;
  if (!act) v0 = v1 = 0.0;
  double p0 = v0, p1 = v1, u0 = 0.0, u1 = 0.0;
  int cnt = 0;
  double beta_prev = 0;
  for (int j = 0; j < num_iter; j++) {
    if (j > 0) {
      const double b2 = WaveSum(act ? fma(u0, u1, 0.0) : 0.0);
      if (b2 < 1e-6) break;
      beta_prev = sqrt(b2);
      if (lane == 0) beta[j - 1] = beta_prev;
      p0 = v0;
      p1 = v1;
      v0 = u0 / beta_prev;
      v1 = u1 / beta_prev;
      if (!act) v0 = v1 = 0.0;
      cnt++;
    }
    u0 = LanczosMatVec<N>(ws, v0);   // WS V.col(0)
    u1 = LanczosMatVec<N>(wst, v1);  // WS^T V.col(1)
    const double a = WaveSum(act ? fma(v0, u1, 0.0) : 0.0);
    if (lane == 0) alpha[j] = a;
    u0 = u0 - a * v0;
    u1 = u1 - a * v1;
    if (j > 0) {
      u0 -= beta_prev * p0;
      u1 -= beta_prev * p1;
    }
  }
  WaveSync();  // alpha / beta were written by lane 0
  TridiagMinMaxWave<N / 2 + 1>(cnt + 1, alpha, beta, &out[0], &out[1]);  // (at most num_iter = N / 2 steps)
}

// EXACT: the order is N (compile-time strides, the packed slack path).  Otherwise any order
// n <= N at run time: the same code with the matrices n apart and the lanes / columns beyond n
// holding zeros (an fma with a zero factor leaves a sum as it is: the same bits as the workgroup
// kernel, which sums over n terms only).
template <int MODE, int N, bool EXACT>
__global__ void __launch_bounds__(256) lmi_prepare_rows(LmiGroup g, StepArgs sa, StepTail tail) {
  static_assert(N > 16 && N <= 32 && (N * N) % 2 == 0, "two DPP rows; 16-byte chunks");
  constexpr int NN = N * N, HALF = NN / 2, CH = (HALF + 63) / 64, ITERS = N / 2;
  __shared__ double sM[4][NN];
  __shared__ double sAB[4][2 * (ITERS + 2)];
  __shared__ double sOut[4][2];
  // with a tail the launch has one workgroup more: it reduces what the others produce and talks to the host
  // (... and, behind it, tail.ny workgroups that write the Newton direction out while it is still in its three parts)
  if (tail.slots && blockIdx.x + 1 + tail.ny >= gridDim.x) {
    if (blockIdx.x + 1 + tail.ny == gridDim.x)
      PrepareTailBlock(tail);
    else
      DirectionBlock(tail, sa, (int)(blockIdx.x + tail.ny - gridDim.x));
    return;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mem = blockIdx.x * 4 + wave;
  if (mem >= g.count) return;  // wave-uniform; no workgroup barrier below
  const int id = g.ids[mem], m = g.m;
  const int n = EXACT ? N : g.n, nn = n * n, half = nn / 2;
  const double c_weight = sa.cw_from ? sa.cw_from[0] * sa.cw_scale : sa.c_weight;
  const double* Cm = g.C + (size_t)mem * nn;
  const double* Wg = g.W + (size_t)mem * nn;
  double* M = sM[wave];
  // minus_s = sum_i y_i A_i - k C (dense_lmi_constraint.cc:8-27)
  const double yv = lane < m ? StepY(sa, sa.cl_perm[sa.cl_ptr[id] + lane]) : 0.0;
  if (EXACT && g.Apk) {
    // from the packed lower triangles (exactly symmetric data: the mirrored entry is the same sum of
    // the same terms): 105 16-byte chunks per matrix instead of 200, lane l owns chunks l and l + 64,
    // twenty matrices per batch -- the whole constraint behind ONE round trip at m <= 20
    constexpr int PK = N * (N + 1) / 2, PH = PK / 2, CP = (PH + 63) / 64, PB = 20;
    static_assert(PK % 2 == 0, "16-byte chunks");
    double2 acc[CP];
    int eo[CP];
#pragma unroll
    for (int u = 0; u < CP; u++) {
      acc[u] = make_double2(0.0, 0.0);
      eo[u] = lane + 64 * u < PH ? lane + 64 * u : PH - 1;
    }
    const double2* base = reinterpret_cast<const double2*>(g.Apk + (size_t)mem * m * PK);
    for (int i0 = 0; i0 < m; i0 += PB) {
      double2 v[PB][CP];
#pragma unroll
      for (int b = 0; b < PB; b++) {
        const int i = i0 + b < m ? i0 + b : m - 1;
#pragma unroll
        for (int u = 0; u < CP; u++) v[b][u] = base[(size_t)i * PH + eo[u]];
      }
#pragma unroll
      for (int b = 0; b < PB; b++) {
        if (i0 + b < m) {  // wave-uniform
          const double yi = ReadLaneUniform(yv, i0 + b);
#pragma unroll
          for (int u = 0; u < CP; u++) {
            acc[u].x += yi * v[b][u].x;
            acc[u].y += yi * v[b][u].y;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CP; u++) {
      const int e2 = lane + 64 * u;
      if (e2 < PH) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          int c = 0, rem = 2 * e2 + h;  // packed position -> (row, column), column by column
          while (rem >= N - c) {
            rem -= N - c;
            c++;
          }
          const int r = c + rem;
          const double a = h == 0 ? acc[u].x : acc[u].y;
          M[r + c * N] = a - c_weight * Cm[r + c * N];
          if (r != c) M[c + r * N] = a - c_weight * Cm[c + r * N];
        }
      }
    }
  } else if (!EXACT && (n & 1)) {
    // an odd order: n^2 doubles per matrix leave every other matrix 8 bytes off a 16-byte boundary --
    // element loads (lane l owns elements l, l + 64, ..), otherwise the loop below
    constexpr int CE = (NN + 63) / 64, BATCH = 6;
    double acc[CE];
    int eo[CE];
#pragma unroll
    for (int u = 0; u < CE; u++) {
      acc[u] = 0.0;
      eo[u] = lane + 64 * u < nn ? lane + 64 * u : nn - 1;
    }
    const double* base = g.A + (size_t)mem * g.a_stride;
    for (int i0 = 0; i0 < m; i0 += BATCH) {
      double v[BATCH][CE];
#pragma unroll
      for (int b = 0; b < BATCH; b++) {
        const int i = i0 + b < m ? i0 + b : m - 1;
#pragma unroll
        for (int u = 0; u < CE; u++) v[b][u] = base[(size_t)i * nn + eo[u]];
      }
#pragma unroll
      for (int b = 0; b < BATCH; b++) {
        if (i0 + b < m) {  // wave-uniform
          const double yi = ReadLaneUniform(yv, i0 + b);
#pragma unroll
          for (int u = 0; u < CE; u++) acc[u] += yi * v[b][u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CE; u++) {
      const int e = lane + 64 * u;
      if (e < nn) M[e] = acc[u] - c_weight * Cm[e];
    }
  } else {
  // lane l owns the 16-byte chunks l, l + 64, ..
  double2 acc[CH];
#pragma unroll
  for (int u = 0; u < CH; u++) acc[u] = make_double2(0.0, 0.0);
  const double2* base = reinterpret_cast<const double2*>(g.A + (size_t)mem * g.a_stride);
  // Ten matrices per batch, every load of a batch issued before its first use, unconditionally
  // from clamped addresses (chunks past the matrix and matrices past m re-read valid data and
  // are masked in the arithmetic): the rolled loop this replaces waited for each matrix's four
  // loads before asking for the next -- twenty dependent round trips per constraint.  Same sums in
  // the same order.
  constexpr int BATCH = 10;
  int eo[CH];
#pragma unroll
  for (int u = 0; u < CH; u++) eo[u] = lane + 64 * u < half ? lane + 64 * u : half - 1;
  for (int i0 = 0; i0 < m; i0 += BATCH) {
    double2 v[BATCH][CH];
#pragma unroll
    for (int b = 0; b < BATCH; b++) {
      const int i = i0 + b < m ? i0 + b : m - 1;
#pragma unroll
      for (int u = 0; u < CH; u++) v[b][u] = base[(size_t)i * half + eo[u]];
    }
#pragma unroll
    for (int b = 0; b < BATCH; b++) {
      if (i0 + b < m) {  // wave-uniform
        const double yi = ReadLaneUniform(yv, i0 + b);
#pragma unroll
        for (int u = 0; u < CH; u++) {
          acc[u].x += yi * v[b][u].x;
          acc[u].y += yi * v[b][u].y;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < CH; u++) {
    const int e = lane + 64 * u;
    if (e < half) {
      M[2 * e] = acc[u].x - c_weight * Cm[2 * e];
      M[2 * e + 1] = acc[u].y - c_weight * Cm[2 * e + 1];
    }
  }
  }
  WaveSync();
  const bool row = lane < n;
  const int r = row ? lane : 0;
  double s[N], w[N], ws[N], wst[N];
#pragma unroll
  for (int c = 0; c < N; c++) {
    const bool in = row && (EXACT || c < n);
    const int at = in ? r + c * n : 0;
    s[c] = in ? M[at] : 0.0;   // lane k: row k of minus_s
    w[c] = in ? Wg[at] : 0.0;  // lane i: row i of W
  }
  RowTimesMatrix<N, 0>::run(w, s, ws);  // WS = W * minus_s, row per lane
  WaveSync();
  if (row) {
    double* T1 = g.T1 + (size_t)mem * nn;
#pragma unroll
    for (int c = 0; c < N; c++)
      if (EXACT || c < n) {
        M[r + c * n] = ws[c];
        if (MODE == 0) T1[r + c * n] = ws[c];
      }
  }
  WaveSync();
#pragma unroll
  for (int c = 0; c < N; c++) wst[c] = (row && (EXACT || c < n)) ? M[c + r * n] : 0.0;  // row of WS^T
  // index of the first maximal diagonal entry of WS
  double dval = 0.0;
#pragma unroll
  for (int c = 0; c < N; c++) dval = (lane == c) ? ws[c] : dval;
  const double dmax = WaveMax(row ? dval : -1.7976931348623157e308);
  const unsigned long long hit = __ballot(row && dval == dmax);
  const int index = __builtin_amdgcn_readfirstlane(hit ? __ffsll((long long)hit) - 1 : 0);
  // PrepareStep starts from WS.col(index), GetWeightedSlackEigenvalues from minus_s.col(index)
  double rvec = 0.0;
#pragma unroll
  for (int c = 0; c < N; c++) rvec = (c == index) ? (MODE == 0 ? ws[c] : s[c]) : rvec;
  LanczosRows<N>(ws, wst, w, rvec, lane, EXACT ? ITERS : n / 2, sAB[wave], sOut[wave], n);
  // tr(WS WS) and tr(WS) with the workgroup kernel's partition: virtual thread t = 64 v + lane
  // takes elements t and t + 256, the four wave sums are added in order
  double t2 = 0.0, t1 = 0.0;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    double p2 = 0.0, p1 = 0.0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int q = 64 * v + lane + 256 * k;
      if (q < nn) {
        int a, b;
        if (EXACT) {
          a = q % N;
          b = q / N;
        } else {  // (q < 1024, n <= 32: exact in single precision after one correction)
          b = (int)(((float)q + 0.5f) * (1.0f / (float)n));
          a = q - b * n;
          if (a < 0) {
            a += n;
            b--;
          } else if (a >= n) {
            a -= n;
            b++;
          }
        }
        p2 = fma(M[q], M[b + a * n], p2);
        if (a == b) p1 += M[q];
      }
    }
    t2 += WaveSum(p2);
    t1 += WaveSum(p1);
  }
  WaveSync();
  if (lane == 0) {
    double mn = sOut[wave][0], mx = sOut[wave][1];
    if (!sa.no_clamp) ClampToSpectrumBound(n, t1, t2, &mn, &mx);
    if (MODE == 0) {
      const double l1 = fabs(sa.e_weight + mn), l2 = fabs(sa.e_weight + mx);
      const double v0 = t2 + 2 * t1 + n, v1 = l1 < l2 ? l2 : l1;
      sa.info[2 * id] = v0;
      sa.info[2 * id + 1] = v1;
      if (tail.slots) {
        AgentStore(tail.slots + 4 * id, v0);
        AgentStore(tail.slots + 4 * id + 1, v1);
      }
    } else {
      sa.info[4 * id] = -mx;      // lambda_min
      sa.info[4 * id + 1] = -mn;  // lambda_max
      sa.info[4 * id + 2] = t2;
      sa.info[4 * id + 3] = -t1;
      if (tail.slots) {
        AgentStore(tail.slots + 4 * id, -mx);
        AgentStore(tail.slots + 4 * id + 1, -mn);
        AgentStore(tail.slots + 4 * id + 2, t2);
        AgentStore(tail.slots + 4 * id + 3, -t1);
      }
    }
  }
}

// order 20 exactly, or any order from 3 below it on the same instance
inline bool LmiPrepareRowsSupports(int n, int m, int herm_d, bool sparse) {
  return n >= 3 && n <= 20 && m <= 64 && herm_d == 0 && !sparse;
}

inline bool LmiTakeStepRowsSupports(int n) { return n <= 32; }

}  // namespace cxk
