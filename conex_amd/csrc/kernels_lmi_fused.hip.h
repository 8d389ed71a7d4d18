// lmi_schur_fused<N, M>: register / scalar-operand formulation of the dense-LMI Schur assembly
// (reference: ConstructSchurComplementSystem(DenseLMIConstraint*), dense_lmi_constraint.cc:72-103)
// for small orders -- the benchmark shape is N = 20, M = 20.
//
// One workgroup per constraint.  The M matrices A_i and C are stacked (M1 = M + 1 matrices,
// M1*N rows); ONE LANE OWNS ONE ROW:  lane (i, r) holds row r of A_i in registers.
//   P_i = A_i W      : row r of P_i = a_r^T W           -- W[k][j] is wave-uniform, fetched with
//   X_i = W A_i W    : row r of X_i = (col r of P_i)^T W    scalar loads, so each v_fma_f64 takes
//                                                          one VGPR and one SGPR operand (no LDS)
//   G(i,j) = <A_i, X_j>, AQc(i) = <A_i, X_C>, <c,Qc> = <C, X_C>, AW(i) = tr(P_i), <w,c> = tr(P_C)
// One LDS buffer of M1*N padded rows is used three times: (1) staging of A (coalesced global
// reads, each wave touches only its own matrices -> no workgroup barrier, loads of one wave
// overlap FMAs of the others), (2) transposition of P (row -> column ownership), (3) the X rows
// every lane then contracts against its own A row.  The pair (i, j) is computed by the lane
// group of i for j = i, i-1, ..., i-M1/2 (mod M1): a circulant assignment that gives every lane
// the same trip count.  Partial sums are reduced over the N rows through a small LDS transpose
// in a fixed order, so results are bit-reproducible.
//
// Symmetry used: A_i, W symmetric => row r of A_i is its column r (contiguous in memory) and
// X_i = P_i^T W.  Mathematically G(i,j) = tr(W A_i W A_j) as in the reference; the summation
// order differs from Eigen's (tolerance parity, tests/test_gpu_parity.py).
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

template <int N, int M>
struct FusedCfg {
  static constexpr int M1 = M + 1;
  static constexpr int MPW = 64 / N;                      // matrices per wavefront
  static constexpr int WAVES = (M1 + MPW - 1) / MPW;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int LD = N + 2;                        // padded row (16-byte aligned, N even)
  static constexpr int D = M1 / 2 + 1;                    // partners i, i-1, ..., i-M1/2
  static constexpr size_t LDS = sizeof(double) * (size_t)M1 * N * LD;
};

typedef const double __attribute__((address_space(4))) * UniformPtr;

// out[j] = sum_k v[k] * W[k][j].  v is read from LDS (stride vs), W is wave-uniform and comes
// in through the scalar cache (s_load), so each v_fma_f64 has one VGPR and one SGPR operand.
// The k loop is deliberately NOT fully unrolled: a full unroll makes the scheduler hoist all
// N*N scalar loads and spill hundreds of SGPRs.
template <int N>
__device__ __forceinline__ void RowTimesW(const double* v, int vs, UniformPtr W, double (&out)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) out[j] = 0.0;
#pragma unroll 2
  for (int k = 0; k < N; k++) {
    const double vk = v[k * vs];
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = fma(vk, W[k * N + j], out[j]);
  }
}

template <int N, int M>
__global__ void __launch_bounds__((FusedCfg<N, M>::THREADS)) lmi_schur_fused(LmiGroup g, Arena ar) {
  using Cfg = FusedCfg<N, M>;
  constexpr int M1 = Cfg::M1, MPW = Cfg::MPW, LD = Cfg::LD, D = Cfg::D;
  constexpr int NN = N * N;
  extern __shared__ double buf[];
  const int mem = blockIdx.x;
  const int id = g.ids[mem];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gsub = lane / N;
  const int r = lane - gsub * N;
  const int i = wave * MPW + gsub;
  const bool active = (gsub < MPW) && (i < M1);
  const double* A = g.A + (size_t)mem * M * NN;
  const double* Cm = g.C + (size_t)mem * NN;
  UniformPtr W = (UniformPtr)(uintptr_t)(g.W + (size_t)mem * NN);

  // (1) stage this wave's matrices: column c of A_i (= row c) -> padded row (i, c)
  {
    const int i0 = wave * MPW;
    const int cnt = (M1 - i0) < MPW ? (M1 - i0) : MPW;
    for (int q = 0; q < cnt; q++) {
      const int mi = i0 + q;
      const double2* src = reinterpret_cast<const double2*>(mi < M ? A + (size_t)mi * NN : Cm);
      for (int e = lane; e < NN / 2; e += 64) {
        const double2 v = src[e];
        const int col = (2 * e) / N, row = (2 * e) % N;
        *reinterpret_cast<double2*>(&buf[(mi * N + col) * LD + row]) = v;
      }
    }
  }
  WaveSync();
  double a[N];
  const int myrow = (active ? (i * N + r) : 0) * LD;
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    const double2 v = *reinterpret_cast<const double2*>(&buf[myrow + k]);
    a[k] = v.x;
    a[k + 1] = v.y;
  }
  // (2) P row, transpose through LDS
  double t[N];
  RowTimesW<N>(&buf[myrow], 1, W, t);
  WaveSync();
  if (active) {
#pragma unroll
    for (int k = 0; k < N; k += 2)
      *reinterpret_cast<double2*>(&buf[myrow + k]) = make_double2(t[k], t[k + 1]);
  }
  WaveSync();
  const int mybase = (active ? i * N : 0) * LD;
  const double diagP = buf[myrow + r];
  // (3) X row
  RowTimesW<N>(&buf[mybase + r], LD, W, t);
  WaveSync();
  if (active) {
#pragma unroll
    for (int k = 0; k < N; k += 2)
      *reinterpret_cast<double2*>(&buf[myrow + k]) = make_double2(t[k], t[k + 1]);
  }
  __syncthreads();
  // (4) contractions <A_i[r,:], X_j[r,:]> for the D circulant partners; the N row-partials of a
  // matrix are folded with ds_bpermute shuffles in a fixed tree order (bit-reproducible), lane
  // r == 0 of the group writes the result.  The d loop stays a real loop: unrolling it lets the
  // scheduler hoist D*N/2 LDS reads and spill.
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  constexpr int P2 = (N > 32) ? 32 : (N > 16) ? 16 : (N > 8) ? 8 : (N > 4) ? 4 : (N > 2) ? 2 : 1;
  auto group_sum = [&](double v) {
    double o = __shfl_down(v, P2, 64);
    if (r < N - P2) v += o;
#pragma unroll
    for (int off = P2 / 2; off > 0; off >>= 1) {
      o = __shfl_down(v, off, 64);
      if (r < off) v += o;
    }
    return v;
  };
  {
    const double tr = group_sum(active ? diagP : 0.0);
    if (active && r == 0) {
      if (i < M)
        AW[i] = tr;
      else
        ar.sc[2 * id] = tr;
    }
  }
#pragma unroll 1
  for (int d = 0; d < D; d++) {
    int j = i - d;
    if (j < 0) j += M1;
    const int jrow = (active ? (j * N + r) : 0) * LD;
    double s = 0;
#pragma unroll
    for (int k = 0; k < N; k += 2) {
      const double2 x = *reinterpret_cast<const double2*>(&buf[jrow + k]);
      s = fma(a[k], x.x, s);
      s = fma(a[k + 1], x.y, s);
    }
    s = group_sum(active ? s : 0.0);
    if (active && r == 0) {
      if ((M1 % 2 == 0) && d == M1 / 2 && i < j) continue;  // pair owned by the other side
      if (i < M && j < M) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        G[hi + (size_t)lo * M] = s;
      } else if (i == M && j == M) {
        ar.sc[2 * id + 1] = s;
      } else {
        AQc[i < j ? i : j] = s;
      }
    }
  }
}

inline bool LmiFusedSupports(int n, int m) { return n == 20 && m == 20; }

inline hipError_t LaunchLmiSchurFused(const LmiGroup& g, const Arena& ar, hipStream_t stream) {
  if (g.n == 20 && g.m == 20) {
    using Cfg = FusedCfg<20, 20>;
    static bool configured = false;
    if (!configured) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lmi_schur_fused<20, 20>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
      if (e != hipSuccess) return e;
      configured = true;
    }
    lmi_schur_fused<20, 20><<<g.count, Cfg::THREADS, Cfg::LDS, stream>>>(g, ar);
    return hipGetLastError();
  }
  return hipErrorNotSupported;
}

}  // namespace cxk
