// lmi_schur_fused<N, M>: register / scalar-operand formulation of the dense-LMI Schur assembly
// (reference: ConstructSchurComplementSystem(DenseLMIConstraint*), dense_lmi_constraint.cc:72-103)
// for small orders -- the benchmark shape is N = 20, M = 20.
//
// One workgroup per constraint.  The M matrices A_i and C are stacked (M1 = M + 1 matrices,
// M1*N rows); ONE LANE OWNS ONE ROW:  lane (i, r) holds row r of A_i in registers.
//   P_i = A_i W : row r of P_i = a_r^T W.  W[k][j] is wave-uniform; it reaches the FMA as lane j of
//                 a 16-lane row through DPP row_newbcast, so a v_fmac_f64 needs no SGPR and no
//                 LDS read for its W operand.
//   G(i,j) = tr(W A_i W A_j) = tr(P_i P_j) = sum_{r,b} P_i[r][b] P_j[b][r]  -- the second product
//                 W (A_i W) of the reference is never formed: only P is needed.
//   AQc(i) = tr(P_i P_C), <c,Qc> = tr(P_C P_C), AW(i) = tr(P_i), <w,c> = tr(P_C)   (C = matrix M)
// One LDS buffer of M1*N padded rows is used twice: (1) staging of A (coalesced global reads;
// each wave touches only its own matrices), (2) the P rows; lane (i, r) then contracts its own
// row r of P_i (still in registers) against COLUMN r of P_j read from LDS.  The pair (i, j) is computed by the lane
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
  static constexpr size_t kRows = (size_t)M1 * N * LD;           // doubles: A / P / X rows
  static constexpr size_t LDS = sizeof(double) * (kRows + (size_t)N * 32);  // + padded copy of W
};

// acc += w_bcast * v : the W operand is lane `J` of each 16-lane row of `w` (DPP row_newbcast,
// the one DPP mode gfx90a+ allows on fp64 VALU ops), so the wave-uniform W entry costs neither
// an SGPR nor an LDS read per FMA.
template <int J>
__device__ __forceinline__ void FmaBcast(double& acc, double w, double v) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(w), "v"(v), "n"(J));
}

template <int N, int J0, int J1>
struct BcastRow {  // out[J0..J1) += w[lane j - base] * v, unrolled at compile time
  template <int BASE>
  static __device__ __forceinline__ void run(double (&out)[N], double w, double v) {
    if constexpr (J0 < J1) {
      FmaBcast<J0 - BASE>(out[J0], w, v);
      BcastRow<N, J0 + 1, J1>::template run<BASE>(out, w, v);
    }
  }
};

// out[j] = sum_k v[k] * W[k][j].  v is read from LDS (stride vs).  Row k of W is read from its
// LDS copy as two per-lane values: x = W[k][lane & 15] and y = W[k][16 + (lane & 15)] (zero
// padded), which the FMAs then consume through row_newbcast.  One row is prefetched ahead.
template <int N>
__device__ __forceinline__ void RowTimesW(const double* v, int vs, const double* sW, int lane,
                                          double (&out)[N]) {
  static_assert(N > 16 && N <= 32, "two 16-lane broadcast registers cover 17..32 columns");
  constexpr int LW = 32;  // padded W row in LDS
#pragma unroll
  for (int j = 0; j < N; j++) out[j] = 0.0;
  const int l15 = lane & 15;
  double x = sW[l15], y = sW[16 + l15], vk = v[0];
#pragma unroll 2
  for (int k = 0; k < N; k++) {
    const int kn = (k + 1 < N) ? k + 1 : k;
    const double xn = sW[kn * LW + l15], yn = sW[kn * LW + 16 + l15], vn = v[kn * vs];
    BcastRow<N, 0, 16>::template run<0>(out, x, vk);
    BcastRow<N, 16, N>::template run<16>(out, y, vk);
    x = xn;
    y = yn;
    vk = vn;
  }
}

#ifdef CXK_DEBUG_STAMPS
__device__ long long g_fused_stamp[8 * 8];
#define FSTAMP(i) do { if (blockIdx.x == 700 && (threadIdx.x & 63) == 0) g_fused_stamp[(threadIdx.x >> 6) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

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
  FSTAMP(0);
  const double* Wg = g.W + (size_t)mem * NN;
  double* sW = buf + Cfg::kRows;  // N rows of 32 doubles (columns >= N are zero)
  for (int e = threadIdx.x; e < N * 32; e += blockDim.x) {
    const int k = e >> 5, j = e & 31;
    sW[e] = (j < N) ? Wg[k * N + j] : 0.0;
  }

  // (1) stage this wave's matrices: column c of A_i (= row c) -> padded row (i, c).  All loads of
  // the wave are issued before the first LDS store (independent registers), otherwise every
  // 16-byte load waits for the previous one's round trip to HBM.
  {
    const int i0 = wave * MPW;
    const int cnt = (M1 - i0) < MPW ? (M1 - i0) : MPW;
    constexpr int CH = NN / 2;                        // 16-byte chunks per matrix
    constexpr int UN = (MPW * CH + 63) / 64;          // chunks per lane
    double2 v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int c = u * 64 + lane;
      const int q = c / CH, e = c - q * CH;
      const int mi = i0 + q;
      const double2* src = reinterpret_cast<const double2*>(mi < M ? A + (size_t)mi * NN : Cm);
      v[u] = (q < cnt) ? src[e] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int c = u * 64 + lane;
      const int q = c / CH, e = c - q * CH;
      const int mi = i0 + q;
      const int col = (2 * e) / N, row = (2 * e) % N;
      if (q < cnt) *reinterpret_cast<double2*>(&buf[(mi * N + col) * LD + row]) = v[u];
    }
  }
  FSTAMP(1);
  __syncthreads();  // W copy visible to every wave (A staging above is wave-private)
  FSTAMP(2);
  const int myrow = (active ? (i * N + r) : 0) * LD;
  // (2) P row (kept in registers) and its LDS copy
  double t[N];
  RowTimesW<N>(&buf[myrow], 1, sW, lane, t);
  FSTAMP(3);
  WaveSync();
  if (active) {
#pragma unroll
    for (int k = 0; k < N; k += 2)
      *reinterpret_cast<double2*>(&buf[myrow + k]) = make_double2(t[k], t[k + 1]);
  }
  const double diagP = active ? buf[myrow + r] : 0.0;  // own store: no barrier needed
  FSTAMP(4);
  __syncthreads();  // every P row is in LDS
  FSTAMP(5);
  // (4) contractions sum_b P_i[r][b] P_j[b][r] for the D circulant partners (+ the trace of P as
  // sum number D).  Sums are processed in chunks of CH: the CH row-partials of a lane are
  // folded over the N rows of its matrix with ds_bpermute shuffles whose dependent steps are
  // interleaved across the chunk (latency amortised CH-fold); the chunk loop stays rolled so
  // the scheduler cannot hoist all D*N/2 LDS reads (that spills).
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  constexpr int P2 = (N > 32) ? 32 : (N > 16) ? 16 : (N > 8) ? 8 : (N > 4) ? 4 : (N > 2) ? 2 : 1;
  constexpr int CH = 2;
  constexpr int NSUM = D + 1;
#pragma unroll 1
  for (int c0 = 0; c0 < NSUM; c0 += CH) {
    double p[CH];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      const int d = c0 + u;
      double s = 0;
      if (d < D) {
        int j = i - d;
        if (j < 0) j += M1;
        const int jcol = (active ? j * N : 0) * LD + r;  // column r of P_j: lanes r consecutive
#pragma unroll
        for (int k = 0; k < N; k++) s = fma(t[k], buf[jcol + k * LD], s);
      } else if (d == D) {
        s = diagP;
      }
      p[u] = active ? s : 0.0;
    }
    {  // fold rows r + P2 .. N-1 onto 0 .. N-P2-1, then a power-of-two tree
      double o[CH];
#pragma unroll
      for (int u = 0; u < CH; u++) o[u] = __shfl_down(p[u], P2, 64);
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (r < N - P2) p[u] += o[u];
#pragma unroll
      for (int off = P2 / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int u = 0; u < CH; u++) o[u] = __shfl_down(p[u], off, 64);
#pragma unroll
        for (int u = 0; u < CH; u++)
          if (r < off) p[u] += o[u];
      }
    }
    if (active && r == 0) {
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int d = c0 + u;
        const double s = p[u];
        if (d == D) {
          if (i < M)
            AW[i] = s;
          else
            ar.sc[2 * id] = s;
        } else if (d < D) {
          int j = i - d;
          if (j < 0) j += M1;
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
  }
  FSTAMP(6);
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
