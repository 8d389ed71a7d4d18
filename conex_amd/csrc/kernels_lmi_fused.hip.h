// lmi_schur_fused<N, M>: dense-LMI Schur assembly for small orders (benchmark shape N = 20,
// M = 20).  Reference semantics: ConstructSchurComplementSystem(DenseLMIConstraint*),
// dense_lmi_constraint.cc:72-103.
//
// One workgroup per constraint; the M matrices A_i and C (matrix index M) are stacked:
// M1 = M + 1 matrices, M1*N rows, ONE LANE OWNS ONE ROW.
//
//   phase 1  stage A into LDS (coalesced 16-byte global loads, all issued before the first
//            LDS store; each wave touches only its own matrices)
//   phase 2  P_i = A_i W on the VALU: row r of P_i = a_r^T W.  W[k][j] is wave-uniform; it
//            reaches the FMA as lane j of a 16-lane row through DPP row_newbcast (the one DPP
//            mode gfx90a+ allows on fp64), so a v_fmac_f64 needs neither an SGPR nor an LDS
//            read for its W operand.  P rows go back to LDS (over A).
//   phase 3  G(i,j) = tr(W A_i W A_j) = tr(P_i P_j) = sum_{r,b} P_i[r][b] P_j[b][r] -- the second
//            product W (A_i W) of the reference is never formed.  This is a (M1 x K)(K x M1)
//            GEMM with K = N^2 and runs on the fp64 matrix pipe (v_mfma_f64_16x16x4): the K
//            range is dealt to the wavefronts, each accumulates the lower 16x16 tiles, the
//            per-wave partial tiles are summed in a fixed order through LDS.  The MFMA pipe is
//            otherwise idle in this kernel and overlaps with the VALU phase of the other
//            workgroup resident on the CU.
//   outputs  G (lower triangle), AQc(i) = tr(P_i P_C), <c,Qc> = tr(P_C P_C), AW(i) = tr(P_i),
//            <w,c> = tr(P_C).
//
// LDS layout: matrix i at i*MS, row r at r*LD, LD odd and MS = 2 (mod 4): both MFMA operand
// reads (A-op: 16 matrices x 4 consecutive elements of one row; B-op: 16 matrices x 4
// consecutive elements of one column) are then bank-conflict free for ds_read_b64.
//
// Mathematically identical to the reference's  vec(W A_i W) . vec(A_j); the summation order
// differs from Eigen's (tolerance parity <= 1e-13 rel, tests/test_gpu_parity.py).  All sums are
// in a fixed order: results are bit-reproducible run to run.
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

template <int N, int M>
struct FusedCfg {
  static_assert(N % 4 == 0 && N > 16 && N <= 32, "row = N/4 MFMA k-steps; two DPP rows cover W");
  static constexpr int M1 = M + 1;
  static constexpr int MPW = 64 / N;                 // matrices per wavefront
  static constexpr int WAVES = (M1 + MPW - 1) / MPW;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int LD = N + 1;                   // odd row stride
  static constexpr int MS = N * LD + (6 - (N * LD) % 4) % 4;  // smallest stride = 2 (mod 4)
  static constexpr int TI = (M1 + 15) / 16;          // 16-row tiles
  // 17 <= M1 <= 24: the lower triangle fits TWO 16 x 16 MFMA tiles instead of three -- tile 0 =
  // rows R..M1-1 x columns 0..15 (R = M1 - 16), tile 1 = the symmetric block over the 2R matrices
  // {0..R-1} u {16..M1-1}, which holds the pairs tile 0 misses (both < R, or both >= 16)
  static constexpr bool TWO = (M1 > 16 && M1 <= 24);
  static constexpr int R = M1 - 16;
  static constexpr int NT = TWO ? 2 : TI * (TI + 1) / 2;  // accumulated tiles
  static constexpr int KSTEPS = N * N / 4;
  static constexpr size_t kRows = (size_t)M1 * MS;   // doubles: A, then P
  static constexpr size_t kW = (size_t)N * 32;       // padded copy of W
  static constexpr size_t kDiag = (size_t)M1 * N;    // P_i[r][r]
  static_assert(MS % 4 == 2, "matrix stride must be 2 mod 4 doubles");
  static_assert((size_t)WAVES * NT * 256 <= kRows, "partial tiles reuse the row buffer");
  static constexpr size_t LDS = sizeof(double) * (kRows + kW + kDiag);
};

// acc += w_bcast * v, W operand = lane J of each 16-lane row of `w` (DPP row_newbcast).
template <int J>
__device__ __forceinline__ void FmaBcast(double& acc, double w, double v) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(w), "v"(v), "n"(J));
}

template <int N, int J0, int J1>
struct BcastRow {  // out[J0..J1) += w[lane (j - BASE) of the row] * v, unrolled at compile time
  template <int BASE>
  static __device__ __forceinline__ void run(double (&out)[N], double w, double v) {
    if constexpr (J0 < J1) {
      FmaBcast<J0 - BASE>(out[J0], w, v);
      BcastRow<N, J0 + 1, J1>::template run<BASE>(out, w, v);
    }
  }
};

// out[j] = sum_k v[k] * W[k][j].  v is read from LDS.  Row k of W comes from its LDS copy as two
// per-lane values x = W[k][lane & 15], y = W[k][16 + (lane & 15)] (zero padded) consumed through
// row_newbcast.  One row is prefetched ahead.
template <int N>
__device__ __forceinline__ void RowTimesW(const double* v, const double* sW, int lane,
                                          double (&out)[N]) {
  constexpr int LW = 32;
#pragma unroll
  for (int j = 0; j < N; j++) out[j] = 0.0;
  const int l15 = lane & 15;
  double x = sW[l15], y = sW[16 + l15], vk = v[0];
#pragma unroll 2
  for (int k = 0; k < N; k++) {
    const int kn = (k + 1 < N) ? k + 1 : k;
    const double xn = sW[kn * LW + l15], yn = sW[kn * LW + 16 + l15], vn = v[kn];
    BcastRow<N, 0, 16>::template run<0>(out, x, vk);
    BcastRow<N, 16, N>::template run<16>(out, y, vk);
    x = xn;
    y = yn;
    vk = vn;
  }
}

#ifdef CXK_DEBUG_STAMPS
__device__ long long g_fused_stamp[8 * 8];
#define FSTAMP(i)                                                                 \
  do {                                                                            \
    if (blockIdx.x == 700 && (threadIdx.x & 63) == 0)                             \
      g_fused_stamp[(threadIdx.x >> 6) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FSTAMP(i) \
  do {            \
  } while (0)
#endif

typedef double d4_t __attribute__((ext_vector_type(4)));

template <int N, int M>
__global__ void __launch_bounds__((FusedCfg<N, M>::THREADS)) lmi_schur_fused(LmiGroup g, Arena ar) {
  using Cfg = FusedCfg<N, M>;
  constexpr int M1 = Cfg::M1, MPW = Cfg::MPW, LD = Cfg::LD, MS = Cfg::MS, TI = Cfg::TI,
                NT = Cfg::NT, WAVES = Cfg::WAVES;
  constexpr int NN = N * N;
  extern __shared__ double buf[];
  double* sW = buf + Cfg::kRows;      // N rows of 32 doubles (columns >= N are zero)
  double* sDiag = sW + Cfg::kW;       // M1 * N
  const int mem = blockIdx.x;
  const int id = g.ids[mem];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gsub = lane / N;
  const int r = lane - gsub * N;
  const int i = wave * MPW + gsub;
  const bool active = (gsub < MPW) && (i < M1);
  const double* A = g.A + (size_t)mem * g.a_stride;
  const double* Cm = g.C + (size_t)mem * NN;
  const double* Wg = g.W + (size_t)mem * NN;
  FSTAMP(0);
  for (int e = threadIdx.x; e < N * 32; e += blockDim.x) {
    const int k = e >> 5, j = e & 31;
    sW[e] = (j < N) ? Wg[k * N + j] : 0.0;
  }
  // phase 1: column c of A_i (= its row c, A_i symmetric) -> LDS row (i, c)
  {
    const int i0 = wave * MPW;
    const int cnt = (M1 - i0) < MPW ? (M1 - i0) : MPW;
    constexpr int CHK = NN / 2;                        // 16-byte chunks per matrix
    constexpr int UN = (MPW * CHK + 63) / 64;          // chunks per lane
    double2 v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int c = u * 64 + lane;
      const int q = c / CHK, e = c - q * CHK;
      const int mi = i0 + q;
      const double2* src = reinterpret_cast<const double2*>(mi < M ? A + (size_t)mi * NN : Cm);
      v[u] = (q < cnt) ? src[e] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int c = u * 64 + lane;
      const int q = c / CHK, e = c - q * CHK;
      const int mi = i0 + q;
      const int col = (2 * e) / N, row = (2 * e) % N;
      if (q < cnt) {
        double* dst = &buf[mi * MS + col * LD + row];
        dst[0] = v[u].x;
        dst[1] = v[u].y;
      }
    }
  }
  FSTAMP(1);
  __syncthreads();  // W copy visible to every wave (the A rows are wave-private)
  FSTAMP(2);
  // phase 2: P row in registers, then over the A row in LDS
  const int myrow = active ? (i * MS + r * LD) : 0;
  double t[N];
  RowTimesW<N>(&buf[myrow], sW, lane, t);
  FSTAMP(3);
  WaveSync();
  if (active) {
#pragma unroll
    for (int k = 0; k < N; k++) buf[myrow + k] = t[k];
    sDiag[i * N + r] = buf[myrow + r];  // own store: program order suffices
  }
  FSTAMP(4);
  __syncthreads();  // every P row is in LDS
  FSTAMP(5);
  // phase 3: tr(P_i P_j) on the matrix pipe.  k-step ks covers P_i[rr][b0..b0+3] x P_j[b0..b0+3][rr].
  d4_t acc[NT];
#pragma unroll
  for (int tt = 0; tt < NT; tt++) acc[tt] = (d4_t){0.0, 0.0, 0.0, 0.0};
  if constexpr (Cfg::TWO) {
    constexpr int R = Cfg::R;
    const int il = lane & 15, kq = lane >> 4;
    const int rowA = (R + il) * MS;                                      // tile 0 rows: matrices R .. M1-1
    const int colA = il * MS;                                            // tile 0 columns: matrices 0 .. 15
    const int setB = (il < R ? il : (il < 2 * R ? 16 + il - R : M1 - 1)) * MS;  // tile 1 rows = columns
#pragma unroll 2
    for (int ks = wave; ks < Cfg::KSTEPS; ks += WAVES) {
      const int rr = ks / (N / 4), b0 = 4 * (ks % (N / 4));
      const int ao = rr * LD + b0 + kq, bo = (b0 + kq) * LD + rr;
      const double a0 = buf[rowA + ao], b0v = buf[colA + bo];
      const double a1 = buf[setB + ao], b1v = buf[setB + bo];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0v, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1v, acc[1], 0, 0, 0);
    }
  } else {
    const int il = lane & 15, kq = lane >> 4;
    int mrow[TI];
#pragma unroll
    for (int I = 0; I < TI; I++) {
      const int mi = 16 * I + il;
      mrow[I] = (mi < M1 ? mi : M1 - 1) * MS;  // padding rows of the last tile alias matrix M1-1
    }
#pragma unroll 2
    for (int ks = wave; ks < Cfg::KSTEPS; ks += WAVES) {
      const int rr = ks / (N / 4), b0 = 4 * (ks % (N / 4));
      double aop[TI], bop[TI];
#pragma unroll
      for (int I = 0; I < TI; I++) {
        aop[I] = buf[mrow[I] + rr * LD + b0 + kq];
        bop[I] = buf[mrow[I] + (b0 + kq) * LD + rr];
      }
      int tt = 0;
#pragma unroll
      for (int I = 0; I < TI; I++)
#pragma unroll
        for (int J = 0; J <= I; J++) {
          acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[I], bop[J], acc[tt], 0, 0, 0);
          tt++;
        }
    }
  }
  __syncthreads();  // all waves are done reading P: the row buffer becomes the partial-tile store
  {
    const int il = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < NT; tt++)
#pragma unroll
      for (int e = 0; e < 4; e++)  // C/D layout: col = lane & 15, row = (lane >> 4) + 4 e
        buf[(wave * NT + tt) * 256 + (kq + 4 * e) * 16 + il] = acc[tt][e];
  }
  __syncthreads();
  FSTAMP(6);
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  const double osc = g.herm_d > 1 ? 1.0 / g.herm_d : 1.0;  // Hermitian cones: tr over the real representation
  for (int idx = threadIdx.x; idx < M1 * M1; idx += blockDim.x) {
    const int ii = idx / M1, jj = idx - ii * M1;
    if (jj > ii) continue;
    int off;
    if constexpr (Cfg::TWO) {
      constexpr int R = Cfg::R;
      if (ii >= R && jj < 16)
        off = (ii - R) * 16 + jj;                             // tile 0
      else if (ii < R)
        off = 256 + ii * 16 + jj;                             // tile 1, both among the first R
      else
        off = 256 + (R + ii - 16) * 16 + (R + jj - 16);       // tile 1, both >= 16
    } else {
      const int I = ii >> 4, J = jj >> 4;
      const int tt = I * (I + 1) / 2 + J;
      off = tt * 256 + (ii & 15) * 16 + (jj & 15);
    }
    double s = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) s += buf[w * NT * 256 + off];
    s *= osc;
    if (ii < M)
      G[ii + (size_t)jj * M] = s;
    else if (jj < M)
      AQc[jj] = s;
    else
      ar.sc[2 * id + 1] = s;
  }
  for (int ii = threadIdx.x; ii < M1; ii += blockDim.x) {
    double s = 0;
#pragma unroll
    for (int q = 0; q < N; q++) s += sDiag[ii * N + q];
    s *= osc;
    if (ii < M)
      AW[ii] = s;
    else
      ar.sc[2 * id] = s;
  }
}

// Shapes with a compiled instance: the benchmark shape (20, 20) and (24, 24) -- BASELINE config 5's
// complex Hermitian cones of order 12 over 24 variables in their real representation.
inline bool LmiFusedSupports(int n, int m) { return (n == 20 && m == 20) || (n == 24 && m == 24); }

template <int N, int M>
inline hipError_t LaunchLmiSchurFusedT(const LmiGroup& g, const Arena& ar, hipStream_t stream) {
  using Cfg = FusedCfg<N, M>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lmi_schur_fused<N, M>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
    if (e != hipSuccess) return e;
    configured = true;
    if (getenv("CXK_DEBUG_LEVELS")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&lmi_schur_fused<N, M>),
                                                         Cfg::THREADS, Cfg::LDS);
      fprintf(stderr, "lmi_schur_fused<%d,%d>: %d threads, %zu B LDS, %d workgroups per CU\n", N, M, Cfg::THREADS,
              (size_t)Cfg::LDS, nb);
    }
  }
  lmi_schur_fused<N, M><<<g.count, Cfg::THREADS, Cfg::LDS, stream>>>(g, ar);
  return hipGetLastError();
}

inline hipError_t LaunchLmiSchurFused(const LmiGroup& g, const Arena& ar, hipStream_t stream) {
  if (g.n == 20 && g.m == 20) return LaunchLmiSchurFusedT<20, 20>(g, ar, stream);
  if (g.n == 24 && g.m == 24) return LaunchLmiSchurFusedT<24, 24>(g, ar, stream);
  return hipErrorNotSupported;
}

}  // namespace cxk
