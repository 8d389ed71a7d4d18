// Batched fp64 GEMM on the matrix pipe: kernels and launchers (interface: kernels_gemm.hip.h).
//
// Two kernels share the 64 x 64 workgroup tile (four wavefronts, 2 x 2 MFMA tiles each):
//
//  gemm_f64_dma   the fast path.  Measured on gfx950 (profiles/r02/mfma_f64_peak.jsonl): while an
//                 fp64 MFMA executes, NO vector-ALU instruction of that SIMD issues, and an LDS
//                 store costs ~10 cycles of the same pipe -- address arithmetic, predicates and
//                 register-staged ds_writes come straight out of the MFMA rate.  So the operand
//                 tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR, no VALU,
//                 no ds_write), 32 k-values per stage, two stages in LDS, ONE raw barrier per stage
//                 with the next stage's DMA in flight across it (counted s_waitcnt, never a fence).
//                 A DMA instruction writes 1 KiB of LDS contiguously in lane order but takes its
//                 SOURCE address per lane: the images are made bank-conflict free for the MFMA
//                 operand reads by rotating which 16-byte chunk a lane fetches (odd k-rows by 8
//                 chunks in the m-major image, row m by m mod 16 chunks in the k-major image), and
//                 k-rows past K are fetched from a page of zeros.  The main loop has no vector-ALU
//                 instruction besides the MFMAs.  Needs even leading dimensions and 16-byte
//                 aligned operands (a chunk is two doubles).
//  gemm_f64_mfma  the general kernel (any alignment / odd sizes): register-staged, predicated loads.
#include "kernels_gemm.hip.h"

#include <algorithm>
#include <type_traits>

#include "device_utils.h"

namespace cxk {

constexpr int kGemmBM = 64, kGemmBN = 64;
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


// ------------------------------------------------------------------------------------------
// LDS-DMA kernel
// ------------------------------------------------------------------------------------------
// k-values per stage: BK = 32 (two stages of two 16 KB tiles = 64 KB of LDS, two workgroups per
// CU) for long K, BK = 16 (32 KB, four workgroups per CU: more tiles in flight to hide a short
// tile's start-up and epilogue) otherwise.  The epilogue stages 64 x 65 doubles in the same array.
constexpr int kDmaLdsDoubles = 4 * 64 * 32;
static_assert(kDmaLdsDoubles >= 64 * 65, "result staging fits the operand stages");
__device__ double g_gemm_zero_page[128];         // k-rows past K are fetched from here (zeros)

// LDS traffic of this wave has landed / is visible, then the workgroup barrier; no vmcnt wait
// (the next stage's DMA stays in flight across it).
__device__ __forceinline__ void GemmBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Which 16-byte chunk lane `c` of DMA instruction `i` (0..15) of an operand tile fetches, and where
// an element sits in the image.
//   m-major image (source contiguous along the tile's 64 rows r): instruction i = k-rows 2i, 2i+1;
//     chunk position p < 32: (k = 2i, rows 2p, 2p+1); p >= 32: (k = 2i+1, rows 2q, 2q+1), q = (p - 8) & 31
//     element (r, k) at  (k >> 1) * 128 + ((k & 1) ? 64 + ((r + 16) & 63) : r)
//   k-major image (source contiguous along k): instruction i = rows 4i .. 4i+3;
//     chunk position p: row r = 4i + (p >> 4), k-pair kp = ((p & 15) - r) & 15
//     element (r, k) at  (r >> 2) * 128 + (r & 3) * 32 + 2 * (((k >> 1) + r) & 15) + (k & 1)
//   (KP = BK / 2 chunks per k-major row, RPI = 64 / KP rows per instruction)
template <bool KMAJOR, int BK>
__device__ __forceinline__ int ImageOffset(int r, int k) {
  constexpr int KP = BK / 2, RPI = 64 / KP;
  if (KMAJOR) return (r / RPI) * 128 + (r % RPI) * BK + 2 * (((k >> 1) + r) % KP) + (k & 1);
  return (k >> 1) * 128 + ((k & 1) ? 64 + ((r + 16) & 63) : r);
}

// Per-lane source geometry of the four DMA instructions a wave issues for one operand per stage.
struct DmaLane {
  unsigned off[4];   // byte offset of the lane's chunk from the operand's tile origin at stage 0 (k = k_base)
  int krel[4];       // m-major: the chunk's k relative to the stage start (to test against K); k-major: first k of the pair
};

// rows = valid rows of the operand in this tile (clamped fetch beyond), ld in doubles
template <bool KMAJOR, int BK>
__device__ __forceinline__ void MakeDmaLane(DmaLane& d, int wave, int lane, int rows, int64_t ld) {
  constexpr int KP = BK / 2, RPI = 64 / KP, NI = BK / 8;  // NI DMA instructions per wave, operand and stage
#pragma unroll
  for (int u = 0; u < NI; u++) {
    const int i = wave + 4 * u;
    int r, k;
    if (KMAJOR) {
      r = RPI * i + lane / KP;
      k = 2 * ((lane % KP - r) & (KP - 1));
    } else {
      k = 2 * i + (lane >> 5);
      const int p = lane & 31;
      r = 2 * ((lane >> 5) ? ((p - 8) & 31) : p);
    }
    int rc = r < rows ? r : (rows - 1) & ~(KMAJOR ? 0 : 1);   // rows past the edge re-read a valid row (results unused)
    if (rc < 0) rc = 0;
    d.krel[u] = k;
    d.off[u] = (unsigned)((KMAJOR ? (int64_t)rc * ld + k : (int64_t)k * ld + rc) * 8);
  }
}

template <bool TA, bool TB, int BK>
__global__ void __launch_bounds__(256, BK == 16 ? 4 : 2) gemm_f64_dma(GemmArgs g) {  // four / two workgroups per CU (LDS)
  constexpr int kDmaBK = BK, kDmaTile = 64 * BK, NI = BK / 8;
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (g.M + kGemmBM - 1) / kGemmBM;
  // Workgroups are dealt round-robin over the 8 XCDs (observed; speed only): relabel them so that
  // the tiles, splits and neighbouring matrices of a batch -- which read the same operand panels
  // -- run on one XCD and meet in its L2 (bijective for any grid size).
  int tile, split, bz;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    tile = id % gx;
    split = (id / gx) % gy;
    bz = id / (gx * gy);
  }
  int tm, tn;
  if (g.lower_only && gridDim.x != (unsigned)(tiles_m * ((g.N + kGemmBN - 1) / kGemmBN))) {
    // compact grid of a square lower-only call: tile columns one after the other, column tn holds
    // the tile rows tn .. tiles_m - 1
    int rem = tile, len = tiles_m;
    tn = 0;
    while (rem >= len) {
      rem -= len;
      len--;
      tn++;
    }
    tm = tn + rem;
  } else {
    tm = tile % tiles_m;
    tn = tile / tiles_m;
  }
  const int m_base = tm * kGemmBM, n_base = tn * kGemmBN;
  if (g.lower_only && m_base + kGemmBM - 1 < n_base) return;  // uniform per workgroup
  const int b1 = bz / g.inner, b2 = bz % g.inner;
  const double* A = g.A + b1 * g.sA1 + b2 * g.sA2;
  const double* B = g.B + b1 * g.sB1 + b2 * g.sB2;
  // K range of this split, in whole kGemmBK steps (as the general kernel deals them)
  const int ksteps = (g.K + kGemmBK - 1) / kGemmBK;
  const int per = (ksteps + g.splits - 1) / g.splits;
  const int k_lo = min(g.K, split * per * kGemmBK), k_hi = min(g.K, k_lo + per * kGemmBK);
  const int nstage = (k_hi - k_lo + kDmaBK - 1) / kDmaBK;

  // A tile: rows = m; stored m-major when !TA (A(m,k) at m + k lda), k-major when TA (k + m lda)
  // B tile: rows = n; stored m-major when TB (B(k,n) at n + k ldb), k-major when !TB (k + n ldb)
  constexpr bool AK = TA, BKM = !TB;
  DmaLane da, db;
  MakeDmaLane<AK, BK>(da, wave, lane, g.M - m_base, g.lda);
  MakeDmaLane<BKM, BK>(db, wave, lane, g.N - n_base, g.ldb);
  const char* a0 = reinterpret_cast<const char*>(A + (AK ? (int64_t)m_base * g.lda + k_lo : (int64_t)k_lo * g.lda + m_base));
  const char* b0 = reinterpret_cast<const char*>(B + (BKM ? (int64_t)n_base * g.ldb + k_lo : (int64_t)k_lo * g.ldb + n_base));
  const int64_t a_step = (AK ? (int64_t)kDmaBK : (int64_t)kDmaBK * g.lda) * 8;
  const int64_t b_step = (BKM ? (int64_t)kDmaBK : (int64_t)kDmaBK * g.ldb) * 8;
  const char* zero = reinterpret_cast<const char*>(g_gemm_zero_page) + 16 * lane;

  auto issue = [&](int s) {
    double* sa = lds + (s & 1) * 2 * kDmaTile;
    double* sb = sa + kDmaTile;
    const int krem = k_hi - k_lo - s * kDmaBK;  // k-values of this stage that exist
    const char* as = a0 + (int64_t)s * a_step;
    const char* bs = b0 + (int64_t)s * b_step;
#pragma unroll
    for (int u = 0; u < NI; u++) {
      const int i = wave + 4 * u;
      const char* src = da.krel[u] < krem ? as + da.off[u] : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                       (__attribute__((address_space(3))) void*)(sa + i * 128), 16, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NI; u++) {
      const int i = wave + 4 * u;
      const char* src = db.krel[u] < krem ? bs + db.off[u] : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                       (__attribute__((address_space(3))) void*)(sb + i * 128), 16, 0, 0);
    }
  };

  gemm_d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = gemm_d4{0.0, 0.0, 0.0, 0.0};
  // A product of at most 32 x 32 (the Gram matrices of the LMI assembly: m + 1 <= 32 columns over
  // K = n^2) has work for one wave's quadrant only: there the four waves share that quadrant and
  // deal the k sub-steps of every stage among themselves; their partial tiles are summed in wave
  // order in the epilogue.
  // (compiled into the Gram form X^T Y only: the other forms' registers are spoken for)
  const bool shared_quadrant = TA && !TB && g.M <= 32 && g.N <= 32 && !g.Ct;
  const int wm = shared_quadrant ? 0 : (wave & 1) * 32, wn = shared_quadrant ? 0 : (wave >> 1) * 32;
  const int l15 = lane & 15, kq = lane >> 4;
  // operand read offsets of this lane at k-sub-step 0 (the k-step offsets are compile-time constants)
  int ao[2], bo[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    ao[i] = ImageOffset<AK, BK>(wm + 16 * i + l15, kq);
    bo[i] = ImageOffset<BKM, BK>(wn + 16 * i + l15, kq);
  }

  // The old C values of an accumulating call (beta != 0) are fetched before the K loop, in the
  // layout of the coalesced epilogue below: their latency rides on the first operand stage's.
  double* C = g.C + b1 * g.sC1 + b2 * g.sC2 + (g.splits > 1 ? split * g.sCs : 0);
  const bool partial = g.splits > 1;
  const bool accumulate = !partial && g.beta != 0.0;
  // (the trailing-update form C -= L L^T only: the 32 registers cost the other forms a spill)
  constexpr bool kEarlyC = !TA && TB;
  double c0[16];
  auto load_c = [&]() {
    const int m = m_base + (tid & 63);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int n = n_base + (tid >> 6) + 4 * u;
      const bool on = m < g.M && n < g.N && (!g.lower_only || m >= n);
      c0[u] = (on && accumulate) ? C[m + (int64_t)n * g.ldc] : 0.0;
    }
  };
  if constexpr (kEarlyC) load_c();
  // 16 x 16 sub-tiles of this wave that hold an entry to be written (inside M x N and, for a
  // lower-only call, on or below the diagonal); the others' MFMAs are skipped (wave-uniform).
  bool act[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int m_lo = m_base + wm + 16 * i, n_lo = n_base + wn + 16 * j;
      act[i][j] = m_lo < g.M && n_lo < g.N && (!g.lower_only || m_lo + 15 >= n_lo);
    }
  const bool any_act = act[0][0] || act[0][1] || act[1][0] || act[1][1];

  if (nstage > 0) issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage s has landed
    GemmBarrier();                                     // ... everybody's; and stage s-1 is no longer read
    if (s + 1 < nstage) issue(s + 1);                  // in flight during the MFMAs below
    const double* sa = lds + (s & 1) * 2 * kDmaTile;
    const double* sb = sa + kDmaTile;
    // element (r, 4 sub + kq): m-major images advance 2 row pairs (256 doubles) per sub-step;
    // k-major images rotate by two chunks inside the row.  The operands of sub-step sub + 1 are
    // read while the MFMAs of sub-step sub issue (a lone ds_read waits ~100 cycles for its data).
    double a[2][2], b[2][2];
    auto fetch = [&](int buf, int sub) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        if (AK) {
          const int r = wm + 16 * i + l15;
          a[buf][i] = sa[ImageOffset<true, BK>(r, 4 * sub + kq)];
        } else {
          a[buf][i] = sa[ao[i] + 256 * sub];
        }
        if (BKM) {
          const int r = wn + 16 * i + l15;
          b[buf][i] = sb[ImageOffset<true, BK>(r, 4 * sub + kq)];
        } else {
          b[buf][i] = sb[bo[i] + 256 * sub];
        }
      }
    };
    if (!any_act) continue;
    if (shared_quadrant) {
#pragma unroll
      for (int t = 0; t < (kDmaBK / 4 + 3) / 4; t++) {
        const int sub = wave + 4 * t;
        if (sub < kDmaBK / 4) {
          fetch(0, sub);
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              if (act[i][j])
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
      }
      continue;
    }
    fetch(0, 0);
#pragma unroll
    for (int sub = 0; sub < kDmaBK / 4; sub++) {
      __builtin_amdgcn_sched_barrier(0);
      if (sub + 1 < kDmaBK / 4) fetch((sub + 1) & 1, sub + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          if (act[i][j])
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sub & 1][i], b[sub & 1][j], acc[i][j], 0, 0, 0);
    }
  }
  if (shared_quadrant) {
    // the waves' partial 32 x 32 tiles side by side in LDS (1024 doubles each), summed in wave order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) lds[wave * 1024 + (16 * i + kq + 4 * e) + 32 * (16 * j + l15)] = acc[i][j][e];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int e = tid + 256 * u, m = m_base + (e & 31), n = n_base + (e >> 5);
      if (m < g.M && n < g.N && (!g.lower_only || m >= n)) {
        const double v = ((lds[e] + lds[1024 + e]) + lds[2048 + e]) + lds[3072 + e];
        double* dst = C + m + (int64_t)n * g.ldc;
        if (partial)
          *dst = v;
        else
          *dst = (g.beta == 0.0) ? g.alpha * v : g.alpha * v + g.beta * *dst;
      }
    }
    return;
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
  {
    // 16 elements per thread
    const int mm = tid & 63, m = m_base + mm;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int nn = (tid >> 6) + 4 * u;
      v[u] = lds[mm + 65 * nn];
    }
    if constexpr (!kEarlyC) load_c();
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int nn = (tid >> 6) + 4 * u, n = n_base + nn;
      if (m < g.M && n < g.N && (!g.lower_only || m >= n)) {
        double* dst = C + m + (int64_t)n * g.ldc;
        if (partial)
          *dst = v[u];
        else
          *dst = (g.beta == 0.0) ? g.alpha * v[u] : g.alpha * v[u] + g.beta * c0[u];
      }
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


// ------------------------------------------------------------------------------------------
// LDS-DMA kernel with 128-row tiles: gemm_f64_dma128<TA, TB, NJ>
// ------------------------------------------------------------------------------------------
// The 64 x 64 tile of gemm_f64_dma moves 16 KB from L2 into LDS for 131 kflop (8 flop per byte: 4.7 TB/s
// of L2 -> LDS traffic at half the MFMA peak, which is where that kernel stops), and a tile row that
// holds 8 of its 64 rows (order 200 = 3.1 tiles) costs a workgroup whose time is all DMA latency.
// Here a workgroup owns 128 x 128 (NJ = 4) or 128 x 64 (NJ = 2) of C: up to 4 x NJ MFMA tiles per
// wavefront (16 flop per byte at NJ = 4), a stage of 16 k-values keeps a wavefront's matrix pipe busy
// for ~4000 cycles -- longer than the DMA of the next stage takes -- and the 16-row sub-tiles that exist
// in a ragged tile are dealt EVENLY to the 2 x 2 wavefronts (order 200: 128 + 72 rows = 8 + 5
// sub-tiles, dealt 4 + 4 and 3 + 2), so no wavefront sits on an empty quadrant.  Operand images are
// gemm_f64_dma's (one 64-row image per half of a tile side, same rotations: conflict-free operand
// reads); results leave straight from the accumulators (a 128-byte line of C is completed by the four
// stores of one MFMA tile; the transposed copy is written in 128-byte runs).
template <bool TA, bool TB, int NJ>
__global__ void __launch_bounds__(256, 2) gemm_f64_dma128(GemmArgs g) {
  constexpr int BK = 16, kHalf = 64 * BK, NI = BK / 8;
  constexpr int BM = 128, BN = 32 * NJ, NBH = BN / 64;  // B side: two 64-column images or one
  constexpr int kStage = (2 + NBH) * kHalf;
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (g.M + BM - 1) / BM;
  int tile, split, bz;
  {
    // (workgroups are dealt round-robin over the 8 XCDs: relabelled so that the tiles of one matrix
    // meet in one XCD's L2, as in gemm_f64_dma)
    const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    tile = id % gx;
    split = (id / gx) % gy;
    bz = id / (gx * gy);
  }
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m_base = tm * BM, n_base = tn * BN;
  if (g.lower_only && m_base + BM - 1 < n_base) return;  // uniform per workgroup
  const int b1 = bz / g.inner, b2 = bz % g.inner;
  const double* A = g.A + b1 * g.sA1 + b2 * g.sA2;
  const double* B = g.B + b1 * g.sB1 + b2 * g.sB2;
  const int ksteps = (g.K + kGemmBK - 1) / kGemmBK;
  const int per = (ksteps + g.splits - 1) / g.splits;
  const int k_lo = min(g.K, split * per * kGemmBK), k_hi = min(g.K, k_lo + per * kGemmBK);
  const int nstage = (k_hi - k_lo + BK - 1) / BK;

  constexpr bool AK = TA, BKM = !TB;
  // halves of the tile sides that hold rows / columns of the matrix (wave-uniform)
  const int rows_m = min(BM, g.M - m_base), cols_n = min(BN, g.N - n_base);
  DmaLane da[2], db[NBH];
  const char* a0[2];
  const char* b0[NBH];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int rows = rows_m - 64 * h;
    MakeDmaLane<AK, BK>(da[h], wave, lane, rows > 64 ? 64 : rows, g.lda);
    const int64_t mb = m_base + 64 * h;
    a0[h] = reinterpret_cast<const char*>(A + (AK ? mb * g.lda + k_lo : (int64_t)k_lo * g.lda + mb));
  }
#pragma unroll
  for (int h = 0; h < NBH; h++) {
    const int rows = cols_n - 64 * h;
    MakeDmaLane<BKM, BK>(db[h], wave, lane, rows > 64 ? 64 : rows, g.ldb);
    const int64_t nb = n_base + 64 * h;
    b0[h] = reinterpret_cast<const char*>(B + (BKM ? nb * g.ldb + k_lo : (int64_t)k_lo * g.ldb + nb));
  }
  const int64_t a_step = (AK ? (int64_t)BK : (int64_t)BK * g.lda) * 8;
  const int64_t b_step = (BKM ? (int64_t)BK : (int64_t)BK * g.ldb) * 8;
  const char* zero = reinterpret_cast<const char*>(g_gemm_zero_page) + 16 * lane;

  auto issue = [&](int s) {
    double* st = lds + (s & 1) * kStage;
    const int krem = k_hi - k_lo - s * BK;  // k-values of this stage that exist
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (rows_m > 64 * h) {  // (a half past the edge of the matrix is never read)
        const char* as = a0[h] + (int64_t)s * a_step;
#pragma unroll
        for (int u = 0; u < NI; u++) {
          const int i = wave + 4 * u;
          const char* src = da[h].krel[u] < krem ? as + da[h].off[u] : zero;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                           (__attribute__((address_space(3))) void*)(st + h * kHalf + i * 128), 16, 0, 0);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NBH; h++) {
      if (cols_n > 64 * h) {
        const char* bs = b0[h] + (int64_t)s * b_step;
#pragma unroll
        for (int u = 0; u < NI; u++) {
          const int i = wave + 4 * u;
          const char* src = db[h].krel[u] < krem ? bs + db[h].off[u] : zero;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                           (__attribute__((address_space(3))) void*)(st + (2 + h) * kHalf + i * 128), 16, 0, 0);
        }
      }
    }
  };

  // the 16-row / 16-column sub-tiles of this tile that exist, dealt evenly to the 2 x 2 wavefronts
  const int R = (rows_m + 15) >> 4, Cn = (cols_n + 15) >> 4;
  const int rh = (R + 1) >> 1, ch = (Cn + 1) >> 1;
  const int wi = wave & 1, wj = wave >> 1;
  const int r_lo = wi ? rh : 0, c_lo = wj ? ch : 0;
  int nr = wi ? R - rh : rh, nc = wj ? Cn - ch : ch;
  // (a share wholly above the diagonal of a lower-only call has nothing to write)
  if (g.lower_only && m_base + 16 * (r_lo + nr) - 1 < n_base + 16 * c_lo) nr = 0;
  if (nr == 0 || nc == 0) nr = nc = 0;
  const int l15 = lane & 15, kq = lane >> 4;
  gemm_d4 acc[4][NJ];
  // operand rows of this lane inside the images (sub-tiles past the wavefront's share re-read its first)
  int ar[4], br[NJ];
#pragma unroll
  for (int i = 0; i < 4; i++) ar[i] = 16 * (r_lo + (i < nr ? i : 0)) + l15;
#pragma unroll
  for (int j = 0; j < NJ; j++) br[j] = 16 * (c_lo + (j < nc ? j : 0)) + l15;
  int ao[4], bo[NJ];  // m-major images: offset at k-sub-step 0 (256 doubles further per sub-step)
#pragma unroll
  for (int i = 0; i < 4; i++) ao[i] = (ar[i] >> 6) * kHalf + ImageOffset<false, BK>(ar[i] & 63, kq);
#pragma unroll
  for (int j = 0; j < NJ; j++) bo[j] = (2 + (br[j] >> 6)) * kHalf + ImageOffset<false, BK>(br[j] & 63, kq);

  double* C = g.C + b1 * g.sC1 + b2 * g.sC2 + (g.splits > 1 ? split * g.sCs : 0);
  const bool partial = g.splits > 1;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = gemm_d4{0.0, 0.0, 0.0, 0.0};

  // The K loop, compiled once per size of a share (NR x NC sub-tiles): guards inside the loop -- one per
  // MFMA, or one per 2 x 2 group -- either moved the MFMAs out of line or made a share of 3 x 3 cost 4 x 4.
  auto stages = [&](auto nrc, auto ncc) {
    constexpr int NR = decltype(nrc)::value, NC = decltype(ncc)::value;
    if (nstage > 0) issue(0);
    for (int s = 0; s < nstage; s++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage s has landed
      GemmBarrier();                                     // ... everybody's; and stage s-1 is no longer read
      if (s + 1 < nstage) issue(s + 1);                  // in flight during the MFMAs below
      if constexpr (NR > 0 && NC > 0) {
        const double* st = lds + (s & 1) * kStage;
        double a[2][NR], b[2][NC];
        auto fetch = [&](int buf, int sub) {
#pragma unroll
          for (int i = 0; i < NR; i++) {
            if (AK)
              a[buf][i] = st[(ar[i] >> 6) * kHalf + ImageOffset<true, BK>(ar[i] & 63, 4 * sub + kq)];
            else
              a[buf][i] = st[ao[i] + 256 * sub];
          }
#pragma unroll
          for (int j = 0; j < NC; j++) {
            if (BKM)
              b[buf][j] = st[(2 + (br[j] >> 6)) * kHalf + ImageOffset<true, BK>(br[j] & 63, 4 * sub + kq)];
            else
              b[buf][j] = st[bo[j] + 256 * sub];
          }
        };
        fetch(0, 0);
#pragma unroll
        for (int sub = 0; sub < BK / 4; sub++) {
          __builtin_amdgcn_sched_barrier(0);
          if (sub + 1 < BK / 4) fetch((sub + 1) & 1, sub + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NR; i++)
#pragma unroll
            for (int j = 0; j < NC; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sub & 1][i], b[sub & 1][j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  auto with_rows = [&](auto nrc) {
    if constexpr (NJ == 4) {
      if (nc == 4) return stages(nrc, std::integral_constant<int, 4>{});
      if (nc == 3) return stages(nrc, std::integral_constant<int, 3>{});
    }
    if (nc == 2) return stages(nrc, std::integral_constant<int, 2>{});
    return stages(nrc, std::integral_constant<int, 1>{});
  };
  switch (nr) {
    case 4: with_rows(std::integral_constant<int, 4>{}); break;
    case 3: with_rows(std::integral_constant<int, 3>{}); break;
    case 2: with_rows(std::integral_constant<int, 2>{}); break;
    case 1: with_rows(std::integral_constant<int, 1>{}); break;
    default: stages(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); break;
  }
  // Results: element e of a lane is (row (lane >> 4) + 4 e, column lane & 15) of its MFMA tile.  They leave
  // through LDS, 32 rows of the tile at a time (column c of the strip at 33 c): C is then read (an
  // accumulating call) and written in runs of 32 rows = 256 bytes per column, the transposed copy in runs
  // of BN columns -- written straight from the accumulators (32-byte pieces) the SYRK shapes ran at half
  // the 64 x 64 kernel's rate.
  double* Ct = (g.Ct && !partial) ? g.Ct + b1 * g.sT1 + b2 * g.sT2 : nullptr;
  constexpr int LDP = 33;
  static_assert(BN * LDP <= 2 * kStage, "the strip fits the operand stages");
  for (int p = 0; p < (rows_m + 31) / 32; p++) {
    __syncthreads();  // the operand images / the strip before this one are no longer read
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int st = r_lo + i;  // sub-tile row of the tile
      if (i < nr && (st >> 1) == p) {  // (wave-uniform)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          if (j < nc) {
            const int c = 16 * (c_lo + j) + l15;
#pragma unroll
            for (int e = 0; e < 4; e++) lds[c * LDP + 16 * (st & 1) + kq + 4 * e] = acc[i][j][e];
          }
      }
    }
    __syncthreads();
    {
      const int row = tid & 31, m = m_base + 32 * p + row;
#pragma unroll
      for (int q = 0; q < BN / 8; q++) {
        const int c = (tid >> 5) + 8 * q, n = n_base + c;
        if (m < g.M && n < g.N && (!g.lower_only || m >= n)) {
          const double v = lds[c * LDP + row];
          C[m + (int64_t)n * g.ldc] = partial ? v : g.alpha * v;  // (beta == 0 here: ChooseTile)
        }
      }
    }
    if (Ct) {
      const int c = tid % BN, n = n_base + c;
#pragma unroll
      for (int q = 0; q < 32 * BN / 256; q++) {
        const int row = tid / BN + (256 / BN) * q, m = m_base + 32 * p + row;
        if (m < g.M && n < g.N)
          Ct[n + (int64_t)m * g.ldct + (g.ctb > 0 ? (int64_t)(n / g.ctb) * g.sTb : 0)] = g.alpha * lds[c * LDP + row];
      }
    }
  }
}

namespace {

bool Aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// The DMA path fetches 16-byte chunks (two doubles): every chunk must be aligned and inside its
// matrix.  m-major operands: even leading dimension (a pair of rows of one column).  k-major
// operands: even leading dimension and even K (a pair of k of one row).
bool DmaEligible(const GemmArgs& g, bool ta, bool tb) {
  if (getenv("CXK_GEMM_GENERAL")) return false;  // comparison runs
  if (g.M < 2 || g.N < 2 || g.K < 2) return false;
  if ((g.lda & 1) || (g.ldb & 1) || (g.sA1 & 1) || (g.sA2 & 1) || (g.sB1 & 1) || (g.sB2 & 1)) return false;
  if (!Aligned16(g.A) || !Aligned16(g.B)) return false;
  if ((ta || !tb) && (g.K & 1)) return false;
  // byte offsets inside a tile are kept in 32 bits
  if ((ta ? 64 * g.lda : (int64_t)g.K * g.lda) * 8 >= (1ll << 31)) return false;
  if ((!tb ? 64 * g.ldb : (int64_t)g.K * g.ldb) * 8 >= (1ll << 31)) return false;
  return true;
}

template <bool TA, bool TB, int BK>
hipError_t LaunchDmaBK(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  constexpr size_t lds = sizeof(double) * (4 * 64 * BK > 64 * 65 ? 4 * 64 * BK : 64 * 65);
  static PerDeviceOnce once;
  const hipError_t ec = once.run([] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_dma<TA, TB, BK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (ec != hipSuccess) return ec;
  gemm_f64_dma<TA, TB, BK><<<grid, 256, lds, stream>>>(g);
  return hipGetLastError();
}

template <bool TA, bool TB, int NJ>
hipError_t LaunchDma128T(const GemmArgs& g, int batch, hipStream_t stream) {
  constexpr size_t lds = sizeof(double) * 2 * (2 + NJ / 2) * 64 * 16;
  static PerDeviceOnce once;
  const hipError_t ec = once.run([] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_dma128<TA, TB, NJ>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (ec != hipSuccess) return ec;
  const dim3 grid(((g.M + 127) / 128) * ((g.N + 32 * NJ - 1) / (32 * NJ)), g.splits > 1 ? g.splits : 1, batch);
  gemm_f64_dma128<TA, TB, NJ><<<grid, 256, lds, stream>>>(g);
  return hipGetLastError();
}

template <bool TA, bool TB>
hipError_t LaunchDma128(const GemmArgs& g, int nj, int batch, hipStream_t stream) {
  return nj == 4 ? LaunchDma128T<TA, TB, 4>(g, batch, stream) : LaunchDma128T<TA, TB, 2>(g, batch, stream);
}

// Which tile a DMA-eligible product takes: 0 = the 64 x 64 kernel, 4 = 128 x 128, 2 = 128 x 64.  The big
// tiles need enough workgroups to fill the chip (two per CU are resident) and more than one 64-row tile
// of rows to be worth it; CXK_GEMM_TILE=64 | 128 | 12864 forces the choice (comparison runs).
int ChooseTile(const GemmArgs& g, int batch) {
  static const int forced = [] {
    const char* e = getenv("CXK_GEMM_TILE");
    return e ? atoi(e) : 0;
  }();
  if (forced == 64) return 0;
  if (g.M <= 32 && g.N <= 32) return 0;  // (the Gram products share one quadrant: gemm_f64_dma)
  if (g.beta != 0.0 && g.splits <= 1) return 0;  // (accumulating calls: the 64 x 64 kernel fetches the old values before its K loop)
  if (forced == 128) return 4;
  if (forced == 12864) return 2;
  if (g.M <= 64) return 0;
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // (a ragged last tile of 1 .. 6 sub-tiles a side leaves the four workgroups of an order-200 matrix with
  // shares of 16 / 12 / 12 / 9 MFMA tiles: measured 0.377 against 0.391 of the MFMA peak for the 64 x 64 tiles)
  auto clean = [](int n, int t) { return n % t == 0 || n % t > t - 16; };
  if (!clean(g.M, 128)) return 0;
  const int64_t reps = (int64_t)(g.splits > 1 ? g.splits : 1) * batch * ((g.M + 127) / 128);
  if (g.N > 64 && clean(g.N, 128) && reps * ((g.N + 127) / 128) >= cus) return 4;
  if (clean(g.N, 64) && reps * ((g.N + 63) / 64) >= 2 * cus) return 2;
  return 0;
}

template <bool TA, bool TB>
hipError_t LaunchDma(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  const int ksteps = (g.K + kGemmBK - 1) / kGemmBK;
  const int per_split = (ksteps + std::max(g.splits, 1) - 1) / std::max(g.splits, 1) * kGemmBK;
  if (per_split >= 512) return LaunchDmaBK<TA, TB, 32>(g, grid, stream);
  return LaunchDmaBK<TA, TB, 16>(g, grid, stream);
}

}  // namespace

hipError_t LaunchGemm(const GemmArgs& g, bool ta, bool tb, int batch, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return hipSuccess;
  const int tiles = ((g.M + kGemmBM - 1) / kGemmBM) * ((g.N + kGemmBN - 1) / kGemmBN);
  dim3 grid(tiles, g.splits > 1 ? g.splits : 1, batch);
  if (DmaEligible(g, ta, tb)) {
    if (const int nj = ChooseTile(g, batch)) {
      if (!ta && !tb) return LaunchDma128<false, false>(g, nj, batch, stream);
      if (ta && !tb) return LaunchDma128<true, false>(g, nj, batch, stream);
      if (!ta && tb) return LaunchDma128<false, true>(g, nj, batch, stream);
      return LaunchDma128<true, true>(g, nj, batch, stream);
    }
    const int tiles_m = (g.M + kGemmBM - 1) / kGemmBM;
    if (g.lower_only && g.M == g.N) grid.x = tiles_m * (tiles_m + 1) / 2;  // lower tiles only
    if (!ta && !tb) return LaunchDma<false, false>(g, grid, stream);
    if (ta && !tb) return LaunchDma<true, false>(g, grid, stream);
    if (!ta && tb) return LaunchDma<false, true>(g, grid, stream);
    return LaunchDma<true, true>(g, grid, stream);
  }
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

hipError_t LaunchGemmSplitK(GemmArgs g, bool ta, bool tb, int batch, double* part, hipStream_t stream) {
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
