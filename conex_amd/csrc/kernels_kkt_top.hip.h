// The last levels of the elimination tree as ONE dense factorization.
//
// Supernodes of 33..64 columns do not fit the row-per-lane register kernels (two DPP rows) and
// otherwise take tree_sweep_block: a 256-thread workgroup per supernode with two barriers per
// column -- 62 us for the 50-column root of BASELINE config 2.  When the last levels of the tree
// hold such a supernode and at most 64 columns in total, the same arithmetic is one right-looking
// Cholesky of the T x T matrix formed by all their variables: the supernodes' diagonal and
// off-diagonal blocks are exactly its lower triangle (entries outside the blocks are structural
// zeros of L and stay zero), the updates those supernodes would publish to each other happen
// inside the elimination, and only updates from below are pulled (consumer-ordered slots, one
// round trip).  Row per lane as in FactorSupernodeRows, up to 64 rows: a column of L spans four
// 16-lane DPP rows, each mirrored to all rows with v_permlane16/32_swap so that every term is one
// row_newbcast fma.  The forward substitution rides along as an extra column; the back
// substitution runs in the same registers (dot product of column j with the solved tail, one
// wave reduction per unknown).  Measured: config 2 176 -> 130 us per KKT solve.  For tops made of
// SMALL supernodes (config 4: 15 + 20 columns) the supernode-by-supernode top is faster (25 us
// against 29 us: a lone wavefront retires a dependent instruction per ~9 cycles, and the wider
// elimination plus the reduction-per-unknown back substitution cost more than the hops they
// save), so the dense range always starts at a level that holds a mid-size supernode.
//
// Same mathematics as block_triangular_operations.cc:114-219 restricted to those levels; sums are
// ordered differently from the supernode-by-supernode form (updates from inside the range are
// applied column by column instead of as one published block), i.e. results agree to rounding.
#pragma once
#include "kernels_kkt.hip.h"

namespace cxk {

constexpr int kTopMaxSn = 16;
constexpr int kTopMaxCols = 64;

constexpr int kTopMaxImage = 4096;  // doubles: panels of the top, and (aliased) their update slots / the dense matrix
constexpr int kTopRhsSrc = 8;       // external forward-solve sources per top row (fixed width)

struct TopDenseArgs {
  int nt, T;                         // supernodes, total columns
  int ns[kTopMaxSn], nsep[kTopMaxSn], start[kTopMaxSn], row0[kTopMaxSn], base[kTopMaxSn];
  long long diag_off[kTopMaxSn], offd_off[kTopMaxSn];
  // consumer-ordered update slots of supernode k: target t (panel position tg_loc[tg_beg + t])
  // reads upd[ubase + t * m + i], i < m.  Slots fed by supernodes INSIDE the top are never written
  // while this kernel does the top (they stay 0.0 and subtract exactly).
  int ubase[kTopMaxSn], m[kTopMaxSn], tg_beg[kTopMaxSn], ntg[kTopMaxSn], ubase_lds[kTopMaxSn], tg_lds[kTopMaxSn];
  const int* top_off;                // [T*T]: image offset of L(r, j), j <= r, or -1
  const int* rhs_src;                // [T * kTopRhsSrc]: updb slots from below the top feeding row r (padded
                                     // with a slot that is always 0.0)
};

struct ElimNoSink {
  __device__ __forceinline__ void operator()(int, double) const {}
};

// sink(j, column): called with column j of L as soon as it is final (lane r holds L[r][j]): a caller
// that stores the factor does it there, where the store's issue time hides behind the next pivot's
// dependent chain (tree_fused.hip).
template <int TM, int J>
struct ElimWide {
  static constexpr int LEN = TM + 1;
  template <typename Sink = ElimNoSink>
  static __device__ __forceinline__ void run(double (&a)[LEN], int lane, int T, bool& bad, Sink sink = Sink()) {
    if constexpr (J < TM) {
      if (J < T) {  // wave-uniform: columns >= T are padding
        const double d = ReadLane(a[J], J);
        bad |= !(d > 0.0);
        double root, inv;
        SqrtAndInverse(d, root, inv);
        a[J] = (lane == J) ? root : a[J] * inv;
        sink(J, a[J]);
        // column J of L, DPP row k mirrored into every row: m[k]
        const RowPair p16 = Swap16(a[J]);  // a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
        double m0 = p16.a, m1 = p16.b, m2 = 0.0, m3 = 0.0;
        if constexpr (TM > 32) {
          const RowPair pa = Swap32(p16.a);  // a = lower half everywhere, b = upper half
          const RowPair pb = Swap32(p16.b);
          m0 = pa.a;
          m2 = pa.b;
          m1 = pb.a;
          m3 = pb.b;
        }
        double naj = -a[J];
        DppOperandFence(m0, m1, naj);
        DppOperandFence(m2, m3, naj);
        constexpr int c0 = J + 1;
        DppColumns<LEN, (c0 > 0 ? c0 : 0), (TM < 16 ? TM : 16), 0>::run(a, m0, naj);
        DppColumns<LEN, (c0 > 16 ? c0 : 16), (TM < 32 ? TM : 32), 16>::run(a, m1, naj);
        if constexpr (TM > 32) {
          DppColumns<LEN, (c0 > 32 ? c0 : 32), (TM < 48 ? TM : 48), 32>::run(a, m2, naj);
          DppColumns<LEN, (c0 > 48 ? c0 : 48), (TM < 64 ? TM : 64), 48>::run(a, m3, naj);
        }
        const double yj = ReadLane(a[TM], J) * inv;
        if (lane > J)
          a[TM] = fma(-yj, a[J], a[TM]);
        else if (lane == J)
          a[TM] = yj;
      }
      ElimWide<TM, J + 1>::run(a, lane, T, bad, sink);
    }
  }
};

// x_J = (x_J - sum_{i > J} L[i][J] x_i) / L[J][J], J = TM-1 .. 0
template <int TM, int J>
struct BackWide {
  static __device__ __forceinline__ void run(const double (&a)[TM + 1], double& x, int lane, int T) {
    if constexpr (J >= 0) {
      if (J < T) {
        const double s = WaveSum((lane > J && lane < T) ? a[J] * x : 0.0);
        if (lane == J) x = (x - s) / a[J];
      }
      BackWide<TM, J - 1>::run(a, x, lane, T);
    }
  }
};

template <int TM, int J>
struct BackWideInv {  // x_J = (x_J - sum_{i > J} L[i][J] x_i) * inv_J
  static __device__ __forceinline__ void run(const double (&a)[TM + 1], double& x, double myinv, int lane, int T) {
    if constexpr (J >= 0) {
      if (J < T) {
        const double s = WaveSum((lane > J && lane < T) ? a[J] * x : 0.0);
        if (lane == J) x = (x - s) * myinv;
      }
      BackWideInv<TM, J - 1>::run(a, x, myinv, lane, T);
    }
  }
};

// LDS (dynamic): doubles sP[kTopMaxImage] | sM[kTopMaxImage] (update slots first, then the dense
// T x TM matrix) | sB[64]; ints sO[T*T <= 4096] | sTg[kTopMaxImage]
constexpr size_t kTopDenseLds = sizeof(double) * (2 * kTopMaxImage + kTopMaxCols) + sizeof(int) * (2 * kTopMaxImage);

template <int TM>
__global__ void __launch_bounds__(256) tree_top_dense(FactorPlan P, TopDenseArgs d, double* __restrict__ slab,
                                                      double* __restrict__ rhs, int* __restrict__ fail,
                                                      int with_rhs, int backward) {
  extern __shared__ double lds[];
  double* sP = lds;
  double* sM = sP + kTopMaxImage;
  double* sB = sM + kTopMaxImage;
  int* sO = reinterpret_cast<int*>(sB + kTopMaxCols);
  int* sTg = sO + kTopMaxImage;
  const int tid = threadIdx.x, T = d.T;
#ifdef CXK_DEBUG_STAMPS
#define TDSTAMP(i) do { if (threadIdx.x == 0) g_cxk_stamp[8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TDSTAMP(i) do { } while (0)
#endif
  TDSTAMP(0);
  // ---- one memory round trip: panels, update slots, their positions, the offset table, the
  // right-hand side and (two dependent loads, on T threads only) the forward-solve values from below
  double rv[kTopRhsSrc];
#pragma unroll
  for (int i = 0; i < kTopRhsSrc; i++) rv[i] = 0.0;
  if (with_rhs && tid >= 64 && tid < 64 + T) {
    const int r = tid - 64;
    int src[kTopRhsSrc];
#pragma unroll
    for (int i = 0; i < kTopRhsSrc; i++) src[i] = d.rhs_src[r * kTopRhsSrc + i];
#pragma unroll
    for (int i = 0; i < kTopRhsSrc; i++) rv[i] = P.updb[src[i]];
  }
#pragma unroll
  for (int k = 0; k < kTopMaxSn; k++)
    if (k < d.nt) {
      const int n2 = d.ns[k] * d.ns[k], no = d.ns[k] * d.nsep[k], nu = d.ntg[k] * d.m[k];
      const double* D = slab + d.diag_off[k];
      const double* B = slab + d.offd_off[k];
      for (int q = tid; q < n2; q += blockDim.x) sP[d.base[k] + q] = D[q];
      for (int q = tid; q < no; q += blockDim.x) sP[d.base[k] + n2 + q] = B[q];
      for (int q = tid; q < nu; q += blockDim.x) sM[d.ubase_lds[k] + q] = P.upd[d.ubase[k] + q];
      for (int q = tid; q < d.ntg[k]; q += blockDim.x) sTg[d.tg_lds[k] + q] = P.tg_loc[d.tg_beg[k] + q];
      if (with_rhs)
        for (int q = tid; q < d.ns[k]; q += blockDim.x) sB[d.row0[k] + q] = rhs[d.start[k] + q];
    }
  for (int q = tid; q < T * T; q += blockDim.x) sO[q] = d.top_off[q];
  __syncthreads();
  TDSTAMP(1);
  // ---- updates from below the top, in the reference's accumulation order (LDS only)
#pragma unroll
  for (int k = 0; k < kTopMaxSn; k++)
    if (k < d.nt) {
      const int m = d.m[k];
      for (int t = tid; t < d.ntg[k]; t += blockDim.x) {
        const int loc = d.base[k] + sTg[d.tg_lds[k] + t];
        double acc = sP[loc];
        const double* u = sM + d.ubase_lds[k] + t * m;
        for (int i = 0; i < m; i++) acc -= u[i];
        sP[loc] = acc;
      }
    }
  if (with_rhs && tid >= 64 && tid < 64 + T) {
    double acc = sB[tid - 64];
#pragma unroll
    for (int i = 0; i < kTopRhsSrc; i++) acc -= rv[i];  // reference order; padding reads a slot that holds 0.0
    sB[tid - 64] = acc;
  }
  __syncthreads();
  TDSTAMP(2);
  // ---- dense lower triangle, row major (stride TM), zeros outside the supernodal structure
  for (int q = tid; q < T * TM; q += blockDim.x) {
    const int r = q / TM, j = q - r * TM;
    const int o = (j <= r) ? sO[r * T + j] : -1;
    sM[q] = o >= 0 ? sP[o] : 0.0;
  }
  __syncthreads();
  TDSTAMP(3);
  if (tid < 64) {
    const int lane = tid;
    const bool row = lane < T;
    double a[TM + 1];
#pragma unroll
    for (int j = 0; j < TM; j++) a[j] = row ? sM[lane * TM + j] : 0.0;
    a[TM] = (with_rhs && row) ? sB[lane] : 0.0;
    bool bad = false;
    ElimWide<TM, 0>::run(a, lane, T, bad);
    TDSTAMP(4);
    if (bad && lane == 0) atomicExch(fail, 1);
    if (row) {
#pragma unroll
      for (int j = 0; j < TM; j++) sM[lane * TM + j] = a[j];
    }
    if (with_rhs) {
      double x = a[TM];
      if (backward) {
        double diag = 1.0;
#pragma unroll
        for (int j = 0; j < TM; j++) diag = (lane == j) ? a[j] : diag;
        BackWideInv<TM, TM - 1>::run(a, x, 1.0 / diag, lane, T);
      }
      if (row) sB[lane] = x;
    }
    TDSTAMP(5);
  }
  __syncthreads();
  for (int q = tid; q < T * TM; q += blockDim.x) {
    const int r = q / TM, j = q - r * TM;
    const int o = (j <= r) ? sO[r * T + j] : -1;
    if (o >= 0) sP[o] = sM[q];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kTopMaxSn; k++)
    if (k < d.nt) {
      const int n2 = d.ns[k] * d.ns[k], no = d.ns[k] * d.nsep[k];
      double* D = slab + d.diag_off[k];
      double* B = slab + d.offd_off[k];
      for (int q = tid; q < n2; q += blockDim.x) D[q] = sP[d.base[k] + q];
      for (int q = tid; q < no; q += blockDim.x) B[q] = sP[d.base[k] + n2 + q];
      if (with_rhs)
        for (int q = tid; q < d.ns[k]; q += blockDim.x) rhs[d.start[k] + q] = sB[d.row0[k] + q];
    }
  TDSTAMP(6);
}

}  // namespace cxk
