// Supernodal KKT kernels: deterministic gather-assembly of the slab, level-scheduled
// left-looking block Cholesky, and level-scheduled block triangular solves.
//
// Reference semantics reproduced here (summation ORDER included, so results do not depend
// on scheduling):
//   SupernodalAssemblerBase::UpdateBlocks (Set/SetLowerTri/Scatter)  supernodal_assembler.cc:113-165
//   SupernodalKKTSolver::Assemble (descending elimination index)     kkt_solver.cc:164-170
//   AssembleSchurComplementResiduals                                 constraint_manager.h:107-124
//   BlockCholeskyInPlace                                             block_triangular_operations.cc:184-219
//   ApplyBlockInverseInPlace / ...OfTransposeInPlace                 block_triangular_operations.cc:114-182
//
// The reference pushes updates through tables of double*; here every target entry PULLS its
// contributions from an index list built on the host in the reference's own order.  Entries
// are owned by exactly one thread, so no atomics are needed and runs are bit-reproducible.
#pragma once
#include "device_utils.h"

namespace cxk {

// Assembly gather, one launch:
//   slab[dst[t]] = sum_k G[src[k]], k in [ptr[t], ptr[t+1]) ; src < 0 means structural zero
//   AW/AQc in permuted order (constraint order sums), the two scalars, and -- when with_rhs --
//   y = k (b bs + AQc cs) - 2 AW  (cone_program.cc:409-411).  Also clears the factor flag.
struct GatherArgs {
  int64_t T;
  const int64_t* dst;
  const int* ptr;
  const int64_t* src;
  const double* G;
  double* slab;
  int N;
  const int* rs_ptr;
  const int64_t* rs_src;
  const double* AWc;
  const double* AQcc;
  double* AW;
  double* AQc;
  int K;
  const double* sc;
  double* sys_sc;
  int with_rhs;
  double k, bs, cs;
  const double* b;
  double* y;
  int* fail;
};

__global__ void __launch_bounds__(256) assemble_gather(GatherArgs a) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = gid; t < a.T; t += stride) {
    double s = 0;
    for (int k = a.ptr[t]; k < a.ptr[t + 1]; k++) {
      const int64_t q = a.src[k];
      if (q >= 0) s += a.G[q];
    }
    a.slab[a.dst[t]] = s;
  }
  for (int64_t p = gid; p < a.N; p += stride) {
    double aw = 0, aq = 0;
    for (int k = a.rs_ptr[p]; k < a.rs_ptr[p + 1]; k++) {
      aw += a.AWc[a.rs_src[k]];
      aq += a.AQcc[a.rs_src[k]];
    }
    a.AW[p] = aw;
    a.AQc[p] = aq;
    if (a.with_rhs) a.y[p] = a.k * (a.b[p] * a.bs + aq * a.cs) - 2 * aw;
  }
  if (blockIdx.x == 0) {  // <w,c> and <c,Qc>: fixed-order strided partial sums + block sum
    __shared__ double red[8];
    double s0 = 0, s1 = 0;
    for (int i = threadIdx.x; i < a.K; i += blockDim.x) {
      s0 += a.sc[2 * i];
      s1 += a.sc[2 * i + 1];
    }
    s0 = BlockSum(s0, red);
    s1 = BlockSum(s1, red);
    if (threadIdx.x == 0) {
      a.sys_sc[0] = s0;
      a.sys_sc[1] = s1;
      *a.fail = 0;
    }
  }
}

// y = k (b bs + AQc cs) - 2 AW   (cone_program.cc:409-411), all in permuted order
__global__ void build_rhs(int N, double k, double bs, double cs, const double* __restrict__ b,
                          const double* __restrict__ AQc, const double* __restrict__ AW,
                          double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = k * (b[p] * bs + AQc[p] * cs) - 2 * AW[p];
}

// y = cb b + cq AQc + cw AW : every right-hand side of the IPM loop (cone_program.cc:181, 409-411, 504)
__global__ void build_rhs_comb(int N, double cb, double cq, double cw, const double* __restrict__ b,
                               const double* __restrict__ AQc, const double* __restrict__ AW,
                               double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = cb * b[p] + cq * AQc[p] + cw * AW[p];
}

// The scalars the host loop needs per iteration (cone_program.cc:343-357, 439-446):
// out = { b.y, AQc.y, |b|^2, |AQc|^2, <w,c>, <c,Qc> } ; one workgroup, fixed summation order.
__global__ void __launch_bounds__(1024)
step_scalars(int N, const double* __restrict__ b, const double* __restrict__ AQc,
             const double* __restrict__ y, const double* __restrict__ sys_sc,
             double* __restrict__ out) {
  __shared__ double red[16];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int p = threadIdx.x; p < N; p += blockDim.x) {
    s0 = fma(b[p], y[p], s0);
    s1 = fma(AQc[p], y[p], s1);
    s2 = fma(b[p], b[p], s2);
    s3 = fma(AQc[p], AQc[p], s3);
  }
  s0 = BlockSum(s0, red);
  s1 = BlockSum(s1, red);
  s2 = BlockSum(s2, red);
  s3 = BlockSum(s3, red);
  if (threadIdx.x == 0) {
    out[0] = s0;
    out[1] = s1;
    out[2] = s2;
    out[3] = s3;
    out[4] = sys_sc[0];
    out[5] = sys_sc[1];
  }
}

// y = AQc cs - b bs  (ComputeMuFromDivergence cone_program.cc:181)
__global__ void build_mu_rhs(int N, double bs, double cs, const double* __restrict__ b,
                             const double* __restrict__ AQc, double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = AQc[p] * cs - b[p] * bs;
}

// In-kernel stamps (diagnostic builds only: -DCXK_DEBUG_STAMPS); values go to a buffer nothing else reads.
#ifdef CXK_DEBUG_STAMPS
__device__ long long g_cxk_stamp[16];
#define CXK_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_cxk_stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CXK_STAMP(i) do { } while (0)
#endif

struct FactorPlan {
  // per supernode
  const int* ns;             // [K]
  const int* nsep;           // [K]
  const int* start;          // [K] first permuted index
  const int64_t* diag_off;   // [K]
  const int64_t* offd_off;   // [K]
  // Every supernode publishes its Schur update  U[k,j] = off[:,k].off[:,j] (k <= j, the
  // reference's S_S enumeration) and its forward-solve update t[c] = off[:,c].b  into private
  // slots; ancestors pull single values in increasing child index (the reference's order).
  const int64_t* upd_off;    // [K] offset of the s(s+1)/2 values in `upd`
  const int* updb_off;       // [K] offset of the s values in `updb`
  const int* tg_ptr;         // [K+1] targets of supernode p
  const int* tg_loc;         // local offset inside [diag ns x ns | off ns x s]
  const int* tr_ptr;         // [T+1] contributions of target t
  const int64_t* tr_src;     // index into `upd`
  const int* fs_ptr;         // [N+1] contributions of permuted row r
  const int* fs_src;         // index into `updb`
  // backward: separator columns in the reference's accumulation order
  const int* bs_ptr;         // [K+1]
  const int* bs_c;           // column index c within off block
  const int* bs_row;         // permuted index of separator variable
  double* upd;
  double* updb;
};

// Stage [diag | off | rhs] of supernode p into the wave's LDS region and apply the published
// updates of its descendants in the reference's order.  Layout: sD ns*ns, sB ns*s, sb ns.
__device__ inline void StageAndPull(const FactorPlan& P, int p, const double* __restrict__ slab,
                                    const double* __restrict__ rhs, double* __restrict__ my,
                                    bool with_matrix) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]), s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
  const double* D = slab + P.diag_off[p];
  const double* B = slab + P.offd_off[p];
  double* sb = my + ns * ns + ns * s;
  const int st = __builtin_amdgcn_readfirstlane(P.start[p]);
  // copy [diag | off] into LDS; loads are issued in independent batches of 8 so their latencies
  // overlap (a plain copy loop waits for each load before the next one is issued)
  {
    const int nd = ns * ns, total = with_matrix ? nd + ns * s : nd;
    for (int base = 0; base < total; base += 8 * 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int q = base + u * 64 + lane;
        v[u] = (q < total) ? (q < nd ? D[q] : B[q - nd]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int q = base + u * 64 + lane;
        if (q < total) my[q] = v[u];
      }
    }
  }
  if (rhs)
    for (int r = lane; r < ns; r += 64) sb[r] = rhs[st + r];
  WaveSync();
  if (with_matrix) {
    for (int t = P.tg_ptr[p] + lane; t < P.tg_ptr[p + 1]; t += 64) {
      const int loc = P.tg_loc[t];
      double acc = my[loc];
      const int q1 = P.tr_ptr[t + 1];
#pragma unroll 4
      for (int q = P.tr_ptr[t]; q < q1; q++) acc -= P.upd[P.tr_src[q]];
      my[loc] = acc;
    }
  }
  if (rhs) {
    for (int r = lane; r < ns; r += 64) {
      double acc = sb[r];
      const int q1 = P.fs_ptr[st + r + 1];
#pragma unroll 4
      for (int q = P.fs_ptr[st + r]; q < q1; q++) acc -= P.updb[P.fs_src[q]];
      sb[r] = acc;
    }
  }
  WaveSync();
}

// Publish U[k,j] = off[:,k].off[:,j] and t[c] = off[:,c].b from the LDS copies sB / sb.
__device__ inline void PublishUpdates(const FactorPlan& P, int p, const double* __restrict__ my,
                                      bool with_matrix, bool with_rhs) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]), s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
  const double* sB = my + ns * ns;
  const double* sb = sB + ns * s;
  if (with_matrix) {
    double* out = P.upd + P.upd_off[p];
    const int npairs = s * (s + 1) / 2;
    for (int t = lane; t < npairs; t += 64) {
      int k = 0, rem = t;
      while (rem >= s - k) {
        rem -= s - k;
        k++;
      }
      const int j = k + rem;
      double dot = 0;
      for (int i = 0; i < ns; i++) dot = fma(sB[i + k * ns], sB[i + j * ns], dot);
      out[t] = dot;
    }
  }
  if (with_rhs) {
    double* out = P.updb + P.updb_off[p];
    for (int c = lane; c < s; c += 64) {
      double dot = 0;
      for (int i = 0; i < ns; i++) dot = fma(sB[i + c * ns], sb[i], dot);
      out[c] = dot;
    }
  }
}

// ---------------------------------------------------------------------------------------
// One wavefront factors one supernode.  Lane j owns COLUMN j of the diagonal block (col[i],
// static register indices) and ROW j of the extra columns [off-diagonal block | rhs] (ext[c]).
// The column loop k is a real loop (compact code: a fully unrolled body is executed once per
// wave and is instruction-fetch bound); per step the pivot lane scales its column and
// publishes it through a 16/32-entry LDS line, every lane applies the rank-1 update to its own
// column, and the extra columns travel by v_readlane with a run-time lane select.
// Right-looking: same update order per entry as the reference's column-by-column LLT + TRSM.
// ---------------------------------------------------------------------------------------
template <int NSMAX, int SMAX>
__device__ inline void CholSupernodeReg(const FactorPlan& P, int p, double* __restrict__ slab,
                                        double* __restrict__ rhs, int* __restrict__ fail,
                                        double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]), s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
  double* sD = my;
  double* sB = my + ns * ns;
  double* sb = sB + ns * s;
  CXK_STAMP(0);
  StageAndPull(P, p, slab, rhs, my, true);
  CXK_STAMP(1);
  const bool active = lane < ns;
  double col[NSMAX];   // column `lane`, rows 0..NSMAX-1 (rows < lane are never read)
  double ext[SMAX + 1];
#pragma unroll
  for (int i = 0; i < NSMAX; i++) col[i] = (active && i < ns && i >= lane) ? sD[i + lane * ns] : 0.0;
#pragma unroll
  for (int c = 0; c < SMAX; c++) ext[c] = (active && c < s) ? sB[lane + c * ns] : 0.0;
  ext[SMAX] = (rhs && active) ? sb[lane] : 0.0;
  double diag = active ? sD[lane + lane * ns] : 1.0;
  WaveSync();
  double* piv = my;  // reuse the head of the staging area as the pivot-column line
  CXK_STAMP(2);
  bool bad = false;
#pragma unroll 1
  for (int k = 0; k < ns; k++) {
    if (lane == k) {
      if (!(diag > 0.0)) bad = true;
      const double d = sqrt(diag);
      const double rd = 1.0 / d;  // one divide per column; scalings are multiplies
      diag = d;
      // rows <= k of this column are never read again, so the whole register column is scaled
      // and published without per-row selects
#pragma unroll
      for (int i = 0; i < NSMAX; i++) {
        col[i] *= rd;
        piv[i] = col[i];
      }
#pragma unroll
      for (int c = 0; c <= SMAX; c++) ext[c] *= rd;
    }
    WaveSync();
    // m = L[lane][k] for lanes below the pivot, 0 elsewhere (finished columns / solved rows)
    const double m = (lane > k && lane < NSMAX) ? piv[lane] : 0.0;
    diag = fma(-m, m, diag);
#pragma unroll
    for (int i = 0; i < NSMAX; i++) col[i] = fma(-piv[i], m, col[i]);
#pragma unroll
    for (int c = 0; c <= SMAX; c++) {
      const double ekc = ReadLane(ext[c], k);
      ext[c] = fma(-m, ekc, ext[c]);
    }
    WaveSync();
  }
  CXK_STAMP(3);
  if (__any(bad)) {
    if (lane == 0) atomicExch(fail, 1);
    return;
  }
  double* D = slab + P.diag_off[p];
  double* B = slab + P.offd_off[p];
  if (active) {
#pragma unroll
    for (int i = 0; i < NSMAX; i++)
      if (i < ns && i > lane) D[i + (size_t)lane * ns] = col[i];
    D[lane + (size_t)lane * ns] = diag;
#pragma unroll
    for (int c = 0; c < SMAX; c++)
      if (c < s) {
        B[lane + (size_t)c * ns] = ext[c];
        sB[lane + c * ns] = ext[c];
      }
    if (rhs) {
      rhs[P.start[p] + lane] = ext[SMAX];
      sb[lane] = ext[SMAX];
    }
  }
  WaveSync();
  CXK_STAMP(4);
  PublishUpdates(P, p, my, true, rhs != nullptr);
  CXK_STAMP(5);
}

// LDS-resident fallback for supernodes that do not fit the register kernels.
__device__ inline void CholSupernodeLds(const FactorPlan& P, int p, double* __restrict__ slab,
                                        double* __restrict__ rhs, int* __restrict__ fail,
                                        double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]), s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
  double* D = slab + P.diag_off[p];
  double* B = slab + P.offd_off[p];
  double* sD = my;
  double* sB = my + ns * ns;
  double* sb = sB + ns * s;
  StageAndPull(P, p, slab, rhs, my, true);
  const int ncols = s + (rhs ? 1 : 0);
  bool bad = false;
  for (int k = 0; k < ns; k++) {
    const double akk = sD[k + k * ns];
    if (!(akk > 0.0)) {
      bad = true;
      break;
    }
    const double d = sqrt(akk);
    WaveSync();
    for (int i = k + lane; i < ns; i += 64) sD[i + k * ns] = (i == k) ? d : sD[i + k * ns] / d;
    for (int c = lane; c < ncols; c += 64) {
      double* col = (c < s) ? sB + c * ns : sb;
      col[k] /= d;
    }
    WaveSync();
    for (int i = k + 1 + lane; i < ns; i += 64) {
      const double lik = sD[i + k * ns];
      for (int j = k + 1; j <= i; j++) sD[i + j * ns] -= lik * sD[j + k * ns];
      for (int c = 0; c < ncols; c++) {
        double* col = (c < s) ? sB + c * ns : sb;
        col[i] -= lik * col[k];
      }
    }
    WaveSync();
  }
  if (bad) {
    if (lane == 0) atomicExch(fail, 1);
    return;
  }
  for (int q = lane; q < ns * ns; q += 64) {
    const int i = q % ns, j = q / ns;
    if (i >= j) D[q] = sD[q];
  }
  for (int q = lane; q < ns * s; q += 64) B[q] = sB[q];
  if (rhs)
    for (int r = lane; r < ns; r += 64) rhs[P.start[p] + r] = sb[r];
  PublishUpdates(P, p, my, true, rhs != nullptr);
}

// b_p <- L_p^{-1} (b_p - published updates); publishes t[c] = off[:,c].b_p.
// Lane i owns b_i; L stays in LDS; the solved entry travels by v_readlane.
__device__ inline void ForwardSupernodeWave(const FactorPlan& P, int p,
                                            const double* __restrict__ slab,
                                            double* __restrict__ rhs, double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]), s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
  double* sD = my;
  double* sB = my + ns * ns;
  double* sb = sB + ns * s;
  const double* B = slab + P.offd_off[p];
  StageAndPull(P, p, slab, rhs, my, false);
  for (int q = lane; q < ns * s; q += 64) sB[q] = B[q];
  const bool active = lane < ns;
  double b = active ? sb[lane] : 0.0;
  const double dinv = active ? 1.0 / sD[lane + lane * ns] : 0.0;
#pragma unroll 1
  for (int k = 0; k < ns; k++) {
    if (lane == k) b *= dinv;
    const double bk = ReadLane(b, k);
    const double lik = (active && lane > k) ? sD[lane + k * ns] : 0.0;
    b -= lik * bk;
  }
  if (active) {
    rhs[P.start[p] + lane] = b;
    sb[lane] = b;
  }
  WaveSync();
  PublishUpdates(P, p, my, false, true);
}

__device__ inline void ForwardSupernodeLds(const FactorPlan& P, int p,
                                           const double* __restrict__ slab,
                                           double* __restrict__ rhs, double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]), s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
  double* sD = my;
  double* sB = my + ns * ns;
  double* sb = sB + ns * s;
  const double* B = slab + P.offd_off[p];
  StageAndPull(P, p, slab, rhs, my, false);
  for (int q = lane; q < ns * s; q += 64) sB[q] = B[q];
  for (int k = 0; k < ns; k++) {
    const double bk = sb[k] / sD[k + k * ns];
    WaveSync();
    if (lane == 0) sb[k] = bk;
    for (int i = k + 1 + lane; i < ns; i += 64) sb[i] -= sD[i + k * ns] * bk;
    WaveSync();
  }
  for (int r = lane; r < ns; r += 64) rhs[P.start[p] + r] = sb[r];
  PublishUpdates(P, p, my, false, true);
}

// b_j <- L_j^{-T} (b_j - sum_c off_j[:,c] y[sep_j[c]]) for ns <= 64; lane i owns y_i.
__device__ inline void BackwardSupernodeWave(const FactorPlan& P, int p,
                                             const double* __restrict__ slab,
                                             double* __restrict__ rhs, double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]);
  const double* D = slab + P.diag_off[p];
  const double* B = slab + P.offd_off[p];
  double* sD = my;
  const bool active = lane < ns;
  const int st = __builtin_amdgcn_readfirstlane(P.start[p]);
  for (int q = lane; q < ns * ns; q += 64) sD[q] = D[q];
  double acc = active ? rhs[st + lane] : 0.0;
  const int q0 = P.bs_ptr[p], q1 = P.bs_ptr[p + 1];
#pragma unroll 4
  for (int q = q0; q < q1; q++) {
    const double yv = rhs[P.bs_row[q]];
    if (active) acc -= B[lane + (size_t)P.bs_c[q] * ns] * yv;
  }
  WaveSync();
  const double dinv = active ? 1.0 / sD[lane + lane * ns] : 0.0;
#pragma unroll 1
  for (int k = ns - 1; k >= 0; k--) {
    if (lane == k) acc *= dinv;
    const double yk = ReadLane(acc, k);
    const double lki = (lane < k) ? sD[k + lane * ns] : 0.0;  // L[k][lane]
    acc -= lki * yk;
  }
  if (active) rhs[st + lane] = acc;
}

__device__ inline void BackwardSupernodeLds(const FactorPlan& P, int p,
                                            const double* __restrict__ slab,
                                            double* __restrict__ rhs, double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]);
  const double* D = slab + P.diag_off[p];
  const double* B = slab + P.offd_off[p];
  double* sD = my;
  double* sb = my + ns * ns;
  const int st = __builtin_amdgcn_readfirstlane(P.start[p]);
  for (int q = lane; q < ns * ns; q += 64) sD[q] = D[q];
  for (int r = lane; r < ns; r += 64) {
    double acc = rhs[st + r];
    for (int q = P.bs_ptr[p]; q < P.bs_ptr[p + 1]; q++)
      acc -= B[r + (size_t)P.bs_c[q] * ns] * rhs[P.bs_row[q]];
    sb[r] = acc;
  }
  WaveSync();
  for (int k = ns - 1; k >= 0; k--) {
    const double yk = sb[k] / sD[k + k * ns];
    WaveSync();
    if (lane == 0) sb[k] = yk;
    for (int i = lane; i < k; i += 64) sb[i] -= sD[k + i * ns] * yk;
    WaveSync();
  }
  for (int r = lane; r < ns; r += 64) rhs[st + r] = sb[r];
}

// mode 0: factor (+ forward if rhs), mode 1: forward only, mode 2: backward.
// Levels [lb, le) ascending for modes 0/1; mode 2 walks the range downwards.  A launch that
// covers several levels (or continues into the backward sweep) must be ONE workgroup: levels
// are then separated by a workgroup barrier instead of a kernel boundary.
__global__ void __launch_bounds__(512)
tree_sweep(FactorPlan P, const int* __restrict__ level_ptr, const int* __restrict__ level_sn, int lb,
           int le, int mode, int then_backward, double* __restrict__ slab, double* __restrict__ rhs,
           int* __restrict__ fail, int lds_per_wave) {
  extern __shared__ double lds[];
  // wave-uniform values are forced into SGPRs: otherwise every loop bound / lane select below
  // is treated as divergent (waterfall loops around v_readlane, vector address arithmetic)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* my = lds + (size_t)wave * lds_per_wave;
  const bool multi = (le - lb > 1) || then_backward;
  if (mode != 2) {
    for (int l = lb; l < le; l++) {
      const int base = level_ptr[l], cnt = level_ptr[l + 1] - base;
      for (int idx = blockIdx.x * nw + wave; idx < cnt; idx += gridDim.x * nw) {
        const int p = __builtin_amdgcn_readfirstlane(level_sn[base + idx]);
        const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]);
        const int s = __builtin_amdgcn_readfirstlane(P.nsep[p]);
        if (mode == 0) {
          if (ns <= 16 && s <= 8)
            CholSupernodeReg<16, 8>(P, p, slab, rhs, fail, my);
          else if (ns <= 24 && s == 0)
            CholSupernodeReg<24, 0>(P, p, slab, rhs, fail, my);
          else if (ns <= 24 && s <= 8)
            CholSupernodeReg<24, 8>(P, p, slab, rhs, fail, my);
          else if (ns <= 32 && s <= 16)
            CholSupernodeReg<32, 16>(P, p, slab, rhs, fail, my);
          else
            CholSupernodeLds(P, p, slab, rhs, fail, my);
        } else {
          if (ns <= 64)
            ForwardSupernodeWave(P, p, slab, rhs, my);
          else
            ForwardSupernodeLds(P, p, slab, rhs, my);
        }
      }
      if (multi) __syncthreads();
    }
  }
  if (mode == 2 || then_backward) {
    for (int l = le - 1; l >= lb; l--) {
      const int base = level_ptr[l], cnt = level_ptr[l + 1] - base;
      for (int idx = blockIdx.x * nw + wave; idx < cnt; idx += gridDim.x * nw) {
        const int p = __builtin_amdgcn_readfirstlane(level_sn[base + idx]);
        const int ns = __builtin_amdgcn_readfirstlane(P.ns[p]);
        if (ns <= 64)
          BackwardSupernodeWave(P, p, slab, rhs, my);
        else
          BackwardSupernodeLds(P, p, slab, rhs, my);
      }
      if (multi) __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
// Multi-GPU exchange.  Buffer layout, all doubles:
//   [ T slab entries (n_xs) | AW_T (n_xv) | AQc_T (n_xv) | fwd_T (n_xv) | <w,c> | <c,Qc> | fail | pad ]
// pack:   fold this rank's subtree updates into its PARTIAL top blocks (pre-reduce pulls), then
//         copy the partial top, the partial residuals of top variables and the forward-solve
//         contributions of this rank's subtrees into the buffer.
// unpack: after the caller's sum all-reduce the buffer holds the complete assembled-and-updated
//         top; write it back, rebuild the right-hand side of top variables and latch `fail`.
// ---------------------------------------------------------------------------------------
struct ExchangeArgs {
  int64_t n_xs;
  int n_xv;
  const int64_t* xs_off;
  const int* xv_idx;
  int64_t pt_T;
  const int64_t* pt_dst;
  const int* pt_ptr;
  const int64_t* pt_src;
  const int* pf_ptr;
  const int* pf_src;
  const double* upd;
  const double* updb;
  double* slab;
  double* AW;
  double* AQc;
  const double* b;
  double* y;
  double* sys_sc;
  int* fail;
  double* x;
  double cb, cq, cw;
};

__global__ void __launch_bounds__(256) exchange_pack(ExchangeArgs a) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // slab entries: partial assembled value minus the updates of this rank's subtrees (exchange_fold)
  for (int64_t i = gid; i < a.n_xs; i += stride) a.x[i] = a.slab[a.xs_off[i]];
  for (int64_t j = gid; j < a.n_xv; j += stride) {
    const int p = a.xv_idx[j];
    a.x[a.n_xs + j] = a.AW[p];
    a.x[a.n_xs + a.n_xv + j] = a.AQc[p];
    double f = 0;
    for (int q = a.pf_ptr[j]; q < a.pf_ptr[j + 1]; q++) f += a.updb[a.pf_src[q]];
    a.x[a.n_xs + 2 * (int64_t)a.n_xv + j] = f;
  }
  if (gid == 0) {
    const int64_t o = a.n_xs + 3 * (int64_t)a.n_xv;
    a.x[o] = a.sys_sc[0];
    a.x[o + 1] = a.sys_sc[1];
    a.x[o + 2] = (double)(*a.fail);
    a.x[o + 3] = 0;
  }
}

// runs BEFORE exchange_pack: subtracts the Schur updates of this rank's subtrees from its partial
// top blocks, in place in the slab (one thread per top entry)
__global__ void __launch_bounds__(256) exchange_fold(ExchangeArgs a) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < a.pt_T;
       t += (int64_t)gridDim.x * blockDim.x) {
    double s = 0;
    for (int q = a.pt_ptr[t]; q < a.pt_ptr[t + 1]; q++) s += a.upd[a.pt_src[q]];
    a.slab[a.pt_dst[t]] -= s;
  }
}

__global__ void __launch_bounds__(256) exchange_unpack(ExchangeArgs a) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gid; i < a.n_xs; i += stride) a.slab[a.xs_off[i]] = a.x[i];
  for (int64_t j = gid; j < a.n_xv; j += stride) {
    const int p = a.xv_idx[j];
    const double aw = a.x[a.n_xs + j], aq = a.x[a.n_xs + a.n_xv + j];
    a.AW[p] = aw;
    a.AQc[p] = aq;
    a.y[p] = a.cb * a.b[p] + a.cq * aq + a.cw * aw - a.x[a.n_xs + 2 * (int64_t)a.n_xv + j];
  }
  if (gid == 0) {
    const int64_t o = a.n_xs + 3 * (int64_t)a.n_xv;
    a.sys_sc[0] = a.x[o];
    a.sys_sc[1] = a.x[o + 1];
    if (a.x[o + 2] > 0.0) *a.fail = 1;
  }
}

// permuted <-> original order copies
__global__ void permute_gather(int N, const int* __restrict__ idx, const double* __restrict__ in,
                               double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
    out[i] = in[idx[i]];
}

}  // namespace cxk
