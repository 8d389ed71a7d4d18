// Supernodal KKT kernels: deterministic gather-assembly of the slab, level-scheduled block
// Cholesky / LDLT (right-looking inside a supernode, published updates pulled by ancestors),
// and level-scheduled block triangular solves.
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
// One record per slab target / per variable: the FIRST source sits in the record, so the common
// entry (one source: everything outside the separator overlaps) costs two dependent memory hops
// (record, value) instead of three (list bounds, list, value); further sources follow in the lists.
struct GatherRec {
  int64_t dst;    // slab offset
  int64_t first;  // index into G, < 0 = structural zero
  int beg, extra; // remaining sources: src[beg .. beg + extra)
};
struct ResidRec {
  int64_t first;  // index into AWc / AQcc, < 0 = none
  int beg, extra;
};

struct GatherArgs {
  int64_t T;
  const GatherRec* rec;
  const int64_t* src;
  const double* G;
  double* slab;
  int N;
  const ResidRec* rrec;
  const int* var_idx;  // residual record t describes variable var_idx[t] (nullptr: variable t)
  const int64_t* rs_src;
  const double* AWc;
  const double* AQcc;
  double* AW;
  double* AQc;
  int K;
  const double* sc;
  double* sys_sc;
  int with_rhs;  // 1: y = k (b bs + AQc cs) - 2 AW (build_rhs), 2: y = cb b + cq AQc + cw AW (build_rhs_comb)
  double k, bs, cs;
  double cb, cq, cw;
  const double* b;
  double* y;
  int* fail;
};

// The gather of workgroup `block` of `nblocks` (assemble_gather, and the gather workgroups that
// ride in the first factor level's launch: tree_factor_level_asm).
__device__ __forceinline__ void GatherBody(const GatherArgs& a, int block, int nblocks) {
  const int64_t gid = block * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)nblocks * blockDim.x;
  const int64_t span = a.T > a.N ? a.T : (int64_t)a.N;
  // slab entries and residual entries are gathered side by side: their loads share the round trips
  for (int64_t t = gid; t < span; t += stride) {
    const bool ht = t < a.T, hp = t < a.N;
    GatherRec g = {0, -1, 0, 0};
    ResidRec r = {-1, 0, 0};
    int var = (int)t;
    if (ht) g = a.rec[t];
    if (hp) r = a.rrec[t];
    if (hp && a.var_idx) var = a.var_idx[t];
    double s = 0, aw = 0, aq = 0, bp = 0;
    if (g.first >= 0) s += a.G[g.first];
    if (r.first >= 0) {
      aw += a.AWc[r.first];
      aq += a.AQcc[r.first];
    }
    if (hp && a.with_rhs) bp = a.b[var];
    for (int k = g.beg; k < g.beg + g.extra; k++) {
      const int64_t q = a.src[k];
      if (q >= 0) s += a.G[q];
    }
    for (int k = r.beg; k < r.beg + r.extra; k++) {
      aw += a.AWc[a.rs_src[k]];
      aq += a.AQcc[a.rs_src[k]];
    }
    if (ht) a.slab[g.dst] = s;
    if (hp) {
      a.AW[var] = aw;
      a.AQc[var] = aq;
      if (a.with_rhs == 1) a.y[var] = a.k * (bp * a.bs + aq * a.cs) - 2 * aw;
      if (a.with_rhs == 2) a.y[var] = a.cb * bp + a.cq * aq + a.cw * aw;
    }
  }
  if (block == 0) {  // <w,c> and <c,Qc>: fixed-order strided partial sums + block sum
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

#ifndef CXK_DEVICE_FUNCTIONS_ONLY  // plain (non-template) kernels: one translation unit only
__global__ void __launch_bounds__(256) assemble_gather(GatherArgs a) { GatherBody(a, blockIdx.x, gridDim.x); }

// y = k (b bs + AQc cs) - 2 AW   (cone_program.cc:409-411), all in permuted order
// (reset: when not null, the factorization-failure flag cleared here instead of by a memset launch)
__global__ void build_rhs(int N, double k, double bs, double cs, const double* __restrict__ b,
                          const double* __restrict__ AQc, const double* __restrict__ AW,
                          double* __restrict__ y, int* __restrict__ reset = nullptr,
                          const double* __restrict__ k_from = nullptr) {
  if (reset && blockIdx.x == 0 && threadIdx.x == 0) *reset = 0;
  if (k_from) k = k_from[0];  // the barrier parameter the device selected (cxk_select_mu_async)
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = k * (b[p] * bs + AQc[p] * cs) - 2 * AW[p];
}

// y = cb b + cq AQc + cw AW : every right-hand side of the IPM loop (cone_program.cc:181, 409-411, 504)
__global__ void build_rhs_comb(int N, double cb, double cq, double cw, const double* __restrict__ b,
                               const double* __restrict__ AQc, const double* __restrict__ AW,
                               double* __restrict__ y, int* __restrict__ reset = nullptr) {
  if (reset && blockIdx.x == 0 && threadIdx.x == 0) *reset = 0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = cb * b[p] + cq * AQc[p] + cw * AW[p];
}

// The scalars the host loop needs per iteration (cone_program.cc:343-357, 439-446):
// out = { b.y, AQc.y, |b|^2, |AQc|^2, <w,c>, <c,Qc> } ; one workgroup, fixed summation order.
__global__ void __launch_bounds__(1024)
step_scalars(int N, const double* __restrict__ b, const double* __restrict__ AQc,
             const double* __restrict__ y, const double* __restrict__ sys_sc,
             double* __restrict__ out) {
  __shared__ double red[4][16];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  // sixteen strided elements per trip, all loads issued before the first fma: a plain loop is one
  // dependent memory round trip per element (15 in a row at C4; with 16 slots C4 is ONE trip).
  // Same fma order: same bits (out-of-range slots contribute fma(0, 0, s) = s).
  constexpr int U = 16;
  const double sc0 = sys_sc[0], sc1 = sys_sc[1];
  for (int p0 = threadIdx.x; p0 < N; p0 += U * blockDim.x) {
    double vb[U], vq[U], vy[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int p = p0 + u * blockDim.x;
      const bool on = p < N;
      vb[u] = on ? b[p] : 0.0;
      vq[u] = on ? AQc[p] : 0.0;
      vy[u] = on ? y[p] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      s0 = fma(vb[u], vy[u], s0);
      s1 = fma(vq[u], vy[u], s1);
      s2 = fma(vb[u], vb[u], s2);
      s3 = fma(vq[u], vq[u], s3);
    }
  }
  // four BlockSums (wave sum, then the wave totals added in wave order) behind ONE barrier pair
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  s0 = WaveSum(s0);
  s1 = WaveSum(s1);
  s2 = WaveSum(s2);
  s3 = WaveSum(s3);
  if (lane == 0) {
    red[0][wave] = s0;
    red[1][wave] = s1;
    red[2][wave] = s2;
    red[3][wave] = s3;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int w = 0; w < nw; w++) t += red[threadIdx.x][w];
    out[threadIdx.x] = t;
  }
  if (threadIdx.x == 4) out[4] = sc0;
  if (threadIdx.x == 5) out[5] = sc1;
}

// y = AQc cs - b bs  (ComputeMuFromDivergence cone_program.cc:181)
__global__ void build_mu_rhs(int N, double bs, double cs, const double* __restrict__ b,
                             const double* __restrict__ AQc, double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = AQc[p] * cs - b[p] * bs;
}

#endif  // CXK_DEVICE_FUNCTIONS_ONLY

// In-kernel stamps (diagnostic builds only: -DCXK_DEBUG_STAMPS); values go to a buffer nothing else reads.
#ifdef CXK_DEBUG_STAMPS
// g_cxk_want (set by cxk_debug_select): 0 = single-level factor launches, 1 = merged level ranges
// (grid > 1), 2 = the one-workgroup top.  Wave 0 of workgroup 0 records stamp i of level l of the
// launch at g_cxk_stamp[8 l + i]; the backward stamps go to [64 + i].
__device__ long long g_cxk_stamp[96];
__device__ int g_cxk_sel, g_cxk_want, g_cxk_lvl;
#define CXK_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_cxk_sel == 1) g_cxk_stamp[8 * g_cxk_lvl + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define CXK_STAMPB(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g_cxk_sel == 2) g_cxk_stamp[64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define CXK_STAMP_SELECT(lb, mode) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_cxk_lvl = 0; const int kind = (lb) == 0 ? 0 : (gridDim.x > 1 ? 1 : 2); g_cxk_sel = (kind == g_cxk_want) ? ((mode) == 0 ? 1 : ((mode) == 2 ? 2 : 0)) : 0; } } while (0)
#define CXK_STAMP_LEVEL(l) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_cxk_lvl = (l); } while (0)
#else
#ifdef CXK_CHAIN_STAMPS  // only the chain kernel's register-held stamps (tree_chain_lean)
__device__ long long g_cxk_stamp[96];
#endif
#define CXK_STAMP(i) do { } while (0)
#define CXK_STAMPB(i) do { } while (0)
#define CXK_STAMP_SELECT(lb, mode) do { } while (0)
#define CXK_STAMP_LEVEL(l) do { } while (0)
#endif

// Everything a wavefront needs to start on one supernode: one 128-byte record per position of
// the level lists (level order), fetched with a single coalesced load.
struct SnRec {
  int p, ns, nsep, start;
  int tg_beg, tg_end;  // pull targets (tg_loc / tr_ptr range)
  int bs_beg, bs_end;  // backward separator list (bs_c / bs_row range)
  int64_t diag_off, offd_off, upd_off;
  int updb_off;
  int m;               // slots per pull target:   upd[ubase + t_local * m + i]
  int64_t ubase;
  int fbase, mf;       // forward-solve slots:     updb[fbase + row * mf + i]
  int nsep_inline;     // > 0: sep[q] = row | column << 26 replaces the bs_* lists
  int pad_[3];
  int sep[8];
};
static_assert(sizeof(SnRec) == 128, "SnRec is read as 32 lanes x 4 bytes");

__device__ __forceinline__ int LoadRecWord(const SnRec* __restrict__ rec, int pos) {
  return reinterpret_cast<const int*>(rec + pos)[threadIdx.x & 31];
}
__device__ __forceinline__ SnRec DecodeRec(int w) {
  auto f = [&](int i) { return __builtin_amdgcn_readlane(w, i); };
  auto f64 = [&](int i) { return ((int64_t)f(i + 1) << 32) | (uint32_t)f(i); };
  SnRec R;
  R.p = f(0);
  R.ns = f(1);
  R.nsep = f(2);
  R.start = f(3);
  R.tg_beg = f(4);
  R.tg_end = f(5);
  R.bs_beg = f(6);
  R.bs_end = f(7);
  R.diag_off = f64(8);
  R.offd_off = f64(10);
  R.upd_off = f64(12);
  R.updb_off = f(14);
  R.m = f(15);
  R.ubase = f64(16);
  R.fbase = f(18);
  R.mf = f(19);
  R.nsep_inline = f(20);
#pragma unroll
  for (int q = 0; q < 8; q++) R.sep[q] = f(24 + q);
  return R;
}
__device__ __forceinline__ SnRec LoadRec(const SnRec* __restrict__ rec, int pos) { return DecodeRec(LoadRecWord(rec, pos)); }

// The root of the tree (no separator) solved backward straight from the registers of its upward
// step: the rows of L go through an LDS image with an odd stride (my[65 j + row]) and come back
// as columns; arithmetic and order are BackwardSupernodeLean's, so are the bits.  `y` is the
// forward-solved right-hand side of lane's row.  Needs 65 NSMAX doubles at `my`.
template <int NSMAX, bool DIAG_IN_ROWS, int LEN>
__device__ __forceinline__ double RootBackward(const double (&a)[LEN], double dg, double y, int ns, double* __restrict__ my) {
  const int lane = threadIdx.x & 63;
  const bool active = lane < ns;
#pragma unroll
  for (int j = 0; j < NSMAX; j++) my[65 * j + lane] = a[j];
  WaveSync();
  double col[NSMAX];
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = my[65 * (active ? lane : 0) + k];
  if constexpr (DIAG_IN_ROWS) dg = my[66 * (active ? lane : 0)];  // a[lane] of the own row
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = (active && k > lane && k < ns) ? col[k] : 0.0;
  dg = active ? dg : 1.0;
  double acc = active ? y : 0.0;
  const double dinv = 1.0 / dg;
#pragma unroll
  for (int k = NSMAX - 1; k >= 0; k--) {
    if (lane == k) acc *= dinv;
    acc = fma(-col[k], ReadLane(acc, k), acc);  // col[k] is zero for lanes >= k
  }
  return acc;
}

// A supernode without descendants whose panel is a permuted block of ONE constraint's Schur block
// (the leaves of a clique tree: BuildPlans checks every gather list): the first factor level
// loads it straight from the Schur kernels' output -- G(pos[r], pos[c]) -- together with its
// right-hand side, and the separate assembly launch disappears (the rest of the gather rides in
// the same launch as extra workgroups: tree_factor_level_asm).
struct AsmRec {
  int64_t g_off;          // the constraint's m x m block in G (column-major, lower triangle written)
  int64_t r_off;          // its entries of AWc / AQcc
  int m;
  int pad_;
  unsigned char pos[72];  // panel row q (the ns rows, then the separator rows) -> position in the constraint
};
static_assert(sizeof(AsmRec) == 96, "AsmRec is read as 24 lanes x 4 bytes");
struct AsmIn {
  const AsmRec* rec;  // [level-0 position]
  const double *G, *AWc, *AQcc, *b;
  double *AW, *AQc;
  double k, bs, cs;   // y = k (b bs + AQc cs) - 2 AW  (cone_program.cc:409-411)
  double cb, cq, cw;  // or (comb != 0) y = cb b + cq AQc + cw AW  (cone_program.cc:181, 504)
  int comb;
  int tag;            // a failed pivot writes fail[1] = tag (fail[0] is being reset by the gather beside it)
};

struct FactorPlan {
  const SnRec* rec;          // [level positions]
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
  const int* tg_reg;         // the same target in the register-shaped LDS image: 64 * column + lane
  const int* tr_ptr;         // [T+1] contributions of target t
  const int64_t* tr_src;     // index into `upd`
  const int* fs_ptr;         // [N+1] contributions of permuted row r
  const int* fs_src;         // index into `updb`
  // backward: separator columns in the reference's accumulation order
  const int* bs_ptr;         // [K+1]
  const int* bs_c;           // column index c within off block
  const int* bs_row;         // permuted index of separator variable
  double* upd;               // slot-ordered (see BuildPlans)
  double* updb;
  const int* pub_dst;        // [child-side numbering upd_off[p] + t] -> slot in upd
  const int* pubb_dst;       // [updb_off[p] + c] -> slot in updb
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
    const int* dst = P.pub_dst + P.upd_off[p];
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
      P.upd[dst[t]] = dot;
    }
  }
  if (with_rhs) {
    const int* dst = P.pubb_dst + P.updb_off[p];
    for (int c = lane; c < s; c += 64) {
      double dot = 0;
      for (int i = 0; i < ns; i++) dot = fma(sB[i + c * ns], sb[i], dot);
      P.updb[dst[c]] = dot;
    }
  }
}

// ---------------------------------------------------------------------------------------
// One wavefront factors one supernode, ROW PER LANE, everything in registers with static
// indices and no LDS traffic in the elimination loop:
//   lane r < ns            row r of the diagonal block          a[j] = L[r][j]
//   lane NSMAX + c, c < s  row of separator variable c          a[j] = off[j][c]
//   a[NSMAX + c]           the (initially zero) separator x separator trailing block: after the
//                          ns elimination steps it holds  -U[.,c] = -off[:, .] . off[:, c]
//   a[RB]                  right-hand side column: rows < ns end as the forward-solved b, the
//                          separator rows end as  -t[c] = -off[:,c] . b
// i.e. the Schur update and the forward-solve update this supernode publishes for its ancestors
// fall out of the same right-looking elimination (same fma chains as separate dot products).
// Step j: d = a[j] of lane j (v_readlane, static lane), L_jj = sqrt(d) and 1/L_jj from one
// v_rsq_f64 refined by two Goldschmidt iterations, column j scaled, then for every later column
// c:  a[c] -= L[c][j] * a[j]  with L[c][j] read from lane c.  Entries above the diagonal pick up
// garbage and are never read.  Padding pivots (ns <= j < NSMAX) are identity steps.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void SqrtAndInverse(double d, double& root, double& inv) {
  const double r0 = __builtin_amdgcn_rsq(d);
  double g = d * r0, h = 0.5 * r0;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double res = fma(-g, g, d);  // final correction: root is within 1 ulp of sqrt(d)
  root = fma(res, h, g);
  inv = h + h;
}

// a[c] += w[lane (c - BASE) of the own 16-lane DPP row] * v   for c in [C0, C1).
template <int LEN, int C0, int C1, int BASE>
struct DppColumns {
  static __device__ __forceinline__ void run(double (&a)[LEN], double w, double v) {
    if constexpr (C0 < C1) {
      asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
          : "+v"(a[C0])
          : "v"(w), "v"(v), "n"(C0 - BASE));
      DppColumns<LEN, C0 + 1, C1, BASE>::run(a, w, v);
    }
  }
};

// The compiler cannot see that the asm above is a DPP instruction, so the hazard "VALU writes a
// VGPR, a DPP instruction reads it within 2 wait states" is covered by hand: operands pass
// through this fence (s_nop 1) after their last write and before any DPP use.
__device__ __forceinline__ void DppOperandFence(double& x, double& y, double& z) {
  asm("s_nop 1" : "+v"(x), "+v"(y), "+v"(z));
}

// DPP rows 0 and 2 of v copied over rows 1 and 3 (v_permlane16_swap, gfx950).
__device__ __forceinline__ double EvenRowsToOddRows(double v) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const u2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b.x, a.x);
}

// Elimination steps J .. NSMAX-1 of FactorSupernodeRows (compile-time recursion: every register
// index, lane select and DPP control is an immediate).
// Step J receives sqrt(d_J) and 1/sqrt(d_J) from step J-1, which starts that dependent chain
// (readlane, v_rsq_f64, two Goldschmidt steps: ~130 cycles on a lone wavefront) as soon as column J
// has taken its own update, so that the chain overlaps the remaining column updates of step J-1
// instead of following them.  Same operations on the same values: results are unchanged.
// NRHS right-hand side columns a[RB ..] (1; 3 in the whole-tree launch with three right-hand sides).
// CHECK = false: no test of the pivots (two instructions per pivot on a lone wavefront's critical path): a pivot
// that is not positive leaves NaNs in its column and in everything eliminated behind it, and the caller looks
// for them in what it produces at its end (tree_fused: the solution entries).
template <int NSMAX, int SMAX, int J, int NRHS = 1, bool CHECK = true>
struct ElimSteps {
  static constexpr int LEN = NSMAX + SMAX + NRHS, RB = NSMAX + SMAX;
  // ns = columns of the supernode (wave-uniform): the padding pivots ns .. NSMAX-1 are identity
  // steps (unit diagonal, zero column) and are skipped.
  static __device__ __forceinline__ void run(double (&a)[LEN], int lane, bool& bad, int ns) {
    if constexpr (J == 0 && NSMAX > 0) {
      const double d = ReadLane(a[0], 0);
      if constexpr (CHECK) bad |= !(d > 0.0);
      double root, inv;
      SqrtAndInverse(d, root, inv);
      step(a, lane, bad, root, inv, ns);
    }
  }
  static __device__ __forceinline__ void step(double (&a)[LEN], int lane, bool& bad, double root, double inv, int ns) {
    if constexpr (J < NSMAX) {
      if (J >= ns) return;
      a[J] = (lane == J) ? root : a[J] * inv;
      double root1 = 1.0, inv1 = 1.0;
      auto next_pivot = [&]() {  // column J+1 is final for step J+1 once it has taken column J's term
        if constexpr (J + 1 < NSMAX) {
          const double d1 = ReadLane(a[J + 1], J + 1);
          if constexpr (CHECK) bad |= !(d1 > 0.0);
          SqrtAndInverse(d1, root1, inv1);
        }
      };
      if constexpr (NSMAX + SMAX <= 16) {
        // the whole panel (supernode rows + separator rows) sits in ONE 16-lane DPP row:
        // row_newbcast:c delivers L[c][J] (c < NSMAX) and L[sep c - NSMAX][J] directly
        double naj = -a[J];
        double dummy = 0.0;
        DppOperandFence(dummy, naj, a[J]);
        DppColumns<LEN, J + 1, (J + 2 < NSMAX + SMAX ? J + 2 : NSMAX + SMAX), 0>::run(a, a[J], naj);
        next_pivot();
        DppColumns<LEN, J + 2, NSMAX + SMAX, 0>::run(a, a[J], naj);
      } else if constexpr (NSMAX + SMAX <= 32 && NSMAX != 16) {
        // the panel (supernode rows, then separator rows at lanes NSMAX..) fills DPP rows 0 and 1.
        // Row 0 mirrored into row 1 serves the columns whose owner lane is < 16, row 1 mirrored
        // into row 0 the columns whose owner lane is >= 16.
        const RowPair xp = Swap16(a[J]);
        double x0 = xp.a, x1 = xp.b;
        double naj = -a[J];
        DppOperandFence(x0, x1, naj);
        constexpr int kEnd = NSMAX + SMAX;
        // column J + 1 first, then the chain of the next pivot, then the rest
        constexpr int n1 = (J + 2 < kEnd) ? J + 2 : kEnd;
        if constexpr (J + 1 < 16)
          DppColumns<LEN, J + 1, (n1 < 16 ? n1 : 16), 0>::run(a, x0, naj);
        else
          DppColumns<LEN, J + 1, n1, 16>::run(a, x1, naj);
        next_pivot();
        constexpr int kLo0 = (J + 2 < 16) ? J + 2 : 16, kLo1 = (kEnd < 16) ? kEnd : 16;
        constexpr int kHi0 = (J + 2 > 16) ? J + 2 : 16;
        DppColumns<LEN, kLo0, kLo1, 0>::run(a, x0, naj);
        DppColumns<LEN, kHi0, kEnd, 16>::run(a, x1, naj);
      } else if constexpr (NSMAX == 16 && SMAX <= 16) {
        // supernode rows fill DPP row 0, separator rows start DPP row 1.  L[c][J] (c < 16) is
        // lane c of row 0: with row 0 mirrored into row 1 a row_newbcast DPP operand delivers it
        // to both rows; L[sep c][J] is lane c of row 1, only row 1 needs the trailing block.
        double x = EvenRowsToOddRows(a[J]);
        double naj = -a[J];
        DppOperandFence(x, naj, a[J]);
        DppColumns<LEN, J + 1, (J + 2 < NSMAX ? J + 2 : NSMAX), 0>::run(a, x, naj);
        next_pivot();
        DppColumns<LEN, J + 2, NSMAX, 0>::run(a, x, naj);
        DppColumns<LEN, NSMAX, NSMAX + SMAX, NSMAX>::run(a, a[J], naj);
      } else {
        if constexpr (J + 1 < NSMAX + SMAX) a[J + 1] = fma(-ReadLane(a[J], J + 1), a[J], a[J + 1]);
        next_pivot();
#pragma unroll
        for (int c = J + 2; c < NSMAX + SMAX; c++) a[c] = fma(-ReadLane(a[J], c), a[J], a[c]);
      }
#pragma unroll
      for (int q = 0; q < NRHS; q++) {
        const double yj = ReadLane(a[RB + q], J) * inv;
        if (lane > J)
          a[RB + q] = fma(-yj, a[J], a[RB + q]);
        else if (lane == J)
          a[RB + q] = yj;
      }
      ElimSteps<NSMAX, SMAX, J + 1, NRHS, CHECK>::step(a, lane, bad, root1, inv1, ns);
    }
  }
};

template <int NSMAX, int SMAX>
__device__ inline void FactorSupernodeRows(const FactorPlan& P, const SnRec& R,
                                           double* __restrict__ slab, double* __restrict__ rhs,
                                           int* __restrict__ fail, double* __restrict__ my) {
  static_assert(NSMAX + SMAX <= 64, "one lane per panel row");
  constexpr int RB = NSMAX + SMAX;
  const int lane = threadIdx.x & 63;
  const int ns = R.ns, s = R.nsep;
  const bool is_row = lane < ns;
  const int sc = lane - NSMAX;
  const bool is_sep = sc >= 0 && sc < s;
  // panel element (lane, j) lives at base[o0 + j * st]:  D[r + j*ns]  or  B[j + c*ns]
  double* base = slab + R.diag_off;
  const unsigned rel = (unsigned)(R.offd_off - R.diag_off);
  const unsigned o0 = is_row ? (unsigned)lane : (is_sep ? rel + (unsigned)(sc * ns) : 0u);
  const unsigned st = is_row ? (unsigned)ns : 1u;
  const int lim = is_row ? lane + 1 : (is_sep ? ns : 0);  // valid j < lim
  CXK_STAMP(0);
  // ---- one round trip: panel, right-hand side, publish destinations and every value this
  // supernode pulls (dense slots: addresses depend on the record only)
  constexpr int TU = 2, MMAX = 8, MFMAX = 8;
  const int ntg = R.tg_end - R.tg_beg;
  const bool fast_pull = ntg <= 64 * TU && R.m <= MMAX && R.mf <= MFMAX;
  double a[NSMAX + SMAX + 1];
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = (j < lim) ? base[o0 + j * st] : 0.0;
  a[RB] = (rhs && is_row) ? rhs[R.start + lane] : 0.0;
#pragma unroll
  for (int c = 0; c < SMAX; c++) a[NSMAX + c] = 0.0;
  // The remaining loads are guarded by wave-uniform branches and use clamped (always valid)
  // addresses instead of per-lane predicates: a leaf skips them at the cost of a scalar branch.
  int pdst[SMAX > 0 ? SMAX : 1], pdstb = 0;
#pragma unroll
  for (int c = 0; c < SMAX; c++) pdst[c] = 0;
  if (s > 0) {
    const int k = is_sep ? sc : 0;
    const int* dst = P.pub_dst + R.upd_off + (k * s - k * (k - 1) / 2 - k);
#pragma unroll
    for (int c = 0; c < SMAX; c++) {
      const int cc = c < k ? k : (c < s ? c : s - 1);
      pdst[c] = dst[cc];
    }
    if (rhs) pdstb = P.pubb_dst[R.updb_off + k];
  }
  double pv[TU][MMAX], pb[MFMAX];
  int ploc[TU];
#pragma unroll
  for (int u = 0; u < TU; u++) {
    ploc[u] = -1;
#pragma unroll
    for (int i = 0; i < MMAX; i++) pv[u][i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < MFMAX; i++) pb[i] = 0.0;
  if (fast_pull && ntg > 0) {
#pragma unroll
    for (int u = 0; u < TU; u++)
      if (64 * u < ntg) {
        const int t = lane + 64 * u;
        const int ts = t < ntg ? t : 0;
        const int loc = P.tg_loc[R.tg_beg + ts];
        ploc[u] = t < ntg ? loc : -1;
        const double* src = P.upd + R.ubase + (int64_t)ts * R.m;
#pragma unroll
        for (int i = 0; i < MMAX; i++)
          if (i < R.m) pv[u][i] = src[i];
      }
  }
  if (fast_pull && rhs && R.mf > 0) {
    const double* src = P.updb + R.fbase + (is_row ? lane : 0) * R.mf;
#pragma unroll
    for (int i = 0; i < MFMAX; i++)
      if (i < R.mf) {
        const double v = src[i];
        pb[i] = is_row ? v : 0.0;
      }
  }
  CXK_STAMP(1);
  if (ntg > 0) {
    // descendants published Schur updates: apply them in the reference's order through an LDS
    // copy laid out like the slab ([diag ns x ns | off ns x s]); tg_loc indexes that copy
    const unsigned l0 = is_row ? (unsigned)lane : (unsigned)(ns * ns + (is_sep ? sc : 0) * ns);
#pragma unroll
    for (int j = 0; j < NSMAX; j++)
      if ((is_row || is_sep) && j < ns) my[l0 + j * st] = a[j];
    WaveSync();
    if (fast_pull) {
#pragma unroll
      for (int u = 0; u < TU; u++)
        if (ploc[u] >= 0) {
          double acc = my[ploc[u]];
#pragma unroll
          for (int i = 0; i < MMAX; i++) acc -= pv[u][i];  // unused slots hold 0.0: exact no-ops
          my[ploc[u]] = acc;
        }
    } else {
      for (int t = R.tg_beg + lane; t < R.tg_end; t += 64) {
        const int loc = P.tg_loc[t];
        double acc = my[loc];
        const int q1 = P.tr_ptr[t + 1];
#pragma unroll 4
        for (int q = P.tr_ptr[t]; q < q1; q++) acc -= P.upd[P.tr_src[q]];
        my[loc] = acc;
      }
    }
    WaveSync();
#pragma unroll
    for (int j = 0; j < NSMAX; j++)
      if (j < lim) a[j] = my[l0 + j * st];
  }
  if (fast_pull) {
#pragma unroll
    for (int i = 0; i < MFMAX; i++) a[RB] -= pb[i];
  } else if (rhs && is_row) {
    double acc = a[RB];
    const int q1 = P.fs_ptr[R.start + lane + 1];
#pragma unroll 4
    for (int q = P.fs_ptr[R.start + lane]; q < q1; q++) acc -= P.updb[P.fs_src[q]];
    a[RB] = acc;
  }
  // padding pivots: unit diagonal
#pragma unroll
  for (int j = 0; j < NSMAX; j++)
    if (j >= ns && lane == j) a[j] = 1.0;
  CXK_STAMP(2);
  bool bad = false;
  ElimSteps<NSMAX, SMAX, 0>::run(a, lane, bad, ns);
  CXK_STAMP(3);
  if (bad) {
    if (lane == 0) atomicExch(fail, 1);
    return;
  }
#pragma unroll
  for (int j = 0; j < NSMAX; j++)
    if (j < lim) base[o0 + j * st] = a[j];
  if (rhs && is_row) rhs[R.start + lane] = a[RB];
  CXK_STAMP(4);
  if (is_sep) {
    // U[k][c], k <= c (child-side numbering t = k*s - k(k-1)/2 + (c - k), the reference's S_S
    // enumeration), written straight into the consumer's slot
#pragma unroll
    for (int c = 0; c < SMAX; c++)
      if (c >= sc && c < s) P.upd[pdst[c]] = -a[NSMAX + c];
    if (rhs) P.updb[pdstb] = -a[RB];
  }
  CXK_STAMP(5);
}

// The same step with a STRAIGHT-LINE load phase, for supernodes whose pulls fit the dense slots
// (FastPull): a lone wavefront pays a full memory round trip (~1.2 us, nothing else to switch to)
// for every wait it meets, and the compiler waits for ALL outstanding loads wherever a loaded
// value is consumed inside or behind a branch that itself holds loads.  FactorSupernodeRows'
// branch ladders (per-lane predicates, "if (i < m)") cost it three round trips in a row: panel,
// pulled Schur values, pulled forward values.  Here every load -- panel, right-hand side, publish
// destinations, pull locations, pulled values -- is unconditional with a clamped, always valid
// address (the host pads the tables, kPullPad), nothing is consumed before the last one is
// issued, and unused values are masked afterwards: ONE round trip.  The pull is applied through
// an LDS image laid out like the registers (my[64 j + lane], P.tg_reg), written and read back
// without predicates.  Arithmetic and its order are those of FactorSupernodeRows (masked slots
// subtract 0.0: exact), so both give the same bits.
// Register shape (NSMAX << 8 | SMAX) the factor kernels pick for a supernode of ns columns and s
// separator rows (the dispatch of tree_sweep); 0 = no register kernel.
__host__ __device__ inline int RegisterShape(int ns, int s) {
  if (ns <= 8 && s <= 8) return 8 << 8 | 8;
  if (ns <= 16 && s <= 8) return 16 << 8 | 8;
  if (ns <= 24 && s == 0) return 24 << 8 | 0;
  if (ns <= 24 && s <= 8) return 24 << 8 | 8;
  if (ns <= 32 && s <= 16) return 32 << 8 | 16;
  return 0;
}

constexpr int kPullPad = 64;       // spare elements behind pub_dst / pubb_dst / tg_reg / upd / updb
constexpr int kFastTargets = 128;  // pull targets per supernode (2 per lane)
constexpr int kFastSlots = 8;      // contributions per target / forward contributions per row

__device__ __forceinline__ bool FastPull(const SnRec& R) {
  return R.tg_end - R.tg_beg <= kFastTargets && R.m <= kFastSlots && R.mf <= kFastSlots;
}

template <int NSMAX, int SMAX, bool RHS, bool ASM = false, bool ROOTBACK = false>
__device__ __forceinline__ void FactorSupernodeLean(const FactorPlan& P, const SnRec& R,
                                                    double* __restrict__ slab, double* __restrict__ rhs,
                                                    int* __restrict__ fail, double* __restrict__ my,
                                                    const AsmIn* ai = nullptr, int aw2 = 0) {
  static_assert(NSMAX + SMAX <= 64, "one lane per panel row");
  constexpr int RB = NSMAX + SMAX, MMAX = kFastSlots, MFMAX = kFastSlots;
  const int lane = threadIdx.x & 63;
  const int ns = R.ns, s = R.nsep;
  const bool is_row = lane < ns;
  const int sc = lane - NSMAX;
  const bool is_sep = sc >= 0 && sc < s;
  double* base = slab + R.diag_off;
  const unsigned rel = (unsigned)(R.offd_off - R.diag_off);
  const unsigned o0 = is_row ? (unsigned)lane : (is_sep ? rel + (unsigned)(sc * ns) : 0u);
  const unsigned st = is_row ? (unsigned)ns : 1u;
  const int lim = is_row ? lane + 1 : (is_sep ? ns : 0);  // valid j < lim
  CXK_STAMP(0);
  // ---- load phase: no consumer before the last load
  double a[NSMAX + SMAX + 1];
  double rb = 0.0, awv = 0.0, aqv = 0.0;
  if constexpr (ASM) {
    // aw2 = this lane's word of the AsmRec (loaded beside the SnRec): block offsets, then the
    // positions, one byte per panel row
    auto f = [&](int i) { return __builtin_amdgcn_readlane(aw2, i); };
    const double* Gk = ai->G + (((int64_t)f(1) << 32) | (uint32_t)f(0));
    const int64_t roff = ((int64_t)f(3) << 32) | (uint32_t)f(2);
    const int M = f(4);
    const int q = is_row ? lane : (is_sep ? ns + sc : 0);
    const int myp = (__builtin_amdgcn_ds_bpermute(4 * (6 + (q >> 2)), aw2) >> (8 * (q & 3))) & 255;
#pragma unroll
    for (int j = 0; j < NSMAX; j++) {
      const int pj = (f(6 + (j >> 2)) >> (8 * (j & 3))) & 255;  // wave-uniform
      const int hi = myp > pj ? myp : pj, lo = myp > pj ? pj : myp;
      a[j] = Gk[(j < lim) ? hi + lo * M : 0];
    }
    const int pr = is_row ? myp : (f(6) & 255);
    awv = ai->AWc[roff + pr];
    aqv = ai->AQcc[roff + pr];
    if constexpr (RHS) rb = ai->b[R.start + (is_row ? lane : 0)];
  } else {
#pragma unroll
    for (int j = 0; j < NSMAX; j++) a[j] = base[(j < lim) ? o0 + j * st : 0u];
    if constexpr (RHS) rb = rhs[R.start + (is_row ? lane : 0)];
  }
  int pdst[SMAX > 0 ? SMAX : 1], pdstb = 0;
  pdst[0] = 0;
  {
    const int k = is_sep ? sc : 0;
    const int* dst = P.pub_dst + R.upd_off + (k * s - k * (k - 1) / 2 - k);
#pragma unroll
    for (int c = 0; c < SMAX; c++) {
      int cc = c < k ? k : (c < s ? c : s - 1);
      cc = cc < 0 ? 0 : cc;
      pdst[c] = dst[cc];
    }
    if constexpr (RHS && SMAX > 0) pdstb = P.pubb_dst[R.updb_off + k];
  }
  const int ntg = R.tg_end - R.tg_beg;
  const int mlast = R.m > 0 ? R.m - 1 : 0, mflast = R.mf > 0 ? R.mf - 1 : 0;
  double pv0[MMAX], pv1[MMAX], pb[MFMAX];
  int ploc0, ploc1 = 0;
  {
    const int ts = lane < ntg ? lane : 0;
    ploc0 = P.tg_reg[R.tg_beg + ts];
    const double* src = P.upd + R.ubase + (int64_t)ts * R.m;
#pragma unroll
    for (int i = 0; i < MMAX; i++) pv0[i] = src[i < R.m ? i : mlast];
  }
#pragma unroll
  for (int i = 0; i < MMAX; i++) pv1[i] = 0.0;
  if (ntg > 64) {  // wave-uniform; loads only
    const int ts = lane + 64 < ntg ? lane + 64 : 0;
    ploc1 = P.tg_reg[R.tg_beg + ts];
    const double* src = P.upd + R.ubase + (int64_t)ts * R.m;
#pragma unroll
    for (int i = 0; i < MMAX; i++) pv1[i] = src[i < R.m ? i : mlast];
  }
  if constexpr (RHS) {
    const double* src = P.updb + R.fbase + (is_row ? lane : 0) * R.mf;
#pragma unroll
    for (int i = 0; i < MFMAX; i++) pb[i] = src[i < R.mf ? i : mflast];
  }
  CXK_STAMP(1);
  // ---- consumers
  if constexpr (ASM) {
    // what assemble_gather would have produced: sums that start from +0.0 (a -0.0 source ends up
    // +0.0), AW / AQc of the own variables for the kernels that follow, and the right-hand side
#pragma unroll
    for (int j = 0; j < NSMAX; j++) a[j] = 0.0 + a[j];
    awv = 0.0 + awv;
    aqv = 0.0 + aqv;
    if (is_row) {
      ai->AW[R.start + lane] = awv;
      ai->AQc[R.start + lane] = aqv;
    }
    if constexpr (RHS) {
      // the expressions of build_rhs / build_rhs_comb, term for term
      if (ai->comb)
        rb = ai->cb * rb + ai->cq * aqv + ai->cw * awv;
      else
        rb = ai->k * (rb * ai->bs + aqv * ai->cs) - 2 * awv;
    }
  }
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = (j < lim) ? a[j] : 0.0;
#pragma unroll
  for (int c = 0; c < SMAX; c++) a[NSMAX + c] = 0.0;
  a[RB] = (RHS && is_row) ? rb : 0.0;
  if (ntg > 0) {
    // descendants published Schur updates: applied in the reference's order on the LDS image
#pragma unroll
    for (int j = 0; j < NSMAX; j++) my[64 * j + lane] = a[j];
    WaveSync();
    if (lane < ntg) {
      double acc = my[ploc0];
#pragma unroll
      for (int i = 0; i < MMAX; i++) acc -= (i < R.m) ? pv0[i] : 0.0;
      my[ploc0] = acc;
    }
    if (lane + 64 < ntg) {
      double acc = my[ploc1];
#pragma unroll
      for (int i = 0; i < MMAX; i++) acc -= (i < R.m) ? pv1[i] : 0.0;
      my[ploc1] = acc;
    }
    WaveSync();
#pragma unroll
    for (int j = 0; j < NSMAX; j++) a[j] = my[64 * j + lane];
  }
  if constexpr (RHS) {
#pragma unroll
    for (int i = 0; i < MFMAX; i++) a[RB] -= (is_row && i < R.mf) ? pb[i] : 0.0;
  }
  // padding pivots: unit diagonal
#pragma unroll
  for (int j = 0; j < NSMAX; j++)
    if (j >= ns && lane == j) a[j] = 1.0;
  CXK_STAMP(2);
  bool bad = false;
  ElimSteps<NSMAX, SMAX, 0>::run(a, lane, bad, ns);
  CXK_STAMP(3);
  if (bad) {
    if (lane == 0) {
      if constexpr (ASM)
        atomicExch(fail + 1, ai->tag);
      else
        atomicExch(fail, 1);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NSMAX; j++)
    if (j < lim) base[o0 + j * st] = a[j];
  if constexpr (ROOTBACK) {
    // the chain's last step: the root is solved backward from these registers (tree_chain_lean)
    static_assert(RHS, "the root is solved backward only with a right-hand side");
    const double yv = RootBackward<NSMAX, true>(a, 0.0, a[RB], ns, my);
    if (is_row) rhs[R.start + lane] = yv;
    return;
  }
  if (RHS && is_row) rhs[R.start + lane] = a[RB];
  CXK_STAMP(4);
  if (is_sep) {
#pragma unroll
    for (int c = 0; c < SMAX; c++)
      if (c >= sc && c < s) P.upd[pdst[c]] = -a[NSMAX + c];
    if constexpr (RHS) P.updb[pdstb] = -a[RB];
  }
  CXK_STAMP(5);
}

// b_j <- L_j^{-T} (b_j - sum_c off_j[:,c] y[sep_j[c]]): lane i owns y_i and column i of L
// (col[k] = L[k][i], k > i) in registers; the solved entry travels by v_readlane.
template <int NSMAX, int SMAX>
__device__ inline void BackwardSupernodeRows(const FactorPlan& P, const SnRec& R,
                                             const double* __restrict__ slab,
                                             double* __restrict__ rhs) {
  const int lane = threadIdx.x & 63;
  const int ns = R.ns;
  const bool active = lane < ns;
  const double* D = slab + R.diag_off + (size_t)(active ? lane : 0) * ns;  // column `lane`
  const double* B = slab + R.offd_off + (active ? lane : 0);
  CXK_STAMPB(1);
  double col[NSMAX];
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = (active && k > lane && k < ns) ? D[k] : 0.0;
  const double dg = active ? D[lane] : 1.0;
  double acc = active ? rhs[R.start + lane] : 0.0;
  const int cnt = R.bs_end - R.bs_beg;
  if (R.nsep_inline == cnt) {
    // separator rows / columns come with the record: y[sep] and off[:, c] load in the same trip
    constexpr int QN = SMAX < 8 ? SMAX : 8;
    double bv[QN > 0 ? QN : 1], yv[QN > 0 ? QN : 1];
#pragma unroll
    for (int q = 0; q < QN; q++) {
      const bool on = q < cnt;
      const unsigned w = (unsigned)R.sep[q];
      yv[q] = on ? rhs[w & 0x3ffffffu] : 0.0;
      bv[q] = (on && active) ? B[(size_t)(w >> 26) * ns] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < QN; q++) acc -= bv[q] * yv[q];
  } else {
#pragma unroll 4
    for (int q = R.bs_beg; q < R.bs_end; q++) {
      const double yq = rhs[P.bs_row[q]];
      if (active) acc -= B[(size_t)P.bs_c[q] * ns] * yq;
    }
  }
  CXK_STAMPB(2);
  const double dinv = 1.0 / dg;
  CXK_STAMPB(3);
#pragma unroll
  for (int k = NSMAX - 1; k >= 0; k--) {
    if (lane == k) acc *= dinv;
    acc = fma(-col[k], ReadLane(acc, k), acc);  // col[k] is zero for lanes >= k
  }
  CXK_STAMPB(4);
  if (active) rhs[R.start + lane] = acc;
}

template <int NSMAX, int SMAX>
__device__ __forceinline__ void BackwardSupernodeLean(const SnRec& R, const double* __restrict__ slab,
                                                      double* __restrict__ rhs) {
  BackwardSupernodeLeanSync<NSMAX, SMAX, false>(R, slab, rhs, [] {});
}

// Where a solve-only sweep takes its right-hand side: form 0 from `rhs` (someone built it), form 1 /
// 2 each supernode forms its own rows on the fly -- the expressions of build_rhs / build_rhs_comb,
// term for term -- so that the separate launch that used to fill `rhs` first disappears.
struct RhsIn {
  int form;
  const double *b, *AQc, *AW;
  double k, bs, cs;    // form 1: k (b bs + AQc cs) - 2 AW   (cone_program.cc:409-411)
  double cb, cq, cw;   // form 2: cb b + cq AQc + cw AW      (cone_program.cc:181, 504)
  const double* k_from;  // form 1, not null: k = k_from[0], the barrier parameter the device selected
};
__device__ __forceinline__ double RhsValue(const RhsIn& ri, const double* __restrict__ rhs, int p) {
  if (ri.form == 0) return rhs[p];
  const double bp = ri.b[p], aq = ri.AQc[p], aw = ri.AW[p];
  const double kk = (ri.form == 1 && ri.k_from) ? ri.k_from[0] : ri.k;
  return ri.form == 1 ? kk * (bp * ri.bs + aq * ri.cs) - 2 * aw : ri.cb * bp + ri.cq * aq + ri.cw * aw;
}

// ForwardSupernodeWave (b_j <- L_j^{-1} (b_j - pulled forward updates), publish t[c] = off[:,c].b)
// in the row-per-lane register layout with a straight-line load phase: lane r < ns holds row r of
// L, lane NSMAX + c holds column c of the off block.  Operations and their order are those of the
// generic kernel (reciprocal of the diagonal, multiply-then-subtract substitution, fma chain over
// the rows for t[c]), so the results are the same bits.  Needs dense forward slots (R.mf <= 8).
template <int NSMAX, int SMAX, bool ROOTBACK = false>
__device__ __forceinline__ void ForwardSupernodeLean(const FactorPlan& P, const SnRec& R,
                                                     const double* __restrict__ slab,
                                                     double* __restrict__ rhs, const RhsIn& ri,
                                                     double* __restrict__ my = nullptr) {
  constexpr int MFMAX = kFastSlots;
  const int lane = threadIdx.x & 63;
  const int ns = R.ns, s = R.nsep;
  const bool is_row = lane < ns;
  const int sc = lane - NSMAX;
  const bool is_sep = sc >= 0 && sc < s;
  const double* base = slab + R.diag_off;
  const unsigned rel = (unsigned)(R.offd_off - R.diag_off);
  const unsigned o0 = is_row ? (unsigned)lane : (is_sep ? rel + (unsigned)(sc * ns) : 0u);
  const unsigned st = is_row ? (unsigned)ns : 1u;
  const int lim = is_row ? lane : (is_sep ? ns : 0);  // strictly lower part of a row; a whole off column
  // ---- load phase: no consumer before the last load
  double a[NSMAX > 0 ? NSMAX : 1];
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = base[(j < lim) ? o0 + j * st : 0u];
  double dg = base[is_row ? (unsigned)lane * (unsigned)(ns + 1) : 0u];
  double b = RhsValue(ri, rhs, R.start + (is_row ? lane : 0));
  int pdstb = 0;
  if constexpr (SMAX > 0) pdstb = P.pubb_dst[R.updb_off + (is_sep ? sc : 0)];
  const int mflast = R.mf > 0 ? R.mf - 1 : 0;
  double pb[MFMAX];
  {
    const double* src = P.updb + R.fbase + (is_row ? lane : 0) * R.mf;
#pragma unroll
    for (int i = 0; i < MFMAX; i++) pb[i] = src[i < R.mf ? i : mflast];
  }
  // ---- consumers
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = (j < lim) ? a[j] : 0.0;
  b = is_row ? b : 0.0;
#pragma unroll
  for (int i = 0; i < MFMAX; i++) b -= (is_row && i < R.mf) ? pb[i] : 0.0;
  const double dinv = is_row ? 1.0 / dg : 0.0;
  double dot = 0.0;
#pragma unroll
  for (int k = 0; k < NSMAX; k++) {
    if (lane == k) b *= dinv;
    const double bk = ReadLane(b, k);  // 0.0 for padding rows k >= ns
    if (is_sep)
      dot = fma(a[k], bk, dot);
    else
      b -= a[k] * bk;  // a[k] is zero for lanes <= k
  }
  if constexpr (ROOTBACK) {  // the chain's last step (no separator): straight back down from these registers
    const double yv = RootBackward<NSMAX, false>(a, dg, b, ns, my);
    if (is_row) rhs[R.start + lane] = yv;
    return;
  }
  if (is_row) rhs[R.start + lane] = b;
  if constexpr (SMAX > 0)
    if (is_sep) P.updb[pdstb] = dot;
}

// BackwardSupernodeRows with a straight-line load phase (see FactorSupernodeLean): column of L,
// off-block entries and the separator values y[sep] all load unconditionally from clamped
// addresses, masks are applied afterwards.  Needs the inline separator list (R.nsep_inline == count).
// `sync` runs between the loads that depend on nothing this launch computes (the supernode's own
// panel and forward-solved values) and the loads of the separator's solution: tree_backward_pair
// passes the workgroup barrier behind which the parent's solution becomes visible.
// FRESH: the separator's solution may have been written by another wavefront of this launch.  Its
// addresses are wave-uniform, so the compiler fetches it with SCALAR loads, and the scalar cache
// is not coherent with vector stores (a line another workgroup pulled in before the parent wrote
// it stays stale): workgroup-scope atomic loads go through the vector path instead.
template <int NSMAX, int SMAX, bool FRESH, typename Sync>
__device__ __forceinline__ void BackwardSupernodeLeanSync(const SnRec& R, const double* __restrict__ slab,
                                                          double* __restrict__ rhs, Sync sync) {
  const int lane = threadIdx.x & 63;
  const int ns = R.ns;
  const bool active = lane < ns;
  const double* D = slab + R.diag_off + (size_t)(active ? lane : 0) * ns;  // column `lane`
  const double* B = slab + R.offd_off + (active ? lane : 0);
  CXK_STAMPB(1);
  double col[NSMAX];
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = D[(active && k > lane && k < ns) ? k : 0];
  double dg = D[active ? lane : 0];
  double acc = rhs[R.start + (active ? lane : 0)];
  const int cnt = R.bs_end - R.bs_beg;
  constexpr int QN = SMAX < 8 ? SMAX : 8;
  double bv[QN > 0 ? QN : 1], yv[QN > 0 ? QN : 1];
#pragma unroll
  for (int q = 0; q < QN; q++) {
    const unsigned w = q < cnt ? (unsigned)R.sep[q] : 0u;
    // unused slots read the diagonal block instead: a supernode without separator has no off
    // block, and the root's would start at the end of the slab
    const double* src = q < cnt ? B + (size_t)(w >> 26) * ns : D;
    bv[q] = src[0];
  }
  sync();
#pragma unroll
  for (int q = 0; q < QN; q++) {
    const unsigned w = q < cnt ? (unsigned)R.sep[q] : 0u;
    if constexpr (FRESH)
      yv[q] = __hip_atomic_load(rhs + (w & 0x3ffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      yv[q] = rhs[w & 0x3ffffffu];
  }
  // ---- consumers
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = (active && k > lane && k < ns) ? col[k] : 0.0;
  dg = active ? dg : 1.0;
  acc = active ? acc : 0.0;
#pragma unroll
  for (int q = 0; q < QN; q++) acc -= ((q < cnt && active) ? bv[q] : 0.0) * (q < cnt ? yv[q] : 0.0);
  CXK_STAMPB(2);
  const double dinv = 1.0 / dg;
  CXK_STAMPB(3);
#pragma unroll
  for (int k = NSMAX - 1; k >= 0; k--) {
    if (lane == k) acc *= dinv;
    acc = fma(-col[k], ReadLane(acc, k), acc);  // col[k] is zero for lanes >= k
  }
  CXK_STAMPB(4);
  if (active) rhs[R.start + lane] = acc;
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

// MODE 0: factor (+ forward if rhs), MODE 1: forward only, MODE 2: backward.
// TOP = false: one level per launch, positions [base0, base0 + cnt0) of the level-ordered
// records, one wavefront per supernode.
// TOP = true: a RANGE of `nl` consecutive levels per launch.  Workgroup g sweeps one connected
// piece of the elimination forest restricted to those levels (a subtree, or the whole top of
// the tree): its records are stored level by level, wg_lev[g * (nl + 1) + l] is the first
// position of its level l.  Levels inside the workgroup are separated by a workgroup barrier
// instead of a kernel boundary; MODE 0/1 walk them upwards and (then_backward) straight back
// down, MODE 2 walks them downwards.  The records of the piece are prefetched into LDS with one
// load at kernel start, so a level step pays one memory round trip (its data) instead of two.
// The kernel is specialised per (MODE, TOP) so that each instance keeps only the plan fields it
// uses in SGPRs.
constexpr int kRangeMaxRecs = 96;  // records of one workgroup's piece held in LDS (12 KB)

template <int MODE, bool TOP>
__global__ void __launch_bounds__(512)
tree_sweep(FactorPlan P, const SnRec* __restrict__ recs, const int* __restrict__ wg_lev, int base0,
           int cnt0, int nl, int then_backward, double* __restrict__ slab, double* __restrict__ rhs,
           int* __restrict__ fail, int lds_per_wave) {
  extern __shared__ double lds[];
  __shared__ int s_rec[TOP ? kRangeMaxRecs * 32 : 32];
  // wave-uniform values are forced into SGPRs: otherwise every loop bound / lane select below
  // is treated as divergent (waterfall loops around v_readlane, vector address arithmetic)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* my = lds + (size_t)wave * lds_per_wave;
  CXK_STAMP_SELECT(TOP ? 1 : 0, MODE);
  CXK_STAMP(6);
  CXK_STAMPB(0);
  const int* lp = nullptr;
  int first = base0;
  bool prefetched = false;
  if constexpr (TOP) {
    lp = wg_lev + (size_t)blockIdx.x * (nl + 1);
    first = lp[0];
    const int nrec = lp[nl] - first;
    prefetched = nrec <= kRangeMaxRecs;  // a long narrow top (chain-shaped trees) reads from HBM
    if (prefetched) {
      const int* src = reinterpret_cast<const int*>(recs + first);
      for (int q = threadIdx.x; q < nrec * 32; q += blockDim.x) s_rec[q] = src[q];
    }
    __syncthreads();
  } else {
    nl = 1;
    then_backward = 0;
  }
  auto load_rec = [&](int pos) -> SnRec {
    if (TOP && prefetched) return LoadRec(reinterpret_cast<const SnRec*>(s_rec), pos - first);
    return LoadRec(recs, pos);
  };
  if constexpr (MODE != 2) {
    for (int l = 0; l < nl; l++) {
      const int base = TOP ? lp[l] : base0;
      const int cnt = TOP ? lp[l + 1] - base : cnt0;
      CXK_STAMP_LEVEL(l);
      for (int idx = (TOP ? 0 : blockIdx.x * nw) + wave; idx < cnt; idx += (TOP ? 1 : gridDim.x) * nw) {
        const SnRec R = load_rec(base + idx);
        const int ns = R.ns, s = R.nsep;
        if constexpr (MODE == 0) {
          if (ns <= 8 && s <= 8)
            FactorSupernodeRows<8, 8>(P, R, slab, rhs, fail, my);
          else if (ns <= 16 && s <= 8)
            FactorSupernodeRows<16, 8>(P, R, slab, rhs, fail, my);
          else if (ns <= 24 && s == 0)
            FactorSupernodeRows<24, 0>(P, R, slab, rhs, fail, my);
          else if (ns <= 24 && s <= 8)
            FactorSupernodeRows<24, 8>(P, R, slab, rhs, fail, my);
          else if (ns <= 32 && s <= 16)
            FactorSupernodeRows<32, 16>(P, R, slab, rhs, fail, my);
          else
            CholSupernodeLds(P, R.p, slab, rhs, fail, my);
        } else {
          if (ns <= 64)
            ForwardSupernodeWave(P, R.p, slab, rhs, my);
          else
            ForwardSupernodeLds(P, R.p, slab, rhs, my);
        }
      }
      if constexpr (TOP) __syncthreads();
    }
  }
  if (MODE == 2 || (TOP && then_backward)) {
    for (int l = nl - 1; l >= 0; l--) {
      const int base = TOP ? lp[l] : base0;
      const int cnt = TOP ? lp[l + 1] - base : cnt0;
      for (int idx = (TOP ? 0 : blockIdx.x * nw) + wave; idx < cnt; idx += (TOP ? 1 : gridDim.x) * nw) {
        const SnRec R = load_rec(base + idx);
        const int ns = R.ns, s = R.nsep;
        if (ns <= 8 && s <= 8)
          BackwardSupernodeRows<8, 8>(P, R, slab, rhs);
        else if (ns <= 16 && s <= 8)
          BackwardSupernodeRows<16, 8>(P, R, slab, rhs);
        else if (ns <= 24 && s <= 8)
          BackwardSupernodeRows<24, 8>(P, R, slab, rhs);
        else if (ns <= 32 && s <= 16)
          BackwardSupernodeRows<32, 16>(P, R, slab, rhs);
        else if (ns <= 64)
          BackwardSupernodeWave(P, R.p, slab, rhs, my);
        else
          BackwardSupernodeLds(P, R.p, slab, rhs, my);
      }
      if constexpr (TOP) __syncthreads();
    }
  }
  CXK_STAMP(7);
  CXK_STAMPB(5);
}

// ---------------------------------------------------------------------------------------
// One factor level whose supernodes all fit ONE register shape (NSMAX, SMAX): the same step as
// tree_sweep<0, false>, compiled for that shape alone.  The generic kernel carries every shape's
// elimination plus the LDS fallback and pays for it in scalar-register spills on the path of each
// shape; a level of a regular clique tree (all of BASELINE config 4) takes this kernel instead.
// ---------------------------------------------------------------------------------------
template <int NSMAX, int SMAX, bool RHS>
__global__ void __launch_bounds__(256)
tree_factor_level(FactorPlan P, const SnRec* __restrict__ recs, int base0, int cnt0,
                  double* __restrict__ slab, double* __restrict__ rhs, int* __restrict__ fail,
                  int lds_per_wave) {
  extern __shared__ double lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* my = lds + (size_t)wave * lds_per_wave;
  const int idx = blockIdx.x * nw + wave;
  CXK_STAMP_SELECT(0, 0);
  CXK_STAMP(6);
  if (idx >= cnt0) return;
  const SnRec R = LoadRec(recs, base0 + idx);
  FactorSupernodeLean<NSMAX, SMAX, RHS>(P, R, slab, rhs, fail, my);
  CXK_STAMP(7);
}

// The first factor level with the assembly folded in: workgroups [0, fwgs) factor supernodes
// whose panels come straight from the Schur blocks (AsmRec), the others run the gather of
// everything else (slab entries of the levels above, their right-hand side, <w,c>, <c,Qc>) --
// needed by the NEXT level's launch only.
template <int NSMAX, int SMAX, bool RHS>
__global__ void __launch_bounds__(256)
tree_factor_level_asm(FactorPlan P, const SnRec* __restrict__ recs, int base0, int cnt0,
                      double* __restrict__ slab, double* __restrict__ rhs, int* __restrict__ fail,
                      int lds_per_wave, AsmIn ai, GatherArgs ga, int fwgs) {
  extern __shared__ double lds[];
  if ((int)blockIdx.x >= fwgs) {
    GatherBody(ga, blockIdx.x - fwgs, gridDim.x - fwgs);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* my = lds + (size_t)wave * lds_per_wave;
  const int idx = blockIdx.x * nw + wave;
  if (idx >= cnt0) return;
  const int lane = threadIdx.x & 63;
  const int aw2 = reinterpret_cast<const int*>(ai.rec + idx)[lane < 24 ? lane : 0];  // same trip as the record
  const SnRec R = LoadRec(recs, base0 + idx);
  FactorSupernodeLean<NSMAX, SMAX, RHS, true>(P, R, slab, rhs, fail, my, &ai, aw2);
}

// tree_factor_level_asm for a first level of TWO register shapes (programs mixing matrix cones with
// second-order cones: config 5): workgroups [0, blocksA) shape A, [blocksA, fwgs) shape B, the rest
// the gather.  AsmRec q belongs to level position q (segment B follows segment A).
template <int NA, int SA, int NB, int SB>
__global__ void __launch_bounds__(256)
tree_factor_level2_asm(FactorPlan P, const SnRec* __restrict__ recs, int baseA, int cntA, int blocksA,
                       int baseB, int cntB, double* __restrict__ slab, double* __restrict__ rhs,
                       int* __restrict__ fail, int lds_per_wave, AsmIn ai, GatherArgs ga, int fwgs) {
  extern __shared__ double lds[];
  if ((int)blockIdx.x >= fwgs) {
    GatherBody(ga, blockIdx.x - fwgs, gridDim.x - fwgs);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* my = lds + (size_t)wave * lds_per_wave;
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x < blocksA) {
    const int idx = blockIdx.x * nw + wave;
    if (idx >= cntA) return;
    const int aw2 = reinterpret_cast<const int*>(ai.rec + idx)[lane < 24 ? lane : 0];
    const SnRec R = LoadRec(recs, baseA + idx);
    FactorSupernodeLean<NA, SA, true, true>(P, R, slab, rhs, fail, my, &ai, aw2);
  } else {
    const int idx = (blockIdx.x - blocksA) * nw + wave;
    if (idx >= cntB) return;
    const int aw2 = reinterpret_cast<const int*>(ai.rec + cntA + idx)[lane < 24 ? lane : 0];
    const SnRec R = LoadRec(recs, baseB + idx);
    FactorSupernodeLean<NB, SB, true, true>(P, R, slab, rhs, fail, my, &ai, aw2);
  }
}

// The backward step of one level, same specialisation (no LDS).
template <int NSMAX, int SMAX>
__global__ void __launch_bounds__(256)
tree_backward_level(const SnRec* __restrict__ recs, int base0, int cnt0, const double* __restrict__ slab,
                    double* __restrict__ rhs) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int idx = blockIdx.x * nw + wave;
  if (idx >= cnt0) return;
  CXK_STAMP_SELECT(0, 2);
  CXK_STAMPB(0);
  const SnRec R = LoadRec(recs, base0 + idx);
  BackwardSupernodeLean<NSMAX, SMAX>(R, slab, rhs);
  CXK_STAMPB(5);
}

template <int NSMAX, int SMAX>
__global__ void __launch_bounds__(256)
tree_forward_level(FactorPlan P, const SnRec* __restrict__ recs, int base0, int cnt0,
                   const double* __restrict__ slab, double* __restrict__ rhs, RhsIn ri) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int idx = blockIdx.x * nw + wave;
  if (idx >= cnt0) return;
  const SnRec R = LoadRec(recs, base0 + idx);
  ForwardSupernodeLean<NSMAX, SMAX>(P, R, slab, rhs, ri);
}

template <int NA, int SA, int NB, int SB>
__global__ void __launch_bounds__(256)
tree_forward_level2(FactorPlan P, const SnRec* __restrict__ recs, int baseA, int cntA, int blocksA,
                    int baseB, int cntB, const double* __restrict__ slab, double* __restrict__ rhs, RhsIn ri) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  if ((int)blockIdx.x < blocksA) {
    const int idx = blockIdx.x * nw + wave;
    if (idx >= cntA) return;
    const SnRec R = LoadRec(recs, baseA + idx);
    ForwardSupernodeLean<NA, SA>(P, R, slab, rhs, ri);
  } else {
    const int idx = (blockIdx.x - blocksA) * nw + wave;
    if (idx >= cntB) return;
    const SnRec R = LoadRec(recs, baseB + idx);
    ForwardSupernodeLean<NB, SB>(P, R, slab, rhs, ri);
  }
}

// Two segments of one level in ONE launch (the supernodes of a level are independent): workgroups
// [0, blocksA) run shape A, the rest shape B.  Programs that mix small cones (second-order cones:
// shape <8,8>) with matrix cones have two shapes on every level; two launches would serialise.
template <int NA, int SA, int NB, int SB, bool RHS>
__global__ void __launch_bounds__(256)
tree_factor_level2(FactorPlan P, const SnRec* __restrict__ recs, int baseA, int cntA, int blocksA,
                   int baseB, int cntB, double* __restrict__ slab, double* __restrict__ rhs,
                   int* __restrict__ fail, int lds_per_wave) {
  extern __shared__ double lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* my = lds + (size_t)wave * lds_per_wave;
  if ((int)blockIdx.x < blocksA) {
    const int idx = blockIdx.x * nw + wave;
    if (idx >= cntA) return;
    const SnRec R = LoadRec(recs, baseA + idx);
    FactorSupernodeLean<NA, SA, RHS>(P, R, slab, rhs, fail, my);
  } else {
    const int idx = (blockIdx.x - blocksA) * nw + wave;
    if (idx >= cntB) return;
    const SnRec R = LoadRec(recs, baseB + idx);
    FactorSupernodeLean<NB, SB, RHS>(P, R, slab, rhs, fail, my);
  }
}

template <int NA, int SA, int NB, int SB>
__global__ void __launch_bounds__(256)
tree_backward_level2(const SnRec* __restrict__ recs, int baseA, int cntA, int blocksA, int baseB, int cntB,
                     const double* __restrict__ slab, double* __restrict__ rhs) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  if ((int)blockIdx.x < blocksA) {
    const int idx = blockIdx.x * nw + wave;
    if (idx >= cntA) return;
    const SnRec R = LoadRec(recs, baseA + idx);
    BackwardSupernodeLean<NA, SA>(R, slab, rhs);
  } else {
    const int idx = (blockIdx.x - blocksA) * nw + wave;
    if (idx >= cntB) return;
    const SnRec R = LoadRec(recs, baseB + idx);
    BackwardSupernodeLean<NB, SB>(R, slab, rhs);
  }
}

// Two consecutive levels of the way DOWN in one launch.  Workgroup g (nine wavefronts) owns one
// supernode of the upper level and the supernodes of the lower level that read its solution
// (BackPairEntry; the host orders a level so that they are consecutive): wavefront 0 solves the
// parent while wavefronts 1 .. 8 already fetch their children's panels, the workgroup barrier
// publishes the parent's solution, the children finish.  Lower-level supernodes that read nothing
// of the upper level ride in parentless workgroups.  Same device function as tree_backward_level
// on the same records: same bits, one launch and one cold start fewer per pair.
struct BackPairEntry {
  int parent;  // record position of the upper-level supernode, -1: none
  int first;   // record position of the first lower-level supernode of this workgroup
  int count;   // how many (consecutive)
  int pad;
};

template <int NP, int SP, int NC, int SC>
__global__ void __launch_bounds__(576)
tree_backward_pair(const SnRec* __restrict__ recs, const BackPairEntry* __restrict__ tab,
                   const double* __restrict__ slab, double* __restrict__ rhs) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const BackPairEntry e = tab[blockIdx.x];
  const bool with_parent = e.parent >= 0;
  if (wave == 0) {
    if (!with_parent) return;
    const SnRec R = LoadRec(recs, e.parent);
    BackwardSupernodeLean<NP, SP>(R, slab, rhs);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    return;
  }
  int i = wave - 1;
  if (i >= e.count) {
    if (with_parent) __syncthreads();  // every wavefront of the workgroup meets the one barrier
    return;
  }
  {
    const SnRec R = LoadRec(recs, e.first + i);
    BackwardSupernodeLeanSync<NC, SC, true>(R, slab, rhs, [&] {
      if (with_parent) {
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
    });
  }
  for (i += 8; i < e.count; i += 8) {
    const SnRec R = LoadRec(recs, e.first + i);
    BackwardSupernodeLeanSync<NC, SC, true>(R, slab, rhs, [] {});
  }
}

// The CHAIN at the top of the tree -- trailing levels of exactly one supernode each (the root and
// what leads to it) -- as one launch of ONE wavefront: steps up (MODE 0 factor + forward, MODE 1
// forward) and straight back down.  Consecutive steps are dependent anyway, so a launch per level
// buys nothing here; the wavefront passes its published values to itself through memory
// (workgroup-scope fence between steps: one CU, one L1).  Shapes A and B cover the chain's
// supernodes.  Dependent memory round trips are what a step costs (~1.2 us each), so: the
// records come from consecutive positions (no table look-up), each is fetched while the step
// before it runs and kept in LDS for the way down, and the root turns around in registers.
constexpr int kChainRing = 64;  // records of the chain's last steps kept in LDS for the way down

template <int MODE, int NA, int SA, int NB, int SB>
__global__ void __launch_bounds__(64)
tree_chain_lean(FactorPlan P, const SnRec* __restrict__ recs, int pos0, int n, double* __restrict__ slab,
                double* __restrict__ rhs, int* __restrict__ fail, RhsIn ri) {
  extern __shared__ double lds[];
  // One supernode per level: the chain's records are consecutive in level order (pos0 ..).  Each
  // record is in flight while the step before it runs; the last kChainRing of them stay in LDS for
  // the way back down, the ones below are fetched again, one step ahead as on the way up (a chain
  // may be thousands of levels long: BASELINE config 3 as the reference arranges it).
  constexpr int IMG = 65 * (NA > NB ? NA : NB);  // RootBackward's image (>= the pull image of 64 columns)
  int* rec_lds = reinterpret_cast<int*>(lds + IMG);
  const int ntop = n;  // upward steps
  const int lane = threadIdx.x & 63;
#if defined(CXK_DEBUG_STAMPS) || defined(CXK_CHAIN_STAMPS)
  long long tstamp[8];  // held in registers, written once at the end: no memory traffic in between
  int nstamp = 0;
  tstamp[nstamp++] = __builtin_amdgcn_s_memtime();
#endif
  int wnext = LoadRecWord(recs, pos0);
  for (int q = 0; q < n; q++) {
    const int w = wnext;
    if (q + 1 < n) wnext = LoadRecWord(recs, pos0 + q + 1);
    if (lane < 32) rec_lds[32 * (q & (kChainRing - 1)) + lane] = w;
    const SnRec R = DecodeRec(w);
    const int shape = RegisterShape(R.ns, R.nsep);
    const bool isA = shape == (NA << 8 | SA);
    // the last step is the root (no separator): solved backward from the registers of its
    // upward step (shape B; a root of another shape takes the steps through memory like the rest)
    const bool root = q + 1 == n && shape == (NB << 8 | SB) && R.nsep == 0;
    if constexpr (MODE == 0) {
      if (root)
        FactorSupernodeLean<NB, SB, true, false, true>(P, R, slab, rhs, fail, lds);
      else if (isA)
        FactorSupernodeLean<NA, SA, true>(P, R, slab, rhs, fail, lds);
      else
        FactorSupernodeLean<NB, SB, true>(P, R, slab, rhs, fail, lds);
    } else {
      if (root)
        ForwardSupernodeLean<NB, SB, true>(P, R, slab, rhs, ri, lds);
      else if (isA)
        ForwardSupernodeLean<NA, SA>(P, R, slab, rhs, ri);
      else
        ForwardSupernodeLean<NB, SB>(P, R, slab, rhs, ri);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#if defined(CXK_DEBUG_STAMPS) || defined(CXK_CHAIN_STAMPS)
    if (nstamp < 7) tstamp[nstamp++] = __builtin_amdgcn_s_memtime();
#endif
    if (root) n--;  // done with the root: the way down starts below it
  }
  WaveSync();
  const int first_cached = ntop - kChainRing;  // records q >= first_cached are in the ring
  int wdown = (n - 1 >= 0 && n - 1 < first_cached) ? LoadRecWord(recs, pos0 + n - 1) : 0;
  for (int q = n - 1; q >= 0; q--) {
    const int wq = q >= first_cached ? rec_lds[32 * (q & (kChainRing - 1)) + (lane & 31)] : wdown;
    if (q - 1 >= 0 && q - 1 < first_cached) wdown = LoadRecWord(recs, pos0 + q - 1);
    const SnRec R = DecodeRec(wq);
    const bool isA = RegisterShape(R.ns, R.nsep) == (NA << 8 | SA);
    if (isA)
      BackwardSupernodeLean<NA, SA>(R, slab, rhs);
    else
      BackwardSupernodeLean<NB, SB>(R, slab, rhs);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#if defined(CXK_DEBUG_STAMPS) || defined(CXK_CHAIN_STAMPS)
    if (nstamp < 8) tstamp[nstamp++] = __builtin_amdgcn_s_memtime();
#endif
  }
#if defined(CXK_DEBUG_STAMPS) || defined(CXK_CHAIN_STAMPS)
  if (threadIdx.x == 0)
    for (int i = 0; i < 8; i++) g_cxk_stamp[80 + i] = i < nstamp ? tstamp[i] : 0;
#endif
}

// ---------------------------------------------------------------------------------------
// Mid-size supernodes (ns > 32 or s > 16, panel still LDS resident): ONE WORKGROUP per
// supernode.  Same arithmetic as the register kernels -- right-looking elimination with
// reciprocal scaling and fma updates, published updates as fma chains over the solved panel --
// executed block-wide on an LDS copy [diag ns x ns | off ns x s | rhs ns], two barriers per column.
// MODE as in tree_sweep.  grid = supernodes of the level (positions base0 ..).
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
tree_sweep_block(FactorPlan P, int base0, double* __restrict__ slab, double* __restrict__ rhs,
                 int* __restrict__ fail) {
  extern __shared__ double lds[];
  __shared__ int s_bad;
  const SnRec R = LoadRec(P.rec, base0 + blockIdx.x);
  const int ns = R.ns, s = R.nsep, tid = threadIdx.x, nt = blockDim.x;
  double* D = slab + R.diag_off;
  double* B = slab + R.offd_off;
  double* sD = lds;
  double* sB = sD + ns * ns;
  double* sb = sB + ns * s;
  if (tid == 0) s_bad = 0;
  if (MODE == 2) {
    // b <- L^{-T} (b - sum_c off[:,c] y[sep c]); separator terms in the reference's order
    for (int q = tid; q < ns * ns; q += nt) sD[q] = D[q];
    for (int i = tid; i < ns; i += nt) {
      double acc = rhs[R.start + i];
      for (int q = R.bs_beg; q < R.bs_end; q++) acc -= B[i + (size_t)P.bs_c[q] * ns] * rhs[P.bs_row[q]];
      sb[i] = acc;
    }
    __syncthreads();
    for (int k = ns - 1; k >= 0; k--) {
      if (tid == 0) sb[k] = sb[k] * (1.0 / sD[k + k * ns]);
      __syncthreads();
      const double yk = sb[k];
      for (int i = tid; i < k; i += nt) sb[i] = fma(-sD[k + i * ns], yk, sb[i]);
      __syncthreads();
    }
    for (int i = tid; i < ns; i += nt) rhs[R.start + i] = sb[i];
    return;
  }
  const bool with_matrix = MODE == 0;
  const bool with_rhs = rhs != nullptr;
  {
    const int nd = ns * ns, total = nd + ns * s;
    for (int q = tid; q < total; q += nt) lds[q] = q < nd ? D[q] : B[q - nd];
    if (with_rhs)
      for (int i = tid; i < ns; i += nt) sb[i] = rhs[R.start + i];
  }
  __syncthreads();
  if (with_matrix)
    for (int t = R.tg_beg + tid; t < R.tg_end; t += nt) {
      const int loc = P.tg_loc[t];
      double acc = lds[loc];
      const int q1 = P.tr_ptr[t + 1];
      for (int q = P.tr_ptr[t]; q < q1; q++) acc -= P.upd[P.tr_src[q]];
      lds[loc] = acc;
    }
  if (with_rhs)
    for (int i = tid; i < ns; i += nt) {
      double acc = sb[i];
      const int q1 = P.fs_ptr[R.start + i + 1];
      for (int q = P.fs_ptr[R.start + i]; q < q1; q++) acc -= P.updb[P.fs_src[q]];
      sb[i] = acc;
    }
  __syncthreads();
  // extra columns carried along: off block (factor sweep only; solved already otherwise), rhs
  const int ext = s + (with_rhs ? 1 : 0), c0 = with_matrix ? 0 : s;
  for (int k = 0; k < ns; k++) {
    double inv;
    if (with_matrix) {
      const double d = sD[k + k * ns];
      double root;
      SqrtAndInverse(d, root, inv);
      if (!(d > 0.0) && tid == 0) s_bad = 1;
      __syncthreads();  // everybody has read the pivot
      for (int i = k + tid; i < ns; i += nt) sD[i + k * ns] = (i == k) ? root : sD[i + k * ns] * inv;
    } else {
      inv = 1.0 / sD[k + k * ns];
    }
    for (int c = c0 + tid; c < ext; c += nt) {
      double* col = c < s ? sB + c * ns : sb;
      col[k] = col[k] * inv;
    }
    __syncthreads();
    const int rows = ns - k - 1;
    const int dcols = with_matrix ? rows : 0;
    for (int idx = tid; idx < rows * (dcols + ext - c0); idx += nt) {
      const int i = k + 1 + idx % rows, cc = idx / rows;
      const double lik = sD[i + k * ns];
      if (cc < dcols) {
        const int j = k + 1 + cc;
        if (j <= i) sD[i + j * ns] = fma(-lik, sD[j + k * ns], sD[i + j * ns]);
      } else {
        const int c = c0 + cc - dcols;
        double* col = c < s ? sB + c * ns : sb;
        col[i] = fma(-lik, col[k], col[i]);
      }
    }
    __syncthreads();
  }
  if (with_matrix && s_bad) {
    if (tid == 0) atomicExch(fail, 1);
    return;
  }
  if (with_matrix) {
    for (int q = tid; q < ns * ns; q += nt)
      if (q % ns >= q / ns) D[q] = sD[q];
    for (int q = tid; q < ns * s; q += nt) B[q] = sB[q];
    const int* dst = P.pub_dst + R.upd_off;
    const int npairs = s * (s + 1) / 2;
    for (int t = tid; t < npairs; t += nt) {
      int k = 0, rem = t;
      while (rem >= s - k) {
        rem -= s - k;
        k++;
      }
      const int j = k + rem;
      double dot = 0;
      for (int i = 0; i < ns; i++) dot = fma(sB[i + k * ns], sB[i + j * ns], dot);
      P.upd[dst[t]] = dot;
    }
  }
  if (with_rhs) {
    for (int i = tid; i < ns; i += nt) rhs[R.start + i] = sb[i];
    const int* dst = P.pubb_dst + R.updb_off;
    for (int c = tid; c < s; c += nt) {
      double dot = 0;
      for (int i = 0; i < ns; i++) dot = fma(sB[i + c * ns], sb[i], dot);
      P.updb[dst[c]] = dot;
    }
  }
}

// ---------------------------------------------------------------------------------------
// LDLT path (programs with equality constraints: multipliers make the KKT matrix indefinite).
// Reference: BlockLDLTInPlace block_triangular_operations.cc:315-349 over Eigen::RLDLT
// (RLDLT.h:298-431: diagonal pivoting on the largest |diagonal|, left-looking column update,
// pivots with |d| <= 1e-9 clamped to +-1e-9), solves ApplyBlockInverseOfMD :265-299 and
// ApplyBlockInverseOfMTranspose :222-263.  One workgroup per supernode, panel in LDS
// [diag ns x ns | off ns x s | rhs ns | temp ns]; `tr` holds the transpositions (local indices)
// of every supernode by first permuted index.  Published updates:
//   U[k][j] = sum_r (D_r off[r][k]) off[r][j]   with off = D^-1 L^-1 P off
//   t[c]    = sum_r off[r][c] b_r               with b = L^-1 P b (D^-1 is applied afterwards)
// ---------------------------------------------------------------------------------------
// HBM: the panel image lives in `ws` (global memory) instead of LDS -- supernodes beyond LDS, one
// workgroup of 1024 threads each, the same operations in the same order (a workgroup barrier orders
// its threads' global accesses as it orders their LDS accesses); the extra columns (off block,
// right-hand side) are then swept by all threads together, column step by column step, instead of
// one thread per column.  Slow (every step is a round trip to L2) but complete: the blocked LDLT with
// the reference's pivot rule needs the whole trailing diagonal at every step.
template <int MODE, bool HBM = false>
__global__ void __launch_bounds__(HBM ? 1024 : 256)
tree_sweep_block_ldlt(FactorPlan P, int base0, double* __restrict__ slab, double* __restrict__ rhs,
                      int* __restrict__ tr_all, int* __restrict__ regularized, double* __restrict__ ws = nullptr) {
  extern __shared__ double lds_dyn[];
  double* lds = HBM ? ws : lds_dyn;
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_piv;
  const SnRec R = LoadRec(P.rec, base0 + blockIdx.x);
  const int ns = R.ns, s = R.nsep, tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  double* D = slab + R.diag_off;
  double* B = slab + R.offd_off;
  double* sD = lds;
  double* sB = sD + ns * ns;
  double* sb = sB + ns * s;
  double* temp = sb + ns;
  int* str = reinterpret_cast<int*>(temp + ns);  // LDS copy of the transpositions
  int* tr = tr_all + R.start;
  if (MODE == 2) {
    for (int q = tid; q < ns * ns; q += nt) sD[q] = D[q];
    for (int i = tid; i < ns; i += nt) {
      double acc = rhs[R.start + i];
      for (int q = R.bs_beg; q < R.bs_end; q++) acc -= B[i + (size_t)P.bs_c[q] * ns] * rhs[P.bs_row[q]];
      sb[i] = acc;
    }
    __syncthreads();
    for (int k = ns - 1; k >= 0; k--) {  // unit-lower-transposed solve
      const double yk = sb[k];
      for (int i = tid; i < k; i += nt) sb[i] = fma(-sD[k + i * ns], yk, sb[i]);
      __syncthreads();
    }
    if (tid == 0)
      for (int k = ns - 1; k >= 0; k--) {  // P^T
        const int t = tr[k];
        if (t != k) {
          const double v = sb[k];
          sb[k] = sb[t];
          sb[t] = v;
        }
      }
    __syncthreads();
    for (int i = tid; i < ns; i += nt) rhs[R.start + i] = sb[i];
    return;
  }
  const bool with_matrix = MODE == 0;
  const bool with_rhs = rhs != nullptr;
  {
    const int nd = ns * ns, total = nd + ns * s;
    for (int q = tid; q < total; q += nt) lds[q] = q < nd ? D[q] : B[q - nd];
    if (with_rhs)
      for (int i = tid; i < ns; i += nt) sb[i] = rhs[R.start + i];
  }
  if (!with_matrix)
    for (int i = tid; i < ns; i += nt) str[i] = tr[i];
  __syncthreads();
  if (with_matrix)
    for (int t = R.tg_beg + tid; t < R.tg_end; t += nt) {
      const int loc = P.tg_loc[t];
      double acc = lds[loc];
      const int q1 = P.tr_ptr[t + 1];
      for (int q = P.tr_ptr[t]; q < q1; q++) acc -= P.upd[P.tr_src[q]];
      lds[loc] = acc;
    }
  if (with_rhs)
    for (int i = tid; i < ns; i += nt) {
      double acc = sb[i];
      const int q1 = P.fs_ptr[R.start + i + 1];
      for (int q = P.fs_ptr[R.start + i]; q < q1; q++) acc -= P.updb[P.fs_src[q]];
      sb[i] = acc;
    }
  __syncthreads();
  if (with_matrix) {
    if (ns == 1) {  // RLDLT.h:311-330: clamps without reporting
      if (tid == 0) {
        if (fabs(sD[0]) < 1e-9) sD[0] = sD[0] < 0 ? -1e-9 : 1e-9;
        tr[0] = 0;
        str[0] = 0;
      }
      __syncthreads();
    } else {
      for (int k = 0; k < ns; k++) {
        // first largest |diagonal| of the trailing part
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < ns; i += nt) {
          const double v = fabs(sD[i + i * ns]);
          if (v > best) {
            best = v;
            bi = i;
          }
        }
        for (int off = 32; off > 0; off >>= 1) {
          const double ov = __shfl_xor(best, off, 64);
          const int oi = __shfl_xor(bi, off, 64);
          if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
          }
        }
        if (lane == 0) {
          s_val[wave] = best;
          s_idx[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
          double b = s_val[0];
          int p = s_idx[0];
          for (int w = 1; w < (nt >> 6); w++)
            if (s_val[w] > b || (s_val[w] == b && s_idx[w] < p)) {
              b = s_val[w];
              p = s_idx[w];
            }
          s_piv = p;
          tr[k] = p;
          str[k] = p;
        }
        __syncthreads();
        const int big = s_piv;
        if (big != k) {  // symmetric transposition on the lower triangle (RLDLT.h:343-362)
          for (int j = tid; j < k; j += nt) {
            const double t = sD[k + j * ns];
            sD[k + j * ns] = sD[big + j * ns];
            sD[big + j * ns] = t;
          }
          for (int i = big + 1 + tid; i < ns; i += nt) {
            const double t = sD[i + k * ns];
            sD[i + k * ns] = sD[i + big * ns];
            sD[i + big * ns] = t;
          }
          for (int i = k + 1 + tid; i < big; i += nt) {
            const double t = sD[i + k * ns];
            sD[i + k * ns] = sD[big + i * ns];
            sD[big + i * ns] = t;
          }
          if (tid == 0) {
            const double t = sD[k + k * ns];
            sD[k + k * ns] = sD[big + big * ns];
            sD[big + big * ns] = t;
          }
          __syncthreads();
        }
        for (int j = tid; j < k; j += nt) temp[j] = sD[j + j * ns] * sD[k + j * ns];
        __syncthreads();
        for (int i = k + tid; i < ns; i += nt) {  // row k: the pivot; rows > k: A21
          double acc = 0;
          for (int j = 0; j < k; j++) acc += sD[i + j * ns] * temp[j];
          sD[i + k * ns] -= acc;
        }
        __syncthreads();
        if (tid == 0) {
          const double akk = sD[k + k * ns];
          if (!(fabs(akk) > 1e-9)) {
            sD[k + k * ns] = akk < 0 ? -1e-9 : 1e-9;
            atomicExch(regularized, 1);
          }
        }
        __syncthreads();
        const double akk = sD[k + k * ns];
        for (int i = k + 1 + tid; i < ns; i += nt) sD[i + k * ns] /= akk;
        __syncthreads();
      }
    }
  }
  // extra columns: off block (factor sweep only) and rhs.  One thread per column:
  // P, then the unit-lower solve; off additionally scaled by D^-1.
  const int ext = s + (with_rhs ? 1 : 0), c0 = with_matrix ? 0 : s;
  if constexpr (HBM) {
    for (int c = c0 + tid; c < ext; c += nt) {
      double* col = c < s ? sB + (size_t)c * ns : sb;
      for (int k = 0; k < ns; k++) {
        const int t = str[k];
        if (t != k) {
          const double v = col[k];
          col[k] = col[t];
          col[t] = v;
        }
      }
    }
    __syncthreads();
    const int ncol = ext - c0;
    for (int j = 0; j + 1 < ns; j++) {  // every entry takes its terms in the order of the one-thread sweep
      const int below = ns - j - 1;
      for (int q = tid; q < below * ncol; q += nt) {
        const int c = c0 + q / below, i = j + 1 + q % below;
        double* col = c < s ? sB + (size_t)c * ns : sb;
        col[i] -= sD[i + (size_t)j * ns] * col[j];
      }
      __syncthreads();
    }
    if (with_matrix)
      for (int q = tid; q < ns * s; q += nt) sB[q] = (1.0 / sD[(q % ns) * (size_t)(ns + 1)]) * sB[q];
  } else
  for (int c = c0 + tid; c < ext; c += nt) {
    double* col = c < s ? sB + c * ns : sb;
    for (int k = 0; k < ns; k++) {
      const int t = str[k];
      if (t != k) {
        const double v = col[k];
        col[k] = col[t];
        col[t] = v;
      }
    }
    for (int j = 0; j < ns; j++) {
      const double cj = col[j];
      for (int i = j + 1; i < ns; i++) col[i] -= sD[i + j * ns] * cj;
    }
    if (c < s)
      for (int r = 0; r < ns; r++) col[r] = (1.0 / sD[r + r * ns]) * col[r];
  }
  __syncthreads();
  if (with_matrix) {
    for (int q = tid; q < ns * ns; q += nt)
      if (q % ns >= q / ns) D[q] = sD[q];
    for (int q = tid; q < ns * s; q += nt) B[q] = sB[q];
    const int* dst = P.pub_dst + R.upd_off;
    const int npairs = s * (s + 1) / 2;
    for (int t = tid; t < npairs; t += nt) {
      int k = 0, rem = t;
      while (rem >= s - k) {
        rem -= s - k;
        k++;
      }
      const int j = k + rem;
      double dot = 0;
      for (int r = 0; r < ns; r++) dot += (sD[r + r * ns] * sB[r + k * ns]) * sB[r + j * ns];
      P.upd[dst[t]] = dot;
    }
  }
  if (with_rhs) {
    const int* dst = P.pubb_dst + R.updb_off;
    for (int c = tid; c < s; c += nt) {
      double dot = 0;
      for (int r = 0; r < ns; r++) dot += sB[r + c * ns] * sb[r];
      P.updb[dst[c]] = dot;
    }
    __syncthreads();
    for (int i = tid; i < ns; i += nt) rhs[R.start + i] = (1.0 / sD[i + i * ns]) * sb[i];
  }
}

#ifndef CXK_DEVICE_FUNCTIONS_ONLY
// ---------------------------------------------------------------------------------------
// Iterative refinement (SupernodalKKTSolver::SolveInPlace, kkt_solver.cc:233-261):
//   y <- y + K^-1 (b - K y)   with K = the assembled matrix, kept in `slab0` by the factor sweep.
// The reference multiplies a dense N x N copy; here K y comes from the supernodal blocks:
// kkt_matvec (one workgroup per supernode) forms  u[sn] = sym(diag) y[sn] + off y[sep]  and
// publishes  t[c] = off[:,c] . y[sn]  into the forward-solve slots' twin `mvb`; refine_residual
// gathers them per row in list order (deterministic) into  r = b - K y, saves y and puts r in its place.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
kkt_matvec(FactorPlan P, const double* __restrict__ slab0, const double* __restrict__ y,
           double* __restrict__ u, double* __restrict__ mvb) {
  const SnRec R = LoadRec(P.rec, blockIdx.x);
  const int ns = R.ns, s = R.nsep, tid = threadIdx.x, nt = blockDim.x;
  const double* D = slab0 + R.diag_off;
  const double* B = slab0 + R.offd_off;
  for (int i = tid; i < ns; i += nt) {
    double acc = 0;
    for (int j = 0; j < ns; j++) acc += (j <= i ? D[i + (size_t)j * ns] : D[j + (size_t)i * ns]) * y[R.start + j];
    for (int q = R.bs_beg; q < R.bs_end; q++) acc += B[i + (size_t)P.bs_c[q] * ns] * y[P.bs_row[q]];
    u[R.start + i] = acc;
  }
  for (int c = tid; c < s; c += nt) {
    double dot = 0;
    for (int i = 0; i < ns; i++) dot += B[i + (size_t)c * ns] * y[R.start + i];
    mvb[P.pubb_dst[R.updb_off + c]] = dot;
  }
}

__global__ void refine_residual(int N, const double* __restrict__ rhs0, const double* __restrict__ u,
                                const int* __restrict__ fs_ptr, const int* __restrict__ fs_src,
                                const double* __restrict__ mvb, double* __restrict__ y,
                                double* __restrict__ ysave) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) {
    double ky = u[p];
    for (int q = fs_ptr[p]; q < fs_ptr[p + 1]; q++) ky += mvb[fs_src[q]];
    ysave[p] = y[p];
    y[p] = rhs0[p] - ky;
  }
}

__global__ void refine_add(int N, const double* __restrict__ ysave, double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) y[p] = ysave[p] + y[p];
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
  const int* xs_pt;   // exchange entry -> list of this rank's published updates into it (pt_ptr), -1 none
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
  int tag;  // a failed pivot in a first level with the assembly folded in is reported as fail[1] == tag
  double* x;
  double cb, cq, cw;
};

__global__ void __launch_bounds__(256) exchange_pack(ExchangeArgs a) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // slab entries: partial assembled value minus the Schur updates of this rank's subtrees (summed
  // in slot order, then subtracted: what a separate fold launch used to leave in the slab)
  for (int64_t i = gid; i < a.n_xs; i += stride) {
    double v = a.slab[a.xs_off[i]];
    const int t = a.xs_pt[i];
    if (t >= 0) {
      double s = 0;
      for (int q = a.pt_ptr[t]; q < a.pt_ptr[t + 1]; q++) s += a.upd[a.pt_src[q]];
      v -= s;
    }
    a.x[i] = v;
  }
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
    // both forms of a failed pivot travel: fail[0] (level kernels) and the tagged word of the
    // fused first level -- otherwise only the failing rank would know and the ranks would part ways
    a.x[o + 2] = (a.fail[0] != 0 || (a.tag != 0 && a.fail[1] == a.tag)) ? 1.0 : 0.0;
    a.x[o + 3] = 0;
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

// Solve-only exchange (right-hand sides after the factorization: mu selection, line search,
// cxk_solve_inplace): only the forward-solve contributions of this rank's subtrees to the top
// variables travel, x[j] = sum of its published t values; after the sum all-reduce every rank
// subtracts the total from its (replicated, complete) right-hand side of the top.
__global__ void __launch_bounds__(256) exchange_pack_solve(ExchangeArgs a) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < a.n_xv; j += (int64_t)gridDim.x * blockDim.x) {
    double f = 0;
    for (int q = a.pf_ptr[j]; q < a.pf_ptr[j + 1]; q++) f += a.updb[a.pf_src[q]];
    a.x[j] = f;
  }
}
__global__ void __launch_bounds__(256) exchange_unpack_solve(ExchangeArgs a) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < a.n_xv; j += (int64_t)gridDim.x * blockDim.x)
    a.y[a.xv_idx[j]] -= a.x[j];
}
// Factor-only exchange (cxk_factor_async on a sharded context): the forward slots hold nothing
// meaningful, the right-hand side of the top is left alone.
__global__ void __launch_bounds__(256) exchange_unpack_matrix(ExchangeArgs a) {
  const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gid; i < a.n_xs; i += stride) a.slab[a.xs_off[i]] = a.x[i];
  for (int64_t j = gid; j < a.n_xv; j += stride) {
    const int p = a.xv_idx[j];
    a.AW[p] = a.x[a.n_xs + j];
    a.AQc[p] = a.x[a.n_xs + a.n_xv + j];
  }
  if (gid == 0) {
    const int64_t o = a.n_xs + 3 * (int64_t)a.n_xv;
    a.sys_sc[0] = a.x[o];
    a.sys_sc[1] = a.x[o + 1];
    if (a.x[o + 2] > 0.0) *a.fail = 1;
  }
}

// out[i] = count[i] ? in[i] : 0 -- a rank's share of a vector whose entries are spread over the
// ranks (own subtrees; the replicated top counts on rank 0 only): the sum all-reduce of these
// shares is the whole vector.
__global__ void masked_copy(int n, const unsigned char* __restrict__ count, const double* __restrict__ in,
                            double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = count[i] ? in[i] : 0.0;
}
// per-constraint pairs (2 doubles each) of the constraints this rank owns, zero elsewhere
__global__ void masked_copy_pairs(int K, const unsigned char* __restrict__ owned, const double* __restrict__ in,
                                  double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * K; i += gridDim.x * blockDim.x)
    out[i] = owned[i >> 1] ? in[i] : 0.0;
}

// step_scalars over this rank's share of the variables (see masked_copy); out[4], out[5] are the
// already complete <w,c>, <c,Qc>.  The caller sum-reduces out[0..3] across ranks.
__global__ void __launch_bounds__(1024)
step_scalars_masked(int N, const unsigned char* __restrict__ count, const double* __restrict__ b,
                    const double* __restrict__ AQc, const double* __restrict__ y,
                    const double* __restrict__ sys_sc, double* __restrict__ out) {
  __shared__ double red[16];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int p = threadIdx.x; p < N; p += blockDim.x) {
    if (!count[p]) continue;
    const double vb = b[p], vq = AQc[p], vy = y[p];
    s0 = fma(vb, vy, s0);
    s1 = fma(vq, vy, s1);
    s2 = fma(vb, vb, s2);
    s3 = fma(vq, vq, s3);
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

// permuted <-> original order copies
__global__ void permute_gather(int N, const int* __restrict__ idx, const double* __restrict__ in,
                               double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
    out[i] = in[idx[i]];
}

#endif  // CXK_DEVICE_FUNCTIONS_ONLY

}  // namespace cxk
