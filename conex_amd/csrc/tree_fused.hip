// tree_fused<NA, SA, NB, SB>: assembly gather + supernodal Cholesky (+ forward substitution) + back
// substitution of the WHOLE elimination tree in one launch, one wavefront per supernode.
//
// Reference semantics (order of every sum included): SupernodalKKTSolver::Assemble
// (kkt_solver.cc:164-170, supernodal_assembler.cc:113-165), AssembleSchurComplementResiduals
// (constraint_manager.h:107-124), BlockCholeskyInPlace (block_triangular_operations.cc:184-219),
// ApplyBlockInverseInPlace / ...OfTransposeInPlace (:114-182), right-hand side cone_program.cc:409-411.
//
// Why one launch.  The level-by-level sweeps (kernels_kkt.hip.h) pay, per level of the tree, a
// kernel boundary (~1.5 us), a record round trip and a data round trip before ~2 us of elimination
// -- 46 us of dependency latency at BASELINE config 4 (5 levels up, 4 down) in which the chip is
// nearly idle.  Here every supernode's wavefront is resident from the start: it fetches its
// record, its panel (straight from the Schur blocks: own block by position, further sources from
// a dense list), its publish destinations and pull locations while its descendants still work,
// and then only WAITS for their values.
//
// Hand-off without flags or fences.  A published value is its own "ready" flag: every slot a
// consumer reads starts as kFusedSentinel (a signalling-NaN bit pattern no arithmetic produces),
// the producer overwrites it with ONE 8-byte write-through store (sc1: agent scope), the consumer
// polls with sc1 loads -- two rounds of them in flight, so that a value is in registers one memory
// latency after it became visible -- until no slot of its own shows the sentinel.  An aligned 8-byte store is
// not torn, and each value is waited for individually, so no ordering between values is needed
// (MI355X_MICROARCH.md, "data-tagged granules").  Slots come in TWO sets used by alternate runs:
// run g publishes into set g & 1 and re-arms the same slot of the other set, which nobody reads
// before the next launch -- no consumer ever writes, and re-arming needs no ordering either.
// Three families of slots: Schur updates (upd2, the consumer-ordered slots of BuildPlans), forward
// values (updb2), and the solution entries a descendant's back substitution reads (ysig).
//
// Deadlock freedom: a supernode waits only for supernodes at LOWER positions (the records are in
// level order), workgroup index = position, and every wait is bounded (kFusedSpinLimit polls):
// when it runs out the wavefront reports failure (fail[1] = tag and the pinned host word) and
// carries on with what it has, so the grid always drains.
//
// Arithmetic is FactorSupernodeLean's and BackwardSupernodeLean's (same expressions in the same
// order): the factor, the direction and AW / AQc / <w,c> / <c,Qc> are the bits the level kernels
// with the separate gather produce.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>

#define CXK_DEVICE_FUNCTIONS_ONLY
#include "kernels_kkt.hip.h"
#include "kernels_kkt_top.hip.h"  // ElimWide: the elimination over four mirrored DPP rows (33 .. 64 columns)
#include "tree_fused.h"

namespace cxk {

constexpr int kFusedSpinLimit = 1 << 18;

// Diagnostic build (-DCXK_FUSED_STAMPS, `make dbg`): every wavefront keeps seven time stamps
// (s_memrealtime: 100 MHz, one clock for the whole chip) in registers and writes them at its end to
// a buffer nothing else reads (tools/fused_tree_stamps.py).
#ifdef CXK_FUSED_STAMPS
constexpr int kFusedStampWaves = 8192;
__device__ long long g_fused_tree_stamp[kFusedStampWaves * 16];
#define FT_STAMP(i) ft_stamp[i] = __builtin_amdgcn_s_memrealtime()
#define FT_COUNT(i, v) ft_stamp[i] = (v)
#define FT_STAMP_DECL long long ft_stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define FT_STAMP_FLUSH(level)                                                              \
  do {                                                                                     \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < kFusedStampWaves) {                        \
      ft_stamp[15] = (level);                                                              \
      for (int i_ = 0; i_ < 16; i_++) g_fused_tree_stamp[blockIdx.x * 16 + i_] = ft_stamp[i_]; \
    }                                                                                      \
  } while (0)
#else
#define FT_STAMP(i) do { } while (0)
#define FT_COUNT(i, v) do { } while (0)
#define FT_STAMP_DECL do { } while (0)
#define FT_STAMP_FLUSH(level) do { } while (0)
#endif

__device__ __forceinline__ double LoadAgent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void StoreAgent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool IsSentinel(double v) {
  return (unsigned long long)__double_as_longlong(v) == kFusedSentinel;
}
__device__ __forceinline__ double SentinelValue() { return __longlong_as_double((long long)kFusedSentinel); }
__device__ __forceinline__ int64_t Join64(int lo, int hi) { return ((int64_t)hi << 32) | (uint32_t)lo; }

// `units` x 64 cycles without issuing anything (s_sleep takes a constant of at most 127)
__device__ __forceinline__ void SleepUnits(int units) {
  for (; units >= 32; units -= 32) __builtin_amdgcn_s_sleep(32);
  for (; units >= 4; units -= 4) __builtin_amdgcn_s_sleep(4);
}

__device__ __forceinline__ void ReportTimeout(const FusedTreeArgs& A) {
  if ((threadIdx.x & 63) == 0) {
    atomicExch(A.fail + 1, A.tag);
    __hip_atomic_store(A.host_flag, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// acc += w[lane J of the own 16-lane DPP row] * v (the multiply-add of DppColumns, kernels_kkt.hip.h; operands
// through DppOperandFence first)
template <int J>
__device__ __forceinline__ void FmacRowBcast(double& acc, double w, double v) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(v), "n"(J));
}

// One supernode: w = this lane's word of its record (kFusedRecWords dwords, one per lane).
// UP_ONLY: the first of two launches (trees with more supernodes than the chip holds wavefronts: a
// supernode that waited for its ANCESTORS' solution while holding its slot could keep them from ever
// starting): stops after publishing, leaves the factor and the forward-solved right-hand side in
// memory; the back substitution is tree_fused_down's.
// FROM_X: a supernode of the replicated top of a sharded context.  Its panel, AW / AQc and the
// forward values of the subtrees below come from the all-reduced exchange buffer (what
// exchange_unpack would have put into the slab and y: the unpack is this load phase), nothing from G.
// NRHS = 3 (kFusedTriple): three right-hand sides ride through the sweep -- bs b, cs AQc and AW -- so that the
// right-hand side of the mu selection (-bs b + cs AQc, cone_program.cc:181) AND the Newton direction for
// whatever mu comes out of it (k (bs b + cs AQc) - 2 AW, :409-411) are combinations of its three
// solutions: the interior-point iteration needs no second sweep over the tree.  Right-hand side 0 uses the
// hand-off slots of every other launch (set = run parity); 1 and 2 have slots of their own behind them
// (fwd_stride / N apart), whose two sets alternate with the TRIPLE launches only (A.tgen).
template <int NSMAX, int SMAX, bool UP_ONLY = false, bool FROM_X = false, int NRHS = 1>
__device__ __forceinline__ void FusedSupernode(const FusedTreeArgs& A, const int w, double* __restrict__ my) {
  static_assert(NSMAX + SMAX <= 64, "one lane per panel row");
  static_assert(NRHS == 1 || (NRHS == 3 && !UP_ONLY && !FROM_X && NSMAX <= 32), "three right-hand sides: the one-launch sweep of a single GPU");
  // (three right-hand sides: pull lists four slots at a time -- the registers of eight are spoken for)
  constexpr int RB = NSMAX + SMAX, MMAX = NRHS > 1 ? 4 : kFastSlots, MFMAX = MMAX, XMAX = kFusedExtraSlots;
  // The right-hand sides ride through the elimination as ROWS of the panel where a free lane takes them
  // (lanes RL .. of DPP row 1, FusedRhsLane): a row under the panel is forward-solved by the column updates
  // every lane takes part in anyway -- y_j = (b_j - sum_k L[j][k] y_k) / L[j][j] in its entry j, -off[:,c] . y
  // in its trailing entry c: the same fma chains, the same bits -- so the nine instructions per pivot and
  // right-hand side of the column form (two v_readlane, a multiply, an fma, selects: 135 on the critical path
  // of a 15-column supernode, 405 with three right-hand sides) become one transposition through LDS in front
  // of the elimination; behind it the values sit where the image of L's rows puts them anyway.
  constexpr int RL = FusedRhsLane(NSMAX, SMAX);
  constexpr bool ROWRHS = RL >= 0 && !UP_ONLY && !FROM_X;
  constexpr int NCOL = ROWRHS ? 0 : NRHS;  // right-hand side COLUMNS a[RB ..] of the elimination
  constexpr int NVMAX = SMAX * (SMAX + 1) / 2 + NRHS * SMAX;  // values a supernode publishes
  constexpr int PR = (NVMAX + 63) / 64;                // ... in this many store instructions
  const int lane = threadIdx.x & 63;
  FT_STAMP_DECL;
  FT_STAMP(0);
  FT_COUNT(10, __builtin_amdgcn_s_getreg((31 << 11) | 4));   // HW_ID: wave, SIMD, CU, SH, SE (tools/fused_tree_stamps.py: placement)
  FT_COUNT(11, __builtin_amdgcn_s_getreg((31 << 11) | 20));  // XCC_ID
  const SnRec R = DecodeRec(w);
  FT_STAMP(1);  // the record is here
  auto f = [&](int i) { return __builtin_amdgcn_readlane(w, 32 + i); };
  const int ns = R.ns, s = R.nsep;
  const bool is_row = lane < ns;
  const int sc = lane - NSMAX;
  const bool is_sep = sc >= 0 && sc < s;
  double* base = A.slab + R.diag_off;
  const unsigned rel = (unsigned)(R.offd_off - R.diag_off);
  const unsigned o0 = is_row ? (unsigned)lane : (is_sep ? rel + (unsigned)(sc * ns) : 0u);
  const unsigned st = is_row ? (unsigned)ns : 1u;
  const int lim = is_row ? lane + 1 : (is_sep ? ns : 0);  // valid j < lim
  const int gen = A.gen;
  double* handG = A.hand + (int64_t)gen * A.hand_stride;        // this run's set of hand-off slots
  double* handO = A.hand + (int64_t)(gen ^ 1) * A.hand_stride;  // the other one: re-armed at the end
  double* ysG = A.ysig + (int64_t)gen * A.ysig_stride;
  double* ysO = A.ysig + (int64_t)(gen ^ 1) * A.ysig_stride;
  // (right-hand sides 1, 2: this run's set and the other one)
  double* handT = A.hand + (int64_t)A.tgen * A.hand_stride;
  double* handTO = A.hand + (int64_t)(A.tgen ^ 1) * A.hand_stride;
  double* ysT = A.ysig + (int64_t)A.tgen * A.ysig_stride;
  double* ysTO = A.ysig + (int64_t)(A.tgen ^ 1) * A.ysig_stride;
  const int pub_beg = __builtin_amdgcn_readlane(w, 23);
  const int npairs = s * (s + 1) / 2, nv = npairs + NRHS * s;

  // ---- load phase, first trip: everything whose address follows from the record
  const double* Gk = A.G + (FROM_X ? 0 : Join64(f(0), f(1)));
  const int64_t roff = FROM_X ? 0 : Join64(f(2), f(3));
  const int M = f(4);
  const int q = is_row ? lane : (is_sep ? ns + sc : 0);
  // position 255 = a structural fill-in row (a separator variable the constraint does not contain: the
  // deferred variables a segmented chain carries along, symbolic.h): its entries start as zeros
  const int mypr = (__builtin_amdgcn_ds_bpermute(4 * (32 + 6 + (q >> 2)), w) >> (8 * (q & 3))) & 255;
  const int myp = mypr == 255 ? 0 : mypr;
  const int has_fill = FROM_X ? 0 : f(5);
  double a[NSMAX + SMAX + NCOL];
  double rv[NRHS];  // this lane's entry of every right-hand side, until the elimination
  double fwx = 0.0;  // FROM_X: what this rank's ... every rank's subtrees subtract from the right-hand side
  if constexpr (FROM_X) {
    // entry (row, j) of the diagonal block sits at xs_base + j ns - j (j - 1) / 2 + (row - j) (lower
    // triangle, column by column), entry (j, c) of the off block behind the triangle at c ns + j
    const int64_t xsb = Join64(f(0), f(1));
    const double* xd = A.x + xsb;
    const unsigned tri = (unsigned)(ns * (ns + 1) / 2);
#pragma unroll
    for (int j = 0; j < NSMAX; j++) {
      const unsigned dj = (unsigned)(j * ns - j * (j - 1) / 2);
      const unsigned idx = is_row ? dj + (unsigned)(lane - j) : tri + (unsigned)(sc * ns + j);
      a[j] = xd[(j < lim) ? idx : 0u];
    }
  } else {
    // entry (row, j) of the panel is G(max(p_row, p_j), min(..)) of the own block (lower triangle,
    // column-major).  EVERY lane / column pair gives an address inside the block (positions of
    // padding rows and columns read as 0), so no load needs a predicate or a clamp -- the entries
    // that do not exist are masked after the loads -- and three integer instructions make an address:
    // byte offsets from a scalar base, positions pre-scaled by 8
    const char* gb = reinterpret_cast<const char*>(Gk);
    const unsigned myp8 = 8u * (unsigned)myp;
#pragma unroll
    for (int j = 0; j < NSMAX; j++) {
      const unsigned pj = (unsigned)((f(6 + (j >> 2)) >> (8 * (j & 3))) & 255);  // wave-uniform
      const unsigned pj8 = 8u * (pj == 255u ? 0u : pj);
      const unsigned hi8 = myp8 > pj8 ? myp8 : pj8, lo8 = myp8 > pj8 ? pj8 : myp8;
      a[j] = *reinterpret_cast<const double*>(gb + (__umul24(lo8, (unsigned)M) + hi8));
    }
    if (has_fill) {  // (wave-uniform; the loads above went to valid addresses either way)
#pragma unroll
      for (int j = 0; j < NSMAX; j++) {
        const unsigned pj = (unsigned)((f(6 + (j >> 2)) >> (8 * (j & 3))) & 255);
        a[j] = (mypr == 255 || pj == 255u) ? 0.0 : a[j];
      }
    }
  }
  const int pr = is_row ? myp : (f(6) & 255);
  double awv, aqv;
  if constexpr (FROM_X) {
    const int64_t xv = A.n_xs + f(2) + (is_row ? lane : 0);
    awv = A.x[xv];
    aqv = A.x[xv + A.n_xv];
    fwx = A.x[xv + 2 * (int64_t)A.n_xv];
  } else {
    awv = A.AWc[roff + pr];
    aqv = A.AQcc[roff + pr];
  }
  double rb = A.b[R.start + (is_row ? lane : 0)];
  // where this supernode's values go: lane t of round r publishes value number t + 64 r (the
  // s (s + 1) / 2 Schur updates in the reference's S_S enumeration, then the s forward values)
  int pd[PR > 0 ? PR : 1], prd[PR > 0 ? PR : 1];
  bool pdt[PR > 0 ? PR : 1];  // value of right-hand side 1 or 2: its slot is in the TRIPLE launches' own sets
  pd[0] = prd[0] = 0;
  pdt[0] = false;
#pragma unroll
  for (int r = 0; r < PR; r++) {
    const int t = lane + 64 * r;
    // (right-hand side q of separator variable k: value npairs + q s + k, slot of value npairs + k, q fwd_stride on)
    const int tq = (NRHS > 1 && t >= npairs + s && s > 0) ? (t - npairs) / s : 0;
    const int tk = t - tq * s;  // the value of right-hand side 0 it sits behind
    pdt[r] = tq > 0;
    pd[r] = A.pub[pub_beg + (t < nv ? tk : 0)] + tq * (int)A.fwd_stride;
    // where value t will sit in the scratch image the separator lanes write after the elimination
    // (lane l's registers a[NSMAX ..] and its right-hand side at my[(SMAX + 1) l + c]): Schur update
    // t = (k, c) of the S_S enumeration, or forward value k
    int k = 0, rem = t < npairs ? t : 0;
    while (rem >= s - k && k < s) {
      rem -= s - k;
      k++;
    }
    const int kk = t < npairs ? k : tk - npairs, cc = t < npairs ? k + rem : SMAX + tq;
    if (ROWRHS && t >= npairs)  // (forward value k of right-hand side tq: trailing entry k of its row)
      prd[r] = (SMAX + NRHS) * (RL + (t < nv ? tq : 0)) + (t < nv ? kk : 0);
    else
      prd[r] = (SMAX + NRHS) * (NSMAX + (t < nv ? kk : 0)) + (t < nv ? cc : 0);
  }
  const int ntg = R.tg_end - R.tg_beg;
  int ploc0 = 0, ploc1 = 0;
  if (ntg > 0) {
    ploc0 = A.tg_reg[R.tg_beg + (lane < ntg ? lane : 0)];
    if (ntg > 64) ploc1 = A.tg_reg[R.tg_beg + (lane + 64 < ntg ? lane + 64 : 0)];
  }
  const int xt_beg = f(24), nxt = FROM_X ? 0 : f(25), mx = f(26), rbase = f(27), mr = FROM_X ? 0 : f(28);
  const int64_t xbase = Join64(f(29), f(30));
  // ---- entries with further sources (descendants' separator blocks) and shared variables: the
  // lists, then the values (two more trips, while the descendants are still at work)
  int xloc0 = 0, xloc1 = 0;
  double gx0[XMAX], gx1[XMAX];
#pragma unroll
  for (int i = 0; i < XMAX; i++) gx0[i] = gx1[i] = 0.0;
  if (nxt > 0) {
    const int mxl = mx > 0 ? mx - 1 : 0;
    {
      const int ts = lane < nxt ? lane : 0;
      xloc0 = A.xreg[xt_beg + ts];
      long long xs[XMAX];
#pragma unroll
      for (int i = 0; i < XMAX; i++) xs[i] = A.xsrc[xbase + (int64_t)ts * mx + (i < mx ? i : mxl)];
#pragma unroll
      for (int i = 0; i < XMAX; i++) {
        const double v = A.G[xs[i] >= 0 ? xs[i] : 0];
        gx0[i] = (i < mx && xs[i] >= 0 && lane < nxt) ? v : 0.0;
      }
    }
    if (nxt > 64) {
      const int ts = lane + 64 < nxt ? lane + 64 : 0;
      xloc1 = A.xreg[xt_beg + ts];
      long long xs[XMAX];
#pragma unroll
      for (int i = 0; i < XMAX; i++) xs[i] = A.xsrc[xbase + (int64_t)ts * mx + (i < mx ? i : mxl)];
#pragma unroll
      for (int i = 0; i < XMAX; i++) {
        const double v = A.G[xs[i] >= 0 ? xs[i] : 0];
        gx1[i] = (i < mx && xs[i] >= 0 && lane + 64 < nxt) ? v : 0.0;
      }
    }
  }
  // what assemble_gather would have produced: sums that start from +0.0 (a -0.0 source ends up
  // +0.0); a variable that several constraints share takes ALL its sources from the list, in the
  // gather's order
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = 0.0 + a[j];
  awv = 0.0 + awv;
  aqv = 0.0 + aqv;
  if (mr > 0) {
    const int mrl = mr - 1;
    long long rs[XMAX];
#pragma unroll
    for (int i = 0; i < XMAX; i++) rs[i] = A.rsrc[rbase + (is_row ? lane : 0) * mr + (i < mr ? i : mrl)];
    double ax[XMAX], qx[XMAX];
#pragma unroll
    for (int i = 0; i < XMAX; i++) {
      ax[i] = A.AWc[rs[i] >= 0 ? rs[i] : 0];
      qx[i] = A.AQcc[rs[i] >= 0 ? rs[i] : 0];
    }
    if (rs[0] >= 0) {  // (a listed row: the own constraint is one of the listed sources)
      awv = 0.0;
      aqv = 0.0;
    }
#pragma unroll
    for (int i = 0; i < XMAX; i++)
      if (i < mr && rs[i] >= 0) {
        awv += ax[i];
        aqv += qx[i];
      }
    // (a variable more than XMAX constraints share: the rest of its list, XMAX at a time)
    for (int c0 = XMAX; c0 < mr; c0 += XMAX) {
#pragma unroll
      for (int i = 0; i < XMAX; i++) rs[i] = A.rsrc[rbase + (is_row ? lane : 0) * mr + (c0 + i < mr ? c0 + i : mrl)];
#pragma unroll
      for (int i = 0; i < XMAX; i++) {
        ax[i] = A.AWc[rs[i] >= 0 ? rs[i] : 0];
        qx[i] = A.AQcc[rs[i] >= 0 ? rs[i] : 0];
      }
#pragma unroll
      for (int i = 0; i < XMAX; i++)
        if (c0 + i < mr && rs[i] >= 0) {
          awv += ax[i];
          aqv += qx[i];
        }
    }
  }
  // the expressions of build_rhs / build_rhs_comb, term for term
  if constexpr (NRHS == 1) {
    if (A.comb)
      rb = A.cb * rb + A.cq * aqv + A.cw * awv;
    else
      rb = A.k * (rb * A.bs + aqv * A.cs) - 2 * awv;
  }
  if constexpr (FROM_X) rb -= fwx;  // (exchange_unpack: y = .. - the forward values of all ranks' subtrees)
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = (j < lim) ? a[j] : 0.0;
  // padding pivots: unit diagonal (set here, ahead of the wait for the descendants -- no pulled entry is a padding
  // one, the LDS image below carries them along -- instead of behind it, where every instruction is on the
  // tree's critical path)
#pragma unroll
  for (int j = 0; j < NSMAX; j++)
    if (j >= ns && lane == j) a[j] = 1.0;
#pragma unroll
  for (int c = 0; c < SMAX; c++) a[NSMAX + c] = 0.0;
  rv[0] = is_row ? (NRHS == 1 ? rb : rb * A.bs) : 0.0;
  if constexpr (NRHS == 3) {
    rv[1] = is_row ? aqv * A.cs : 0.0;
    rv[2] = is_row ? awv : 0.0;
  }
  FT_STAMP(2);  // panel and right-hand side assembled (own block; further sources still to add)

  const bool pulls = ntg > 0 || R.mf > 0;
  if (nxt > 0 || ntg > 0) {
    // the register-shaped LDS image (my[64 j + lane]): further sources are added, pulled Schur
    // updates subtracted, each in its list's order
#pragma unroll
    for (int j = 0; j < NSMAX; j++) my[64 * j + lane] = a[j];
    WaveSync();
    if (nxt > 0) {
      if (lane < nxt) {
        double acc = my[xloc0];
#pragma unroll
        for (int i = 0; i < XMAX; i++) acc += gx0[i];  // (absent sources add +0.0: exact)
        my[xloc0] = acc;
      }
      if (lane + 64 < nxt) {
        double acc = my[xloc1];
#pragma unroll
        for (int i = 0; i < XMAX; i++) acc += gx1[i];
        my[xloc1] = acc;
      }
      // (an entry with more than XMAX further sources: the rest of its list, XMAX at a time)
      for (int c0 = XMAX; c0 < mx; c0 += XMAX)
        for (int half = 0; half < 2 && 64 * half < nxt; half++) {
          const int t = lane + 64 * half;
          const int ts = t < nxt ? t : 0;
          long long xs[XMAX];
          double gv[XMAX];
#pragma unroll
          for (int i = 0; i < XMAX; i++) xs[i] = A.xsrc[xbase + (int64_t)ts * mx + (c0 + i < mx ? c0 + i : mx - 1)];
#pragma unroll
          for (int i = 0; i < XMAX; i++) gv[i] = A.G[xs[i] >= 0 ? xs[i] : 0];
          const int xloc = half ? xloc1 : xloc0;
          double acc = my[xloc];
#pragma unroll
          for (int i = 0; i < XMAX; i++) acc += (c0 + i < mx && xs[i] >= 0) ? gv[i] : 0.0;
          if (t < nxt) my[xloc] = acc;
        }
      WaveSync();
    }
  }
  if (pulls) {
    // ---- wait for the descendants: the values themselves are polled, each checked against the
    // sentinel, with TWO rounds of loads in flight (every load unconditional, unused slots re-read slot
    // 0 and are masked: the number of loads per round is static, so the compiler's waits are exact
    // and a round is examined while the next is on its way).  A value is then in registers one
    // memory latency after it became visible -- the earlier form (one arrival word per publisher
    // polled first, then the values) paid two dependent round trips behind every level of the tree.
    double pv0[MMAX], pv1[MMAX], pb[MFMAX];
    double pbx[NRHS > 1 ? NRHS - 1 : 1][MFMAX];  // forward values of right-hand sides 1, 2
    const double* srcx = handT + A.updb_base + A.fwd_stride + R.fbase + (is_row ? lane : 0) * R.mf;  // (2: fwd_stride further)
    const double* src0 = handG + R.ubase + (int64_t)(lane < ntg ? lane : 0) * R.m;
    const double* src1 = handG + R.ubase + (int64_t)(lane + 64 < ntg ? lane + 64 : 0) * R.m;
    const double* srcb = handG + A.updb_base + R.fbase + (is_row ? lane : 0) * R.mf;
    // Lists longer than the MMAX slots a round holds (the interfaces near the top of a segmented chain
    // collect an update from every step that carried them: ~2 log2(segments)) are taken MMAX slots at
    // a time, in slot order -- the order of the subtractions is the list's either way.
    const int span = R.m > R.mf ? R.m : R.mf;
    SleepUnits(f(31) * A.up_sleep);  // (level x the least a level below can take: nothing to poll for before)
    for (int sb0 = 0; sb0 < (span > 0 ? span : 1); sb0 += MMAX) {
    const int mleft = R.m - sb0, fleft = R.mf - sb0;  // slots of this chunk in use: i < mleft / i < fleft
#pragma unroll
    for (int i = 0; i < MMAX; i++) pv1[i] = 0.0;
    if (ntg <= 64) {
      // a round asks for MM slots per entry and row: four where no list of this chunk is longer (C4: three
      // children share a parent's variable), the eight of MMAX otherwise -- half the loads per round
      auto poll = [&](auto mmc) {
        constexpr int MM = decltype(mmc)::value;
        double p0[MM], pf[MM], q0[MM], qb[MM];  // two rounds in flight
        double pfx[NRHS > 1 ? NRHS - 1 : 1][MM], qbx[NRHS > 1 ? NRHS - 1 : 1][MM];
        auto arrived = [&](const double (&v0)[MM], const double (&vb)[MM], const double (&vx)[NRHS > 1 ? NRHS - 1 : 1][MM]) {
          // (no short-circuit: a chain of || became a ladder of exec-mask branches, ~0.4 us of instruction
          // issue per round on this lone wavefront)
          const bool is_tg = lane < ntg;
          bool pending = false;
#pragma unroll
          for (int i = 0; i < MM; i++) pending |= (i < mleft) & is_tg & IsSentinel(v0[i]);
#pragma unroll
          for (int i = 0; i < MM; i++) pending |= (i < fleft) & is_row & IsSentinel(vb[i]);
          if constexpr (NRHS > 1) {
#pragma unroll
            for (int q = 0; q < NRHS - 1; q++)
#pragma unroll
              for (int i = 0; i < MM; i++) pending |= (i < fleft) & is_row & IsSentinel(vx[q][i]);
          }
          return __ballot(pending) == 0;
        };
#define CXK_FUSED_ISSUE(V0, VB, VX)                                                      \
  do {                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < MM; i_++) V0[i_] = LoadAgent(src0 + (i_ < mleft ? sb0 + i_ : 0));   \
    _Pragma("unroll") for (int i_ = 0; i_ < MM; i_++) VB[i_] = LoadAgent(srcb + (i_ < fleft ? sb0 + i_ : 0)); \
    if constexpr (NRHS > 1) {                                                            \
      _Pragma("unroll") for (int q_ = 0; q_ < NRHS - 1; q_++)                            \
        _Pragma("unroll") for (int i_ = 0; i_ < MM; i_++)                                \
          VX[q_][i_] = LoadAgent(srcx + q_ * A.fwd_stride + (i_ < fleft ? sb0 + i_ : 0)); \
    }                                                                                    \
  } while (0)
        CXK_FUSED_ISSUE(p0, pf, pfx);
        for (int spin = 0;; spin++) {
          CXK_FUSED_ISSUE(q0, qb, qbx);
          if (arrived(p0, pf, pfx)) {
            FT_COUNT(12, 2 * spin);
            break;
          }
          CXK_FUSED_ISSUE(p0, pf, pfx);
          if (arrived(q0, qb, qbx) || 2 * spin >= kFusedSpinLimit) {
            if (2 * spin >= kFusedSpinLimit) ReportTimeout(A);
#pragma unroll
            for (int i = 0; i < MM; i++) p0[i] = q0[i];
#pragma unroll
            for (int i = 0; i < MM; i++) pf[i] = qb[i];
            if constexpr (NRHS > 1) {
#pragma unroll
              for (int q = 0; q < NRHS - 1; q++)
#pragma unroll
                for (int i = 0; i < MM; i++) pfx[q][i] = qbx[q][i];
            }
            FT_COUNT(12, 2 * spin + 1);
            break;
          }
        }
#undef CXK_FUSED_ISSUE
#pragma unroll
        for (int i = 0; i < MMAX; i++) pv0[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MFMAX; i++) pb[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MM; i++) pv0[i] = i < mleft ? p0[i] : 0.0;
#pragma unroll
        for (int i = 0; i < MM; i++) pb[i] = i < fleft ? pf[i] : 0.0;
        if constexpr (NRHS > 1) {
#pragma unroll
          for (int q = 0; q < NRHS - 1; q++) {
#pragma unroll
            for (int i = 0; i < MFMAX; i++) pbx[q][i] = 0.0;
#pragma unroll
            for (int i = 0; i < MM; i++) pbx[q][i] = i < fleft ? pfx[q][i] : 0.0;
          }
        }
      };
      if (mleft <= 4 && fleft <= 4)
        poll(std::integral_constant<int, 4>{});
      else
        poll(std::integral_constant<int, MMAX>{});
    } else {
      // (more than 64 pulled entries: one round at a time)
      for (int spin = 0;; spin++) {
        bool pending = false;
#pragma unroll
        for (int i = 0; i < MMAX; i++) pv0[i] = pv1[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MFMAX; i++) pb[i] = 0.0;
#pragma unroll
        for (int i = 0; i < MMAX; i++)
          if (i < mleft) {
            pv0[i] = LoadAgent(src0 + sb0 + i);
            pv1[i] = LoadAgent(src1 + sb0 + i);
          }
#pragma unroll
        for (int i = 0; i < MFMAX; i++)
          if (i < fleft) pb[i] = LoadAgent(srcb + sb0 + i);
        if constexpr (NRHS > 1) {
#pragma unroll
          for (int q = 0; q < NRHS - 1; q++)
#pragma unroll
            for (int i = 0; i < MFMAX; i++) {
              pbx[q][i] = i < fleft ? LoadAgent(srcx + q * A.fwd_stride + sb0 + i) : 0.0;
              pending |= is_row & IsSentinel(pbx[q][i]);
            }
        }
#pragma unroll
        for (int i = 0; i < MMAX; i++) {
          pending |= (lane < ntg) & IsSentinel(pv0[i]);
          pending |= (lane + 64 < ntg) & IsSentinel(pv1[i]);
        }
#pragma unroll
        for (int i = 0; i < MFMAX; i++) pending |= is_row & IsSentinel(pb[i]);
        if (__ballot(pending) == 0) {
          FT_COUNT(12, spin);
          break;
        }
        if (spin >= kFusedSpinLimit) {
          ReportTimeout(A);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (ntg > 0) {
      if (lane < ntg) {
        double acc = my[ploc0];
#pragma unroll
        for (int i = 0; i < MMAX; i++) acc -= pv0[i];  // (slots not in use subtract 0.0: exact)
        my[ploc0] = acc;
      }
      if (lane + 64 < ntg) {
        double acc = my[ploc1];
#pragma unroll
        for (int i = 0; i < MMAX; i++) acc -= pv1[i];
        my[ploc1] = acc;
      }
      WaveSync();
    }
#pragma unroll
    for (int i = 0; i < MFMAX; i++) rv[0] -= is_row ? pb[i] : 0.0;
    if constexpr (NRHS > 1) {
#pragma unroll
      for (int q = 0; q < NRHS - 1; q++)
#pragma unroll
        for (int i = 0; i < MFMAX; i++) rv[1 + q] -= is_row ? pbx[q][i] : 0.0;
    }
    }  // chunks of MMAX slots
  }
  const bool imaged = nxt > 0 || ntg > 0;
  if (imaged) {
    if constexpr (ROWRHS) {
      // right-hand side q becomes row RL + q of the panel, entry j = lane j's value (zero beyond the columns): lane j
      // puts it where the image has that entry, and the read below hands the rows out with the panel's -- no pass of
      // its own behind the descendants' values (NSMAX reads and two waits per level of the tree, ~0.3 us each)
      if (lane < NSMAX) {
#pragma unroll
        for (int q = 0; q < NRHS; q++) my[64 * lane + RL + q] = rv[q];
      }
      WaveSync();
    }
#pragma unroll
    for (int j = 0; j < NSMAX; j++) a[j] = my[64 * j + lane];
    WaveSync();  // (the image is reused below)
  }
  if constexpr (ROWRHS) {
    if (!imaged) {  // (a leaf: no image so far)
#pragma unroll
      for (int q = 0; q < NRHS; q++) my[64 * q + lane] = rv[q];
      WaveSync();
      if (lane >= RL && lane < RL + NRHS) {
#pragma unroll
        for (int j = 0; j < NSMAX; j++) a[j] = my[64 * (lane - RL) + j];
      }
      WaveSync();  // (the image is reused below)
    }
  } else {
#pragma unroll
    for (int q = 0; q < NRHS; q++) a[RB + q] = rv[q];
  }
  FT_STAMP(3);  // descendants' values are in
  bool bad = false;
  if constexpr (NSMAX > 32) {
    // a supernode of 33 .. 64 columns without separator (the one dense block of a single-constraint
    // program: BASELINE config 2's 50 x 50): a column of L spans four 16-lane DPP rows
    static_assert(SMAX == 0, "the wide elimination has no separator rows");
    // every finished column goes to the slab and to the transposed LDS image at once: the issue time
    // of those stores hides behind the next pivot's dependent chain (after the loop they cost ~6 us)
    ElimWide<NSMAX, 0>::run(a, lane, ns, bad, [&](int j, double colj) {
      if (j < lim) base[o0 + j * st] = colj;
      my[65 * j + lane] = colj;
    });
  } else {
    ElimSteps<NSMAX, SMAX, 0, NCOL, !ROWRHS>::run(a, lane, bad, ns);  // (ROWRHS: the pivots are judged by the solution, below)
  }
  FT_STAMP(4);  // eliminated
  if (bad && lane == 0) atomicExch(A.fail + 1, A.tag);  // (carries on: everybody above must still drain)
  // ---- publish (the ancestors are waiting): the values sit in the separator lanes' registers --
  // value (k, c), c >= k, in a[NSMAX + c] of lane NSMAX + k -- and leave through LDS so that ONE
  // store instruction carries 64 of them (a store costs a lone wavefront ~70 cycles of issue)
  if constexpr (SMAX > 0) {
    if (s > 0) {
      // (every lane writes, no predicates: SMAX + 1 LDS stores, one read, one global store)
#pragma unroll
      for (int c = 0; c < SMAX; c++) my[(SMAX + NRHS) * lane + c] = a[NSMAX + c];
#pragma unroll
      for (int q = 0; q < NCOL; q++) my[(SMAX + NRHS) * lane + SMAX + q] = a[RB + q];
      WaveSync();
#pragma unroll
      for (int r = 0; r < PR; r++) {
        const double v = -my[prd[r]];
        if (lane + 64 * r < nv) StoreAgent((NRHS > 1 && pdt[r] ? handT : handG) + pd[r], v);
      }
      WaveSync();
    }
  }
  FT_STAMP(7);  // published
  if constexpr (UP_ONLY) {
#pragma unroll
    for (int j = 0; j < NSMAX; j++)
      if (j < lim) base[o0 + j * st] = a[j];
    if (is_row) {
      A.y[R.start + lane] = a[NCOL > 0 ? RB : 0];  // (UP_ONLY: the column form)
      A.AW[R.start + lane] = awv;
      A.AQc[R.start + lane] = aqv;
    }
    if constexpr (SMAX > 0) {
      if (s > 0) {
#pragma unroll
        for (int r = 0; r < PR; r++)
          if (lane + 64 * r < nv) StoreAgent(handO + pd[r], SentinelValue());
      }
    }
    return;
  }
  // ---- the way back down: rows of L become columns through an LDS image with an odd stride
  // (RootBackward's), lane i owns y_i; the separator's solution comes from the ancestors
  const int cnt = R.bs_end - R.bs_beg;
  const bool active = is_row;
  if constexpr (NSMAX <= 32) {
#pragma unroll
    for (int j = 0; j < NSMAX; j++) my[65 * j + lane] = a[j];
  }
  WaveSync();
  if (cnt > 0) {
    // (the root keeps its factor for after its back substitution: everybody waits for that)
#pragma unroll
    for (int j = 0; j < NSMAX; j++)
      if (j < lim) base[o0 + j * st] = a[j];
  }
  FT_STAMP(8);  // factor stored
  const int li = active ? lane : 0;
  double col[NSMAX];
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = my[65 * li + k];
  double dg = my[66 * li];
  FT_STAMP(9);  // columns of L back from the image
  constexpr int QN = SMAX < 8 ? SMAX : 8;
  double bv[QN > 0 ? QN : 1], yv[QN > 0 ? QN : 1], Mb[QN > 0 ? QN : 1];
  double yvx[NRHS > 1 ? NRHS - 1 : 1][QN > 0 ? QN : 1];
  bv[0] = Mb[0] = 0.0;
  // (the solution entries of the separator: zeros beyond its cnt variables, so that the combination below needs
  // no select per term -- the lanes beyond cnt are cleared once, before the values are handed out)
#pragma unroll
  for (int qq = 0; qq < (QN > 0 ? QN : 1); qq++) {
    yv[qq] = 0.0;
#pragma unroll
    for (int q = 0; q < (NRHS > 1 ? NRHS - 1 : 1); q++) yvx[q][qq] = 0.0;
  }
  if constexpr (QN > 0) {
#pragma unroll
    for (int qq = 0; qq < QN; qq++) {
      const unsigned sw = qq < cnt ? (unsigned)R.sep[qq] : 0u;
      bv[qq] = my[65 * li + NSMAX + (int)(sw >> 26)];
    }
  }
  // While the ancestors are still at work: the back substitution is linear in the separator's
  // solution, y = L^-T z - (L^-T B) y_sep, so u = L^-T z and the columns M = L^-T B are solved for
  // now (one sweep, 1 + cnt right-hand sides) and the arrival of y_sep is followed by cnt
  // multiply-adds instead of the NSMAX dependent steps of the sweep.  (Same solution to rounding;
  // BackwardSupernodeLean subtracts B y_sep first and sweeps once.)
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = (active && k > lane && k < ns) ? col[k] : 0.0;
  dg = active ? dg : 1.0;
  // the forward-solved right-hand side of this lane's row: in the column a[RB ..], or entry `lane` of row RL + q
  // (the image holds a_l[j] at 65 j + l)
  double ub, ubx[NRHS > 1 ? NRHS - 1 : 1];  // right-hand side 0; 1, 2
  ubx[0] = 0.0;
  if constexpr (ROWRHS) {
    ub = active ? my[65 * li + RL] : 0.0;
#pragma unroll
    for (int q = 0; q < NRHS - 1; q++) ubx[q] = active ? my[65 * li + RL + 1 + q] : 0.0;
  } else {
    ub = active ? a[NCOL > 0 ? RB : 0] : 0.0;
#pragma unroll
    for (int q = 0; q < NCOL - 1; q++) ubx[q] = active ? a[RB + 1 + q] : 0.0;
  }
  if constexpr (QN > 0) {
#pragma unroll
    for (int qq = 0; qq < QN; qq++) Mb[qq] = (qq < cnt && active) ? bv[qq] : 0.0;
  }
  {
    const double dinv = 1.0 / dg;
#pragma unroll
    for (int k = NSMAX - 1; k >= 0; k--) {
      if (lane == k) ub *= dinv;
      ub = fma(-col[k], ReadLane(ub, k), ub);  // col[k] is zero for lanes >= k
      if constexpr (NRHS > 1) {
#pragma unroll
        for (int q = 0; q < NRHS - 1; q++) {
          if (lane == k) ubx[q] *= dinv;
          ubx[q] = fma(-col[k], ReadLane(ubx[q], k), ubx[q]);
        }
      }
      if constexpr (QN > 0) {
        if (cnt > 0) {
#pragma unroll
          for (int qq = 0; qq < QN; qq++) {
            if (lane == k) Mb[qq] *= dinv;
            Mb[qq] = fma(-col[k], ReadLane(Mb[qq], k), Mb[qq]);
          }
        }
      }
    }
  }
  // (a supernode of at most 16 columns: its rows sit in DPP row 0, where the separator's solution -- entry qq in
  // lane qq of the polled register -- reaches every row as the broadcast operand of the multiply-add itself;
  // the negated columns of M wait in registers: two v_readlane per term less behind the arrival)
  constexpr bool ROW0 = NSMAX <= 16 && QN > 0;
  double vsep = 0.0, vsepx[NRHS > 1 ? NRHS - 1 : 1];
#pragma unroll
  for (int q = 0; q < (NRHS > 1 ? NRHS - 1 : 1); q++) vsepx[q] = 0.0;
  if constexpr (ROW0) {
#pragma unroll
    for (int qq = 0; qq < QN; qq++) {
      Mb[qq] = -Mb[qq];
      asm volatile("" : "+v"(Mb[qq]));  // (negated here, ahead of the wait: not in front of each multiply-add)
    }
  }
  if constexpr (QN > 0) {
    if (cnt > 0) {
      // lane qq < cnt polls the solution entry of separator variable qq
      const int sepw = __builtin_amdgcn_ds_bpermute(4 * (24 + (lane < 8 ? lane : 0)), w);
      const int sidx = lane < cnt ? (sepw & 0x3ffffff) : (R.sep[0] & 0x3ffffff);
      const double* src = ysG + sidx;
      double v;
      double vx[NRHS > 1 ? NRHS - 1 : 1];
      vx[0] = 0.0;
      for (int spin = 0;; spin++) {
        v = LoadAgent(src);
        bool pending = (lane < cnt) & IsSentinel(v);
        if constexpr (NRHS > 1) {
#pragma unroll
          for (int q = 0; q < NRHS - 1; q++) {
            vx[q] = LoadAgent(ysT + (int64_t)(1 + q) * A.y_stride + sidx);
            pending |= (lane < cnt) & IsSentinel(vx[q]);
          }
        }
        if (__ballot(pending) == 0) {
          FT_COUNT(14, spin);
          break;
        }
        if (spin >= kFusedSpinLimit) {
          ReportTimeout(A);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      v = lane < cnt ? v : 0.0;
      if constexpr (NRHS > 1) {
#pragma unroll
        for (int q = 0; q < NRHS - 1; q++) vx[q] = lane < cnt ? vx[q] : 0.0;
      }
      if constexpr (ROW0) {
        vsep = v;
        if constexpr (NRHS > 1) {
#pragma unroll
          for (int q = 0; q < NRHS - 1; q++) vsepx[q] = vx[q];
        }
      } else {
#pragma unroll
        for (int qq = 0; qq < QN; qq++) yv[qq] = ReadLane(v, qq);
        if constexpr (NRHS > 1) {
#pragma unroll
          for (int q = 0; q < NRHS - 1; q++)
#pragma unroll
            for (int qq = 0; qq < QN; qq++) yvx[q][qq] = ReadLane(vx[q], qq);
        }
      }
    }
  }
  FT_STAMP(5);  // the separator's solution is in
  double accx[NRHS > 1 ? NRHS - 1 : 1];
  accx[0] = 0.0;
  double acc = ub;
  if constexpr (ROW0) {
    // acc = fma(y_sep[qq], -M[qq], acc), qq ascending: the products and the order of the sums of the other form
    if constexpr (NRHS > 1) {
      static_assert(NRHS == 3, "two further right-hand sides");
      accx[0] = ubx[0];
      accx[1] = ubx[1];
      DppOperandFence(vsep, vsepx[0], vsepx[1]);
    } else {
      double d0 = 0.0, d1 = 0.0;
      DppOperandFence(vsep, d0, d1);
    }
    auto chain = [&](auto self, auto qc) {
      constexpr int qq = decltype(qc)::value;
      if constexpr (qq < QN) {
        FmacRowBcast<qq>(acc, vsep, Mb[qq]);
        if constexpr (NRHS > 1) {
          FmacRowBcast<qq>(accx[0], vsepx[0], Mb[qq]);
          FmacRowBcast<qq>(accx[1], vsepx[1], Mb[qq]);
        }
        self(self, std::integral_constant<int, qq + 1>{});
      }
    };
    chain(chain, std::integral_constant<int, 0>{});
  } else {
    if constexpr (NRHS > 1 && QN > 0) {
#pragma unroll
      for (int q = 0; q < NRHS - 1; q++) {
        accx[q] = ubx[q];
#pragma unroll
        for (int qq = 0; qq < QN; qq++) accx[q] = fma(-Mb[qq], yvx[q][qq], accx[q]);
      }
    } else if constexpr (NRHS > 1) {
#pragma unroll
      for (int q = 0; q < NRHS - 1; q++) accx[q] = ubx[q];
    }
    if constexpr (QN > 0) {
#pragma unroll
      for (int qq = 0; qq < QN; qq++) acc = fma(-Mb[qq], yv[qq], acc);
    }
  }
  if (active) StoreAgent(ysG + R.start + lane, acc);
  if constexpr (NRHS > 1) {
#pragma unroll
    for (int q = 0; q < NRHS - 1; q++)
      if (active) StoreAgent(ysT + (int64_t)(1 + q) * A.y_stride + R.start + lane, accx[q]);
  }
  FT_STAMP(6);
  if constexpr (ROWRHS) {
    // a pivot that was not positive (here or in a descendant) has left NaNs in everything behind it: the test of
    // the factorization, two instructions per pivot inside the elimination, is one comparison out here (behind
    // the stores the descendants wait for)
    if (__ballot(active && !(acc == acc)) != 0 && lane == 0) atomicExch(A.fail + 1, A.tag);
  }
  // ---- nobody waits for the rest: the solution and AW / AQc for the kernels that follow, the
  // root's factor, the re-armed slots of the other set
  if (active) {
    if constexpr (NRHS == 3) {
      // K^-1 (bs b), K^-1 (cs AQc), K^-1 AW for whoever forms the Newton direction, and y = -first + second:
      // the solution for the right-hand side of the mu selection (-bs b + cs AQc)
      A.y3[R.start + lane] = acc;
      A.y3[A.y_stride + R.start + lane] = accx[0];
      A.y3[2 * A.y_stride + R.start + lane] = accx[1];
      A.y[R.start + lane] = accx[0] - acc;
      StoreAgent(ysTO + A.y_stride + R.start + lane, SentinelValue());
      StoreAgent(ysTO + 2 * A.y_stride + R.start + lane, SentinelValue());
    } else {
      A.y[R.start + lane] = acc;
    }
    A.AW[R.start + lane] = awv;
    A.AQc[R.start + lane] = aqv;
    StoreAgent(ysO + R.start + lane, SentinelValue());
  }
  if (NSMAX <= 32 && cnt == 0) {
#pragma unroll
    for (int j = 0; j < NSMAX; j++)
      if (j < lim) base[o0 + j * st] = a[j];
  }
  if constexpr (SMAX > 0) {
    if (s > 0) {
#pragma unroll
      for (int r = 0; r < PR; r++)
        if (lane + 64 * r < nv) StoreAgent((NRHS > 1 && pdt[r] ? handTO : handO) + pd[r], SentinelValue());
    }
  }
  FT_STAMP_FLUSH(f(31));
}

// The solve-only sweep on a stored factor (mu selection, Newton direction after it, line search,
// refinement corrections): forward substitution up the tree, back substitution down, one launch,
// the same hand-off slots (forward values, arrival words, solution entries).  Arithmetic and its
// order are ForwardSupernodeLean's and BackwardSupernodeLean's.  The rows of L (forward) and its
// columns (backward) both come straight from the slab.
// PHASE 0: forward and back substitution; 1: forward only (the forward-solved right-hand side goes
// to y); 2: back substitution only (reads it from there): the two launches of a tree too large to
// be resident at once, see FusedSupernode.
template <int NSMAX, int SMAX, int PHASE = 0>
__device__ __forceinline__ void FusedSolveSupernode(const FusedTreeArgs& A, const int w) {
  static_assert(NSMAX + SMAX <= 64, "one lane per panel row");
  constexpr int MFMAX = kFastSlots;
  constexpr int NVMAX = SMAX * (SMAX + 1) / 2 + SMAX;
  constexpr int PR = (NVMAX + 63) / 64;
  const int lane = threadIdx.x & 63;
  const SnRec R = DecodeRec(w);
  const int ns = R.ns, s = R.nsep;
  const bool is_row = lane < ns;
  const int sc = lane - NSMAX;
  const bool is_sep = sc >= 0 && sc < s;
  const double* base = A.slab + R.diag_off;
  const unsigned rel = (unsigned)(R.offd_off - R.diag_off);
  const unsigned o0 = is_row ? (unsigned)lane : (is_sep ? rel + (unsigned)(sc * ns) : 0u);
  const unsigned st = is_row ? (unsigned)ns : 1u;
  const int lim = is_row ? lane : (is_sep ? ns : 0);  // strictly lower part of a row; a whole off column
  const int gen = A.gen;
  double* handG = A.hand + (int64_t)gen * A.hand_stride;
  double* handO = A.hand + (int64_t)(gen ^ 1) * A.hand_stride;
  double* ysG = A.ysig + (int64_t)gen * A.ysig_stride;
  double* ysO = A.ysig + (int64_t)(gen ^ 1) * A.ysig_stride;
  const int pub_beg = __builtin_amdgcn_readlane(w, 23);
  const int npairs = s * (s + 1) / 2, nv = npairs + s;
  const int cnt = R.bs_end - R.bs_beg;
  // ---- one trip: rows of L, right-hand side, columns of L, off-block entries, destinations
  double a[NSMAX > 0 ? NSMAX : 1];
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = base[(j < lim) ? o0 + j * st : 0u];
  double dg = base[is_row ? (unsigned)lane * (unsigned)(ns + 1) : 0u];
  const int p = R.start + (is_row ? lane : 0);
  double b;
  if (PHASE == 2 || A.form == 0) {
    b = A.y[p];
  } else {
    const double bp = A.b[p], aq = A.AQc[p], aw = A.AW[p];
    const double kk = (A.form == 1 && A.k_from) ? A.k_from[0] : A.k;
    b = A.form == 1 ? kk * (bp * A.bs + aq * A.cs) - 2 * aw : A.cb * bp + A.cq * aq + A.cw * aw;
  }
  const bool active = is_row;
  const double* D = A.slab + R.diag_off + (size_t)(active ? lane : 0) * ns;  // column `lane`
  const double* B = A.slab + R.offd_off + (active ? lane : 0);
  double col[NSMAX];
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = D[(active && k > lane && k < ns) ? k : 0];
  constexpr int QN = SMAX < 8 ? SMAX : 8;
  double bv[QN > 0 ? QN : 1], yv[QN > 0 ? QN : 1];
  bv[0] = yv[0] = 0.0;
#pragma unroll
  for (int qq = 0; qq < QN; qq++) {
    const unsigned sw = qq < cnt ? (unsigned)R.sep[qq] : 0u;
    const double* src = qq < cnt ? B + (size_t)(sw >> 26) * ns : D;
    bv[qq] = src[0];
  }
  int pdb = 0;
  if constexpr (SMAX > 0) pdb = A.pub[pub_beg + npairs + (is_sep ? sc : 0)];
  // ---- wait for the descendants' forward values (the values themselves are polled, as in the factor
  // sweep; lists longer than MFMAX slots MFMAX at a time, in slot order) and subtract them
  if constexpr (PHASE != 2) b = is_row ? b : 0.0;
  if (PHASE != 2 && R.mf > 0) {
    // (two rounds of loads in flight, every load unconditional: see the factor sweep)
    const double* srcb = handG + A.updb_base + R.fbase + (is_row ? lane : 0) * R.mf;
    for (int sb0 = 0; sb0 < R.mf; sb0 += MFMAX) {
      const int fleft = R.mf - sb0;
      double pb[MFMAX], qb[MFMAX];
      auto arrived = [&](const double (&vb)[MFMAX]) {
        bool pending = false;
#pragma unroll
        for (int i = 0; i < MFMAX; i++) pending |= (i < fleft) & is_row & IsSentinel(vb[i]);  // (no short-circuit: FusedSupernode)
        return __ballot(pending) == 0;
      };
#define CXK_FUSED_ISSUE(VB) \
  _Pragma("unroll") for (int i_ = 0; i_ < MFMAX; i_++) VB[i_] = LoadAgent(srcb + (i_ < fleft ? sb0 + i_ : 0))
      CXK_FUSED_ISSUE(pb);
      for (int spin = 0;; spin++) {
        CXK_FUSED_ISSUE(qb);
        if (arrived(pb)) break;
        CXK_FUSED_ISSUE(pb);
        if (arrived(qb) || 2 * spin >= kFusedSpinLimit) {
          if (2 * spin >= kFusedSpinLimit) ReportTimeout(A);
#pragma unroll
          for (int i = 0; i < MFMAX; i++) pb[i] = qb[i];
          break;
        }
      }
#undef CXK_FUSED_ISSUE
#pragma unroll
      for (int i = 0; i < MFMAX; i++) b -= (is_row && i < fleft) ? pb[i] : 0.0;
    }
  }
  // ---- forward substitution (ForwardSupernodeLean)
  if constexpr (PHASE != 2) {
#pragma unroll
  for (int j = 0; j < NSMAX; j++) a[j] = (j < lim) ? a[j] : 0.0;
  const double dinvf = is_row ? 1.0 / dg : 0.0;
  double dot = 0.0;
#pragma unroll
  for (int k = 0; k < NSMAX; k++) {
    if (lane == k) b *= dinvf;
    const double bk = ReadLane(b, k);  // 0.0 for padding rows k >= ns
    if (is_sep)
      dot = fma(a[k], bk, dot);
    else
      b -= a[k] * bk;  // a[k] is zero for lanes <= k
  }
  if constexpr (SMAX > 0) {
    if (s > 0) {
      if (is_sep) StoreAgent(handG + pdb, dot);
    }
  }
  }  // PHASE != 2
  if constexpr (PHASE == 1) {
    if (is_row) A.y[R.start + lane] = b;
  }
  // ---- back substitution (BackwardSupernodeLean)
  if constexpr (PHASE != 1) {
  if constexpr (QN > 0) {
    if (cnt > 0) {
      const int sepw = __builtin_amdgcn_ds_bpermute(4 * (24 + (lane < 8 ? lane : 0)), w);
      const double* src = ysG + (lane < cnt ? (sepw & 0x3ffffff) : (R.sep[0] & 0x3ffffff));
      double v;
      for (int spin = 0;; spin++) {
        v = LoadAgent(src);
        if (__ballot(lane < cnt && IsSentinel(v)) == 0) break;
        if (spin >= kFusedSpinLimit) {
          ReportTimeout(A);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int qq = 0; qq < QN; qq++) yv[qq] = ReadLane(v, qq);
    }
  }
#pragma unroll
  for (int k = 0; k < NSMAX; k++) col[k] = (active && k > lane && k < ns) ? col[k] : 0.0;
  dg = active ? dg : 1.0;
  double acc = active ? b : 0.0;
  if constexpr (QN > 0) {
#pragma unroll
    for (int qq = 0; qq < QN; qq++) acc -= ((qq < cnt && active) ? bv[qq] : 0.0) * (qq < cnt ? yv[qq] : 0.0);
  }
  const double dinv = 1.0 / dg;
#pragma unroll
  for (int k = NSMAX - 1; k >= 0; k--) {
    if (lane == k) acc *= dinv;
    acc = fma(-col[k], ReadLane(acc, k), acc);  // col[k] is zero for lanes >= k
  }
  if (active) {
    StoreAgent(ysG + R.start + lane, acc);
    A.y[R.start + lane] = acc;
    StoreAgent(ysO + R.start + lane, SentinelValue());
  }
  }  // PHASE != 1
  // re-arm the other set: ALL of this supernode's slots, the Schur-update ones too (a factor sweep
  // may be the next run)
  if constexpr (SMAX > 0 && PHASE != 2) {
    if (s > 0) {
#pragma unroll
      for (int r = 0; r < PR; r++) {
        const int t = lane + 64 * r;
        const int d = A.pub[pub_beg + (t < nv ? t : 0)];
        if (t < nv) StoreAgent(handO + d, SentinelValue());
      }
    }
  }
}

// <w,c> and <c,Qc> as assemble_gather's workgroup 0 sums them (256 threads there: four wavefronts'
// strided partial sums, wave totals added in wave order), and the failure flag's reset.
__device__ __forceinline__ void FusedScalars(const FusedTreeArgs& A) {
  const int lane = threadIdx.x & 63;
  double t0 = 0, t1 = 0;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    double s0 = 0, s1 = 0;
    for (int i = 64 * v + lane; i < A.K; i += 256) {
      s0 += A.sc[2 * i];
      s1 += A.sc[2 * i + 1];
    }
    t0 += WaveSum(s0);
    t1 += WaveSum(s1);
  }
  if (lane == 0) {
    A.sys_sc[0] = t0;
    A.sys_sc[1] = t1;
    *A.fail = 0;
  }
}

// MODE 0: the whole sweep (assembly + factor + solve); 1: its upward half only (two launches: trees
// too large to be resident at once); 2: the whole sweep with three right-hand sides (kFusedTriple)
template <int NA, int SA, int NB, int SB, int MODE>
__global__ void __launch_bounds__(64, 2) tree_fused(FusedTreeArgs A) {
  extern __shared__ double lds[];
  const int pos = blockIdx.x;
  if (pos >= A.count) {
    FusedScalars(A);
    return;
  }
  const int w = A.rec[(size_t)pos * kFusedRecWords + (threadIdx.x & 63)];
  const int ns = __builtin_amdgcn_readlane(w, 1), s = __builtin_amdgcn_readlane(w, 2);
  constexpr int NRHS = MODE == 2 ? 3 : 1;
  if (NA == NB && SA == SB) {
    FusedSupernode<NA, SA, MODE == 1, false, NRHS>(A, w, lds);
  } else if (RegisterShape(ns, s) == (NA << 8 | SA)) {
    FusedSupernode<NA, SA, MODE == 1, false, NRHS>(A, w, lds);
  } else {
    FusedSupernode<NB, SB, MODE == 1, false, NRHS>(A, w, lds);
  }
}

// ---- sharded contexts ---------------------------------------------------------------------
// A value published by one of this rank's supernodes, waited for like every hand-off value.
__device__ __forceinline__ double WaitValue(const FusedTreeArgs& A, const double* p) {
  double v = LoadAgent(p);
  for (int spin = 0; IsSentinel(v); spin++) {
    if (spin >= kFusedSpinLimit) {
      atomicExch(A.fail + 1, A.tag);
      __hip_atomic_store(A.host_flag, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    v = LoadAgent(p);
  }
  return v;
}

// exchange_pack (kernels_kkt.hip.h) riding in the up launch, one lane per buffer entry: the partial
// assembled value of a top entry -- own-rank sources in the gather's order, what assemble_gather
// would have left in the slab -- minus the Schur updates this rank's subtrees publish into it (summed
// in slot order, then subtracted); per top variable the partial AW / AQc and the sum of the
// published forward values.
__device__ __forceinline__ void ShardPackItem(const FusedTreeArgs& A, int64_t item) {
  const double* handG = A.hand + (int64_t)A.gen * A.hand_stride;
  if (item < A.n_xs) {
    const GatherRec g = A.xg[item];
    double s = 0;
    if (g.first >= 0) s += A.G[g.first];
    for (int k = g.beg; k < g.beg + g.extra; k++) {
      const int64_t q = A.as_src[k];
      if (q >= 0) s += A.G[q];
    }
    const int t = A.xs_pt[item];
    if (t >= 0) {
      double u = 0;
      for (int q = A.pt_ptr[t]; q < A.pt_ptr[t + 1]; q++) u += WaitValue(A, handG + A.pt_src[q]);
      s -= u;
    }
    A.x[item] = s;
  } else if (item < A.n_xs + A.n_xv) {
    const int64_t j = item - A.n_xs;
    const ResidRec r = A.xr[j];
    double aw = 0, aq = 0;
    if (r.first >= 0) {
      aw += A.AWc[r.first];
      aq += A.AQcc[r.first];
    }
    for (int k = r.beg; k < r.beg + r.extra; k++) {
      aw += A.AWc[A.rs_src[k]];
      aq += A.AQcc[A.rs_src[k]];
    }
    double fw = 0;
    for (int q = A.pf_ptr[j]; q < A.pf_ptr[j + 1]; q++) fw += WaitValue(A, handG + A.updb_base + A.pf_src[q]);
    A.x[A.n_xs + j] = aw;
    A.x[A.n_xs + A.n_xv + j] = aq;
    A.x[A.n_xs + 2 * (int64_t)A.n_xv + j] = fw;
  }
}

// The buffer's tail: this rank's part of <w,c> and <c,Qc> (FusedScalars' sums; constraints of other
// ranks hold zeros) and the failure flag -- read once every supernode of the launch has counted
// itself done, so that a failed pivot anywhere in this rank's subtrees travels with the exchange.
__device__ __forceinline__ void ShardPackTail(const FusedTreeArgs& A) {
  FusedScalars(A);
  const int lane = threadIdx.x & 63;
  {
    // lane l waits for counter l: done_target launches x the supernodes at positions l (mod 64)
    const unsigned long long mine = lane < A.count_up ? (unsigned long long)((A.count_up - lane + 63) / 64) : 0ull;
    const unsigned long long want = A.done_target * mine;
    for (int spin = 0;; spin++) {
      const unsigned long long have = __hip_atomic_load(A.done + 16 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__ballot(have < want) == 0) break;
      if (spin >= kFusedSpinLimit) {
        if (lane == 0) {
          atomicExch(A.fail + 1, A.tag);
          __hip_atomic_store(A.host_flag, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  if (lane == 0) {
    const int64_t o = A.n_xs + 3 * (int64_t)A.n_xv;
    const int f0 = __hip_atomic_load(A.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int f1 = __hip_atomic_load(A.fail + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    A.x[o] = A.sys_sc[0];
    A.x[o + 1] = A.sys_sc[1];
    A.x[o + 2] = (f0 != 0 || (A.tag != 0 && f1 == A.tag)) ? 1.0 : 0.0;
    A.x[o + 3] = 0;
  }
}

// kFusedShardUp: workgroups [0, count_up) the rank's own subtrees (UP_ONLY: factor, forward-solved
// right-hand side and AW / AQc go to memory), then the pack of the exchange buffer (64 entries per
// workgroup), then its tail.
template <int NA, int SA, int NB, int SB>
__global__ void __launch_bounds__(64) tree_fused_shard_up(FusedTreeArgs A) {
  extern __shared__ double lds[];
  const int pos = blockIdx.x;
  if (pos < A.count_up) {
    const int w = A.rec[(size_t)pos * kFusedRecWords + (threadIdx.x & 63)];
    const int ns = __builtin_amdgcn_readlane(w, 1), s = __builtin_amdgcn_readlane(w, 2);
    if (NA == NB && SA == SB) {
      FusedSupernode<NA, SA, true>(A, w, lds);
    } else if (RegisterShape(ns, s) == (NA << 8 | SA)) {
      FusedSupernode<NA, SA, true>(A, w, lds);
    } else {
      FusedSupernode<NB, SB, true>(A, w, lds);
    }
    // Behind this wavefront's failure report, if any (an atomic too: the wait orders the two).  64
    // counters on lines of their own, position modulo 64: one counter would serialise its adders at
    // ~12 ns each (2300 supernodes of a rank of config 5: 28 us behind the last elimination), and an
    // agent-scope RELEASE here would write the XCD's L2 back once per wavefront.
    if ((threadIdx.x & 63) == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __hip_atomic_fetch_add(A.done + 16 * (pos & 63), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int64_t items = A.n_xs + A.n_xv;
  const int64_t first = (int64_t)(pos - A.count_up) * 64;
  if (first < items) {
    ShardPackItem(A, first + (threadIdx.x & 63));
    return;
  }
  ShardPackTail(A);
}

// kFusedShardTop: the top supernodes first (all resident: they wait for each other in both
// directions), then this rank's subtrees root side first (a supernode waits for its ancestors only),
// then the buffer's tail (scalars, failure flag).
template <int NA, int SA, int NB, int SB>
__global__ void __launch_bounds__(64) tree_fused_shard_top(FusedTreeArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, ntop = A.count - A.count_up;
  if (b >= A.count) {
    if ((threadIdx.x & 63) == 0) {
      const int64_t o = A.n_xs + 3 * (int64_t)A.n_xv;
      A.sys_sc[0] = A.x[o];
      A.sys_sc[1] = A.x[o + 1];
      if (A.x[o + 2] > 0.0) *A.fail = 1;
    }
    return;
  }
  const int pos = b < ntop ? A.count_up + b : A.count - 1 - b;
  const int w = A.rec[(size_t)pos * kFusedRecWords + (threadIdx.x & 63)];
  const int ns = __builtin_amdgcn_readlane(w, 1), s = __builtin_amdgcn_readlane(w, 2);
  const bool isA = (NA == NB && SA == SB) || RegisterShape(ns, s) == (NA << 8 | SA);
  if (b < ntop) {
    if (isA)
      FusedSupernode<NA, SA, false, true>(A, w, lds);
    else
      FusedSupernode<NB, SB, false, true>(A, w, lds);
  } else {
    if (isA)
      FusedSolveSupernode<NA, SA, 2>(A, w);
    else
      FusedSolveSupernode<NB, SB, 2>(A, w);
  }
}

// PHASE 0: forward + back substitution in one launch; 1: forward only; 2: back substitution only,
// workgroups in REVERSE position order (the root first: a supernode waits for its ancestors)
template <int NA, int SA, int NB, int SB, int PHASE>
__global__ void __launch_bounds__(64) tree_fused_solve(FusedTreeArgs A) {
  const int pos = PHASE == 2 ? A.count - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int w = A.rec[(size_t)pos * kFusedRecWords + (threadIdx.x & 63)];
  const int ns = __builtin_amdgcn_readlane(w, 1), s = __builtin_amdgcn_readlane(w, 2);
  if (NA == NB && SA == SB) {
    FusedSolveSupernode<NA, SA, PHASE>(A, w);
  } else if (RegisterShape(ns, s) == (NA << 8 | SA)) {
    FusedSolveSupernode<NA, SA, PHASE>(A, w);
  } else {
    FusedSolveSupernode<NB, SB, PHASE>(A, w);
  }
}

namespace {
// doubles of LDS one wavefront needs: the transposed image of L (65 NSMAX) / the publish scratch (64 (SMAX + 1))
constexpr int FusedImage(int nsmax, int smax) { return 65 * nsmax > 64 * (smax + 3) ? 65 * nsmax : 64 * (smax + 3); }  // (+ 3: kFusedTriple)

struct FusedKernels {
  const void* k[8];  // FusedTreeMode order
  int image;         // doubles of LDS of the factor sweeps
};

template <int NA, int SA, int NB, int SB>
FusedKernels KernelsOf() {
  FusedKernels f;
  f.k[kFusedFull] = reinterpret_cast<const void*>(&tree_fused<NA, SA, NB, SB, 0>);
  f.k[kFusedSolve] = reinterpret_cast<const void*>(&tree_fused_solve<NA, SA, NB, SB, 0>);
  f.k[kFusedUp] = reinterpret_cast<const void*>(&tree_fused<NA, SA, NB, SB, 1>);
  f.k[kFusedForward] = reinterpret_cast<const void*>(&tree_fused_solve<NA, SA, NB, SB, 1>);
  f.k[kFusedDown] = reinterpret_cast<const void*>(&tree_fused_solve<NA, SA, NB, SB, 2>);
  f.k[kFusedShardUp] = nullptr;
  f.k[kFusedShardTop] = nullptr;
  f.k[kFusedTriple] = nullptr;
  if constexpr (NA <= 32 && NB <= 32) f.k[kFusedTriple] = reinterpret_cast<const void*>(&tree_fused<NA, SA, NB, SB, 2>);
  if constexpr (NA <= 32 && NB <= 32) {  // (the wide single-supernode instances are single-GPU)
    f.k[kFusedShardUp] = reinterpret_cast<const void*>(&tree_fused_shard_up<NA, SA, NB, SB>);
    f.k[kFusedShardTop] = reinterpret_cast<const void*>(&tree_fused_shard_top<NA, SA, NB, SB>);
  }
  f.image = FusedImage(NA, SA) > FusedImage(NB, SB) ? FusedImage(NA, SA) : FusedImage(NB, SB);
  return f;
}

bool ForPair(int sa, int sb, FusedKernels* out) {
#define CXK_FUSED_PAIR(NA_, SA_, NB_, SB_)                          \
  if (sa == ((NA_) << 8 | (SA_)) && sb == ((NB_) << 8 | (SB_))) {   \
    if (out) *out = KernelsOf<NA_, SA_, NB_, SB_>();                \
    return true;                                                    \
  }
  CXK_FUSED_PAIR(8, 8, 8, 8)
  CXK_FUSED_PAIR(16, 8, 16, 8)
  CXK_FUSED_PAIR(24, 0, 24, 0)
  CXK_FUSED_PAIR(24, 8, 24, 8)
  CXK_FUSED_PAIR(32, 16, 32, 16)
  CXK_FUSED_PAIR(8, 8, 16, 8)
  CXK_FUSED_PAIR(8, 8, 24, 0)
  CXK_FUSED_PAIR(8, 8, 24, 8)
  CXK_FUSED_PAIR(8, 8, 32, 16)
  CXK_FUSED_PAIR(16, 8, 24, 0)
  CXK_FUSED_PAIR(16, 8, 24, 8)
  CXK_FUSED_PAIR(16, 8, 32, 16)
  CXK_FUSED_PAIR(24, 0, 24, 8)
  CXK_FUSED_PAIR(24, 0, 32, 16)
  CXK_FUSED_PAIR(24, 8, 32, 16)
  CXK_FUSED_PAIR(40, 0, 40, 0)
  CXK_FUSED_PAIR(48, 0, 48, 0)
  CXK_FUSED_PAIR(56, 0, 56, 0)
  CXK_FUSED_PAIR(64, 0, 64, 0)
#undef CXK_FUSED_PAIR
  return false;
}
}  // namespace

bool FusedTreeCompiled(int sa, int sb) { return ForPair(sa, sb, nullptr); }

int FusedTreeOccupancy(int sa, int sb, bool sharded) {
  FusedKernels f;
  if (!ForPair(sa, sb, &f)) return 0;
  int nb = 0, nbs = 0;
  if (sharded) {
    if (!f.k[kFusedShardTop]) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f.k[kFusedShardTop], 64, sizeof(double) * (size_t)f.image) != hipSuccess) return 0;
    return nb;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f.k[kFusedFull], 64, sizeof(double) * (size_t)f.image) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbs, f.k[kFusedSolve], 64, 0) != hipSuccess) return 0;
  return nb < nbs ? nb : nbs;
}

hipError_t LaunchFusedTree(const FusedTreeArgs& a, int sa, int sb, int mode, hipStream_t stream, hipEvent_t ev_start,
                           hipEvent_t ev_stop) {
  FusedKernels f;
  if (mode < 0 || mode > kFusedTriple || !ForPair(sa, sb, &f) || !f.k[mode]) return hipErrorInvalidValue;
  const bool factor = mode == kFusedFull || mode == kFusedUp || mode == kFusedShardUp || mode == kFusedShardTop || mode == kFusedTriple;
  FusedTreeArgs args = a;
  void* params[] = {&args};
  // (the factor sweeps' extra workgroup sums the two scalars; the sharded up launch carries the pack of
  // the exchange buffer -- 64 entries per workgroup -- and its tail behind the supernodes)
  int nwg = a.count + (factor ? 1 : 0);
  if (mode == kFusedShardUp) nwg = a.count_up + (int)((a.n_xs + a.n_xv + 63) / 64) + 1;
  const dim3 grid(nwg);
  const size_t lds = factor ? sizeof(double) * (size_t)f.image : 0;
  if (ev_start && ev_stop) return hipExtLaunchKernel(f.k[mode], grid, dim3(64), params, lds, stream, ev_start, ev_stop, 0);
  return hipLaunchKernel(f.k[mode], grid, dim3(64), params, lds, stream);
}

}  // namespace cxk

#ifdef CXK_FUSED_STAMPS
extern "C" int cxk_debug_fused_tree_stamps(long long* out, int waves) {
  if (waves > cxk::kFusedStampWaves) waves = cxk::kFusedStampWaves;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cxk::g_fused_tree_stamp), sizeof(long long) * 16 * (size_t)waves) == hipSuccess ? 0 : 1;
}
#endif
