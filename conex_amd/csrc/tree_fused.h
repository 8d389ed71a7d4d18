// The whole elimination tree in ONE launch: assembly gather, supernodal Cholesky with the forward
// substitution riding in it, and the back substitution (tree_fused.hip).  Host-side interface.
//
// Reference semantics: SupernodalKKTSolver::Assemble / Factor / SolveInPlace
// (kkt_solver.cc:164-170, 180-193, 220-263), BlockCholeskyInPlace and the block solves
// (block_triangular_operations.cc:114-219), AssembleSchurComplementResiduals
// (constraint_manager.h:107-124), the right-hand side of cone_program.cc:409-411 / :181.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cxk {

// A published value doubles as its own "ready" flag: slots start as this bit pattern (a signalling
// NaN no arithmetic produces) and the consumer polls until it is gone.
constexpr unsigned long long kFusedSentinel = 0x7FF4C0DEC0DE5EEDull;

constexpr int kFusedRecWords = 64;   // one record per supernode: 256 bytes, one dword per lane
constexpr int kFusedExtraTargets = 128;  // panel entries with more than one source (two per lane)
constexpr int kFusedExtraSlots = 8;      // further sources per such entry / sources per shared variable fetched at once
constexpr int kFusedExtraMax = 64;       // ... at most (longer lists: the level kernels)
constexpr int kFusedMaxSlots = 4096;     // published values one entry / one row pulls, at most (taken kFastSlots at a time)

// Words of a record (position = dependency order: a supernode only waits for lower positions):
//   [0, 32)   SnRec (kernels_kkt.hip.h); its spare word 23: pub_beg
//   [32, 56)  AsmRec: the Schur block of the supernode's own constraint and the position of every
//             panel row in it.  A supernode of the replicated top of a sharded context (its panel
//             comes from the exchange buffer): 32,33 xs_base, 34 xv_base -- where its slab entries
//             and its variables start in the buffer
//   56 xt_beg  57 nxt  58 mx   panel entries with further sources: image location xreg[xt_beg + t],
//   61,62 xbase                sources xsrc[xbase + t * mx + i] (index into G, -1 none), gather order
//   59 rbase   60 mr           variables shared with other constraints: ALL their sources in gather
//                              order rsrc[rbase + row * mr + i] (index into AWc / AQcc, -1 none); a
//                              row whose list is empty has its own constraint as the one source
struct GatherRec;
struct ResidRec;

struct FusedTreeArgs {
  const int* rec;
  int count;  // supernodes = workgroups (one more workgroup sums the two scalars)
  const double *G, *AWc, *AQcc, *b;
  double *AW, *AQc;
  double* slab;
  double* y;
  const int* pub;      // pub[pub_beg + t]: hand-off slot of a supernode's published value number t
  const int* tg_reg;
  const int* xreg;
  const long long* xsrc;
  const long long* rsrc;
  // Hand-off slots, two sets (run parity, see tree_fused.hip): [0, updb_base) the consumer-ordered
  // Schur-update slots of BuildPlans, then the forward-value slots; and the solution entries a
  // descendant's back substitution reads.
  double* hand;
  long long hand_stride;
  long long updb_base;
  double* ysig;
  long long ysig_stride;
  int gen;  // parity of this run
  // kFusedTriple: right-hand sides 1 and 2 have hand-off slots of their own -- forward values fwd_stride
  // and 2 fwd_stride behind those of right-hand side 0, solution entries y_stride and 2 y_stride behind --
  // in sets that alternate with the triple launches only (tgen); their solutions go to y3 (3 x y_stride)
  int tgen;
  long long fwd_stride, y_stride;
  double* y3;
  int* fail;
  int tag;  // a failed pivot writes fail[1] = tag
  double k, bs, cs;   // y = k (b bs + AQc cs) - 2 AW  (cone_program.cc:409-411)
  double cb, cq, cw;  // or (comb != 0) y = cb b + cq AQc + cw AW  (cone_program.cc:181, 504)
  int comb;
  int form;  // solve-only sweeps: 0 the right-hand side is in y, 1 k (b bs + AQc cs) - 2 AW, 2 cb b + cq AQc + cw AW
  const double* k_from;  // form 1, not null: k = k_from[0], the barrier parameter the device selected (cxk_select_mu_async)
  const double* sc;  // per-constraint <w,c>, <c,Qc>
  double* sys_sc;
  int K;
  double* host_flag;  // pinned host word: set to 1.0 when a wait ran out (the sets are then rebuilt)
  // A wavefront whose values cannot be there yet SLEEPS before it starts to poll (it shares a SIMD with
  // wavefronts that are eliminating, which are bound by instruction issue): word 63 of its record is its
  // level, a level of the tree takes at least up_sleep units of 64 cycles.  (Measured on C4, same box:
  // 30.3 -> 29.8 us at 20 .. 40 units, back to 30.1 at 60, 31.7 at 100; the same on the way down: nothing.)
  int up_sleep;
  // ---- sharded contexts (kFusedShardUp / kFusedShardTop).  Positions [0, count_up) are
  // this rank's own subtrees, [count_up, count) the replicated top of the tree.  The exchange buffer
  // x = [T slab entries (n_xs) | AW_T | AQc_T | fwd_T (n_xv each) | <w,c> <c,Qc> fail pad] is what ONE
  // sum all-reduce carries between the two launches (layout of exchange_pack, kernels_kkt.hip.h).
  int count_up;
  double* x;
  long long n_xs;
  int n_xv;
  // pack (workgroups behind the supernodes of kFusedShardUp): per exchange entry its own-rank sources
  // in G (assemble_gather's record over as_src) and the published Schur updates of this rank's
  // subtrees into it (xs_pt -> pt_ptr / pt_src, hand-off slots); per top variable its own-rank
  // sources of AW / AQc (xr over rs_src) and the published forward values (pf_ptr / pf_src)
  const GatherRec* xg;
  const int64_t* as_src;
  const int* xs_pt;
  const int* pt_ptr;
  const int64_t* pt_src;
  const ResidRec* xr;
  const int64_t* rs_src;
  const int* pf_ptr;
  const int* pf_src;
  // every supernode of the up launch counts itself at its end -- 64 counters, 128 bytes apart, position
  // modulo 64 -- and the workgroup that writes the buffer's tail (scalars, failure flag) waits until
  // each shows done_target (= up launches so far) times its share of the supernodes
  unsigned long long* done;
  unsigned long long done_target;
};

// The lane that takes the first right-hand side as a ROW of the panel in tree_fused's elimination (the next
// ones follow it), or -1: shapes whose column updates reach a free lane -- <16, 8>: separator rows end at lane
// 23, the updates cover DPP row 1; <24, 0>: rows end at lane 23, the updates cover DPP rows 0 and 1.
constexpr int FusedRhsLane(int nsmax, int smax) { return (nsmax == 16 && smax == 8) || (nsmax == 24 && smax == 0) ? 24 : -1; }

// What a launch does.  A tree whose supernodes are all resident at once takes kFusedFull (assembly,
// factorization with the first right-hand side, back substitution) and kFusedSolve (forward + back
// substitution on the stored factor); a larger one the same work as two launches each -- kFusedUp
// then kFusedDown, kFusedForward then kFusedDown -- because a wavefront that waits for its
// ANCESTORS while it holds a slot could keep them from ever starting.
// Sharded contexts: kFusedShardUp = assembly + factorization + forward substitution of the rank's own
// subtrees with the PACK of the exchange buffer riding behind them (extra workgroups that wait for the
// published values); after the all-reduce kFusedShardTop = the replicated top straight from the
// buffer (the unpack is the top supernodes' load phase), factored and solved, and the back
// substitution down the own subtrees -- top workgroups first, then the subtrees root side first, so
// only the top has to be resident at once.
// kFusedTriple: kFusedFull with THREE right-hand sides -- bs b, cs AQc, AW -- whose solutions go to y3 (y_stride
// apart) and y = K^-1 (-bs b + cs AQc): the mu selection's solve and, by linearity, the Newton direction
// for any mu without another sweep (tree_fused.hip, FusedSupernode).
enum FusedTreeMode { kFusedFull = 0, kFusedSolve = 1, kFusedUp = 2, kFusedForward = 3, kFusedDown = 4,
                     kFusedShardUp = 5, kFusedShardTop = 6, kFusedTriple = 7 };

// Register shapes (NSMAX << 8 | SMAX) of the tree's supernodes: at most two (shape_b == shape_a for
// one).  False when no instance is compiled for the pair.
bool FusedTreeCompiled(int shape_a, int shape_b);
// Workgroups per CU the hardware can hold of the instance (0 on error); sharded: of kFusedShardTop.
int FusedTreeOccupancy(int shape_a, int shape_b, bool sharded = false);
// ev_start / ev_stop (both or neither): a hipEvent pair carried by the dispatch itself (its begin / end time stamps)
hipError_t LaunchFusedTree(const FusedTreeArgs& a, int shape_a, int shape_b, int mode, hipStream_t stream,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

}  // namespace cxk
