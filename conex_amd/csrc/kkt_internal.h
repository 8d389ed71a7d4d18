// Shared by the translation units of the cxk_* path: the context, its host-side types and the
// functions that cross between them.  kkt_plans.hip builds the symbolic plans (tree structure,
// partition, index tables: host code only), kkt_context.hip holds the launches and the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library itself is loaded on demand (cxk_comm_init_rccl)

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/conex_kkt_hip.h"
#include "kernels_kkt.hip.h"      // record types, FactorPlan (kkt_plans.hip defines CXK_DEVICE_FUNCTIONS_ONLY first)
#include "kernels_kkt_top.hip.h"  // TopDenseArgs
#include "lmi_types.h"
#include "symbolic.h"
#include "tree_fused.h"

using namespace cxk;

namespace cxk_host {


struct ConstraintRec {
  int type = 0, n = 0, m = 0;
  int herm_d = 0;  // Hermitian PSD over R/C/H: number of real planes d; n is then d * order
  int eq_rows = 0; // CXK_STATIC built from EqualityConstraints: number of multipliers (last clique entries)
  std::vector<double> A, C;
  std::vector<double> Q;  // CXK_QUAD: n x n inner-product matrix, empty = identity
  int group = -1, member = -1;
  bool sparse = false;  // LMI evaluated from its nonzeros (kernels_lmi_sparse.hip.h)
  // every A_i and C equals its transpose.  The fast kernels use tr(W A_i W A_j) = tr(P_i P_j),
  // P = A W, which needs that; anything else takes the literal kernels (dense_lmi_constraint.cc:72-88)
  bool symmetric = true;
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  // `zeroed`: the code relies on the initial zeros (slots that are never written but read).
  // Everything else is scratch that must be written before it is read: with CXK_DEBUG_FILL_NAN=1
  // in the environment such buffers start as NaN (all-ones bytes), so that a read of unwritten
  // memory shows up in the results instead of passing by luck (diagnostic runs of the test suite).
  hipError_t alloc(size_t count, bool zeroed = false) {
    release();
    n = count;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    static const bool nan_fill = [] {
      const char* v = getenv("CXK_DEBUG_FILL_NAN");
      return v && atoi(v) != 0;
    }();
    // the fill runs on the null stream, kernels on the context's stream, which may be
    // non-blocking (no implicit ordering with the null stream): wait for it on the host
    if (e == hipSuccess) e = hipMemset(p, (nan_fill && !zeroed) ? 0xFF : 0, count * sizeof(T));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
  }
  hipError_t upload(const std::vector<T>& v) {
    hipError_t e = alloc(v.size());
    if (e != hipSuccess || v.empty()) return e;
    e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
  }
};

struct Group {
  int type = 0, n = 0, m = 0;
  std::vector<int> ids;
  DevBuf<double> A, C, W, T1, T2;
  DevBuf<double> Apad;  // lmi_schur_mfma at a padded order: [A_1 .. A_m | C] per member, zero-padded (LmiMfmaPaddedOrder)
  DevBuf<int> dids;
  bool has_q = false;             // CXK_QUAD: the members carry an inner-product matrix Q
  DevBuf<double> Aleft;           // Hermitian C / H groups on the GEMM assembly: first n / herm_d columns of [A_i | C]
  DevBuf<double> Apk;             // CXK_LMI: packed lower triangles of the A_i (LmiGroup::Apk), may be empty
  DevBuf<double> qQ, qGram, qS;   // CXK_QUAD: Q, A1' Q A1, the state kept between PrepareStep and TakeStep
  int herm_d = 0;
  bool mfma = false;     // lmi_schur_mfma (lmi_fused_mfma.hip)
  bool literal = false;  // non-symmetric data: literal kernels only
  // orders beyond the LDS-resident kernels: HBM-resident matrices + MFMA GEMM pipeline
  bool large = false;
  // assembly through the batched MFMA GEMM pipeline (always when `large`; also for LDS-resident
  // shapes from order 9 up without a register-kernel instance, where it measured 1.5-3.3x faster than the LDS kernel)
  bool schur_gemm = false;
  DevBuf<double> ws_main, ws_gf, ws_part;
  DevBuf<int> ws_piv;
  int splits = 1;
  // sparse LMI groups: nonzeros matrix-major and position-major instead of the dense A
  bool sparse = false;
  int sp_lpp = 1;                  // lanes sharing one (i, j) pair sum
  int sp_chunks = 1, sp_emax = 0;  // sp_emax: most nonzeros in one constraint
  bool sp_cdense = false;          // dense affine term: X = W C W instead of pair sums with C
  bool sp_small = false;           // W (and X) of a constraint fit in LDS
  DevBuf<int> sp_eptr, sp_erc, sp_pptr, sp_pvar, sp_pairs;
  DevBuf<double> sp_eval, sp_pval;
};


}  // namespace cxk_host
using namespace cxk_host;

struct cxk_context {
  int num_vars = 0;
  int device = -1;
  int cus = 256;  // multiprocessors of the device
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<ConstraintRec> cons;
  IntLists cliques, dual_vars;
  // Reference identity (the default): the reference AS WRITTEN -- (i) BindDiagonalBlock's unchecked
  // direct_update placement on fill-in supernodes (supernodal_assembler.cc:72-91), (ii) raw Lanczos
  // Ritz values (approximate_eigenvalues.cc:178-239).  0 = the corrections (scatter by position,
  // Samuelson clamp): cxk_set_reference_identity(ctx, 0) or CXK_REFERENCE_QUIRKS=0 in the
  // environment, before cxk_finalize.
  int reference_identity = -1;  // -1: take the environment's word at finalize
  bool finalized = false;     // symbolic analysis done (getters)
  bool device_ready = false;  // device buffers and plans built: numeric entry points may run
  int rank = 0, world = 1;
  MatrixData md;
  Layout lay;
  // the reference's structure (what the getters report); md / lay differ from it only when a long
  // chain-shaped tree is factored in a segment-parallel order (symbolic.h, SegmentChain)
  MatrixData md_ref;
  Layout lay_ref;
  int chain_segments = -1;  // cxk_set_chain_segments: -1 automatic / environment, 0 off, P segments
  int segments = 0;         // segments in use (0: the reference's order)
  std::vector<Group> groups;
  std::vector<int64_t> g_off, r_off;
  std::vector<unsigned char> owned;      // constraint i assembled/updated by this rank
  // elimination-tree structure + partition
  std::vector<int> t_ns, t_nsep, t_start, t_level, t_parent;
  std::vector<unsigned char> sn_top;     // supernode belongs to the replicated top T
  std::vector<unsigned char> sn_mine;    // supernode factored by this rank (own subtree or T)
  std::vector<unsigned char> var_valid;  // permuted variable whose y this rank holds
  int nlev = 0, cut_level = 0;           // levels >= cut_level form T (world > 1)
  int64_t n_xs = 0;                      // exchange: T slab entries
  int n_xv = 0;                          // exchange: T variables
  // levels
  std::vector<int> level_ptr, level_sn;
  size_t chol_lds = 0, solve_lds = 0;  // bytes of LDS staging one supernode needs
  int top_level = 0;                   // levels [top_level, nlev) run inside one workgroup
  // iterative refinement (kkt_solver.cc:233-261): steps per solve, the assembled matrix kept by the
  // factor sweep, the right-hand side of the running solve, K y pieces, the iterate
  int refine_iters = 0;
  bool slab0_valid = false;
  DevBuf<double> slab0, rhs0, mv_u, mvb, ysave;
  bool no_ranges = false;              // CXK_NO_RANGES=1: downward sweeps level by level (comparison runs)
  bool no_lean = false;                // CXK_NO_LEAN=1: generic kernels only (comparison runs)
  std::vector<unsigned char> level_big;  // level holds a supernode beyond the wave-per-supernode kernels
  // A level's (non-huge) supernodes are sorted into SEGMENTS of one register shape; a segment
  // whose pulls fit the dense slots takes tree_factor_level, one whose separator lists are inline
  // takes tree_backward_level, anything else the generic tree_sweep on its sub-range.
  struct LevelSeg {
    int begin = 0, end = 0;  // positions into the level-ordered records
    int shape = 0;           // NSMAX << 8 | SMAX, 0 = no register kernel
    bool fast = false, inl = false;
  };
  std::vector<std::vector<LevelSeg>> level_segs;
  std::vector<unsigned char> level_lean;  // every segment of the level has both lean kernels
  // the chain at the top: levels [chain_level, nlev) hold one lean supernode each and are swept by
  // one wavefront in one launch (tree_chain_lean); chain_a / chain_b = the (at most two) shapes
  int chain_level = 0, chain_a = 0, chain_b = 0;
  // two consecutive downward levels in one launch (tree_backward_pair): indexed by the UPPER level
  struct BackPair {
    int nwg = 0, shape_p = 0, shape_c = 0;
    DevBuf<BackPairEntry> tab;
  };
  std::vector<std::unique_ptr<BackPair>> back_pairs;
  // supernodes whose panel exceeds LDS sit at the END of their level list and are swept one by
  // one through the blocked HBM path (kernels_kkt_big.hip.h); level_nh = count of the others
  std::vector<int> level_nh;
  std::vector<SnRec> h_recs;  // host copy of the level-ordered records
  DevBuf<double> step_slots;  // sharded: world x 4 partial step results behind one sum all-reduce
  DevBuf<double> big_ws;
  DevBuf<int> big_flags;  // big_chol_dataflow's per-block-column words (null: the host-driven panel loop)
  int big_gen = 0;
  // Level ranges below the top that are swept by one launch each: workgroup g of range r sweeps
  // one connected piece of the elimination forest restricted to levels [lo, hi)
  struct SweepRange {
    int lo = 0, hi = 0, groups = 0, waves = 1;
    DevBuf<int> wg_lev;  // [groups * (hi - lo + 1)] positions into rec_r
  };
  std::vector<std::unique_ptr<SweepRange>> ranges;
  // the top levels as one dense factorization (kernels_kkt_top.hip.h), when they hold <= 64 columns
  struct TopDense {
    bool on = false;
    TopDenseArgs args;
    DevBuf<int> off, pl_ptr, pl_dst, pl_src, plb_ptr, plb_src;
  } top_dense;
  int dense_level = 0;  // first level of the dense range (== nlev when off)
  DevBuf<SnRec> rec_r;   // records in (range, workgroup, level) order
  // device state
  DevBuf<double> G, AWc, AQcc, sc, slab, y, b, AW, AQc, sys_sc, info2, info4, red_out, scal_out;
  DevBuf<int64_t> d_g_off, d_r_off, as_dst, as_src, rs_src;
  DevBuf<GatherRec> as_rec;
  DevBuf<ResidRec> rs_rec;
  DevBuf<int> as_ptr, rs_ptr, cl_ptr, cl_perm, d_level_sn, d_level_ptr, d_fail, d_pinv, tg_loc, tg_reg;
  DevBuf<unsigned char> d_mask;
  DevBuf<int> p_ns, p_nsep, p_start, tg_ptr, tr_ptr, fs_ptr, fs_src, bs_ptr, bs_c, bs_row, updb_off;
  DevBuf<int64_t> p_diag, p_offd, tr_src, upd_off;
  DevBuf<SnRec> p_rec;
  DevBuf<int> pub_dst, pubb_dst;
  DevBuf<double> upd, updb, xbuf;
  DevBuf<int64_t> xs_off, pt_dst, pt_src;
  DevBuf<int> xv_idx, pt_ptr, pf_ptr, pf_src, xs_pt;
  int64_t as_T = 0;
  // the assembly folded into the first factor level (tree_factor_level_asm): records of the level's
  // supernodes, and the gather lists without what those supernodes load themselves
  bool fused_asm = false;
  DevBuf<AsmRec> asm_rec;
  DevBuf<GatherRec> as_rec2;
  DevBuf<ResidRec> rs_rec2;
  DevBuf<int> rs_var2;
  int64_t as_T2 = 0;
  int rs_N2 = 0;
  struct AsmPending {
    bool on = false;
    int with_rhs = 0;  // GatherArgs::with_rhs
    double k = 0, bs = 0, cs = 0, cb = 0, cq = 0, cw = 0;
  } asm_pending;
  // The whole tree in one launch (tree_fused.hip): records, the lists of entries / variables with
  // several sources, the two sets of hand-off slots and their initial images (re-uploaded after a
  // wait ran out), the run counter whose parity picks the set
  bool fused_tree = false;
  bool fused_split = false;  // more supernodes than resident wavefronts: the way up and the way down are two launches
  bool fused_sweep = false;  // solve-only sweeps in one launch too (CXK_NO_FUSED_SWEEP=1 turns this part off)
  // sharded contexts: own subtrees up (+ pack of the exchange buffer) and top + way down as two launches
  // around the all-reduce; fused_up = positions below the cut; pack tables; the done counter of the up launch
  bool fused_shard = false;
  int fused_up = 0;
  DevBuf<GatherRec> fx_xg;
  DevBuf<ResidRec> fx_xr;
  DevBuf<unsigned long long> fx_done;
  unsigned long long fx_done_target = 0;
  int fused_sa = 0, fused_sb = 0;
  DevBuf<int> fx_rec, fx_xreg;
  DevBuf<long long> fx_xsrc, fx_rsrc;
  DevBuf<int> fx_pub;
  DevBuf<double> fx_hand, fx_ysig;
  std::vector<double> fx_hand_init;
  long long fx_updb_base = 0;
  // the whole-tree launch with three right-hand sides (tree_fused.h kFusedTriple): stride between the
  // right-hand sides' forward slots, parity of the triple launches, their three solutions, and "y is to be
  // formed from them by the PrepareStep that follows" (cxk_newton_direction_device_mu then launches nothing)
  long long fx_fwd_stride = 0;
  unsigned fused_tgen = 0;
  DevBuf<double> y3;
  bool y3_valid = false;
  bool y_deferred = false;  // the direction is still y3 and mu_dev: combined inside the PrepareStep launch that follows (or FlushDeferred)
  bool no_y_deferral = false;  // CXK_NO_Y_DEFERRAL=1 at cxk_create: the direction in a launch of its own (newton_from_three)
  DevBuf<unsigned long long> y_done;  // count of the direction's workgroups, all launches so far
  unsigned long long y_done_target = 0;
  bool no_triple = false;  // CXK_NO_TRIPLE=1 at cxk_create: the mu selection's solve and the Newton direction as two sweeps
  unsigned fused_gen = 0;
  double* fx_flag = nullptr;  // pinned host word the kernel sets when a wait ran out
  int debug_timeout_at = -1, fused_launches = 0;  // CXK_DEBUG_FUSED_TIMEOUT_AT (test hook, LaunchFusedTreeSolve)
  int fused_timeouts = 0;            // times that happened (the whole-tree launch is given up at the first)
  bool timeout_pending = false;      // seen (and the slots rebuilt) by MakeFusedTreeArgs, not yet acted on
  bool timeout_unreported = false;   // ... and cxk_fused_tree_timed_out has not told the caller yet
  bool asm_deferred = false;  // cxk_assemble ran the Schur kernels; the gather waits for the factorization that follows
  // solve-only sweeps whose every forward launch is a lean kernel form the right-hand side inside
  // those kernels (RhsIn) instead of in a launch of their own
  bool forward_all_lean = false;
  RhsIn rhs_in{};  // form 0 unless such a sweep is being enqueued
  int asm_tag = 0;    // tag of the latest fused launch (a failed pivot there writes d_fail[1] = tag)
  bool fail_clean = false;  // d_fail[0] was cleared by the latest gather and no factorization has run since
  int fail_tag = 0;   // what mailbox_pack compares d_fail[1] with: asm_tag, or 0 after any other factorization
  FactorPlan plan{};
  // index of the next PrepareStep / eigenvalue query (keys the Hermitian start vectors)
  unsigned long long lanczos_calls = 0;
  // equality constraints: next multiplier id, LDLT state (kkt_solver.cc:180-193)
  int dual_start = -1;
  bool use_ldlt = false;
  DevBuf<int> d_tr, d_reg;
  DevBuf<double> y2;  // second solve vector of the line search
  std::vector<double> y_at_prepare;  // lambda_ = y.tail(rows) is latched by PrepareStep
  // Collectives of a sharded context (world > 1): RCCL over xGMI (cxk_comm_init_rccl: librccl.so is
  // loaded on demand, all-reduces run on the context's stream), or a caller-supplied all-reduce
  // (cxk_comm_set_allreduce: other transports, tests).  count_mask[p] = 1 where this rank's value
  // of permuted variable p counts in a cross-rank sum (own subtrees; the replicated top on rank 0).
  struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
  } rccl;
  cxk_allreduce_fn coll_fn = nullptr;
  void* coll_user = nullptr;
  DevBuf<unsigned char> d_count_mask;
  DevBuf<double> shard_tmp;        // N doubles: this rank's share of a vector / of the per-constraint pairs
  double rhs_c[3] = {0, 0, 0};     // right-hand side of the running factor-and-solve: cb b + cq AQc + cw AW
  // kernel clocks (bench.py's roofline entries): a hipEvent pair on every timing_period-th launch of
  // the kernels of a slot (CXK_CLOCK_*: assembly, the tree launch of a KKT solve, solve-only sweep,
  // eigenvalue query, PrepareStep, TakeStep) -- attached to the dispatch itself where the slot is ONE
  // kernel (its own begin / end time stamps, the quantity rocprofv3 reports), recorded around the
  // launches otherwise (which adds their boundaries)
  bool timing = false;
  int timing_period = 1;
  int timing_tick[CXK_CLOCK_COUNT] = {0, 0, 0, 0, 0, 0};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<int> ev_slot;
  size_t ev_used = 0;
  hipEvent_t clk_e0 = nullptr, clk_e1 = nullptr;  // the pair the next whole-tree launch carries on its dispatch
  double time_acc_ms[CXK_CLOCK_COUNT] = {0, 0, 0, 0, 0, 0};
  int time_samples[CXK_CLOCK_COUNT] = {0, 0, 0, 0, 0, 0};
  // kkt_solver = CONEX_QR_FACTORIZATION (kkt_solver.cc:172-231): the reference factors the DENSE
  // N x N KKT matrix with a column-pivoted Householder QR (Eigen::ColPivHouseholderQR) and solves
  // with it -- a debugging mode for rank-deficient systems, O(N^2) memory and O(N^3) work on one
  // host core there.  Same here: Factor brings the assembled slab to the host, solves go through
  // the host; everything else of the iteration stays on the device.  Orders beyond kQrMaxOrder are
  // refused.
  int solver_mode = 0;               // 0 LLT / LDLT by structure (the reference's modes 0 and 1), 2 QR
  struct DenseQr {
    int n = 0, rank = 0;
    std::vector<double> qr, tau;     // Householder vectors below the diagonal, R on and above
    std::vector<int> piv;
    bool valid = false;
  } qr;
  // Per-phase device timers (the reference's START_TIMER / END_TIMER of debug_macros.h:18-52 around
  // Assemble / Factor / Solve / Update, cone_program.cc:338-437): hipEvents recorded on the stream
  // at every phase mark; the time between two consecutive marks belongs to the earlier phase.
  bool phase_on = false;
  std::vector<std::pair<hipEvent_t, int>> phase_marks;  // (event, phase that starts there)
  std::vector<hipEvent_t> phase_pool;
  double phase_us[CXK_PHASE_COUNT] = {0, 0, 0, 0, 0};
  // Host mailbox (pinned, device-visible): the scalars the IPM loop reads every iteration --
  // reduced step info / eigenvalue bounds [0..3], step scalars [4..9], factor-failure flag [10],
  // sequence number [11] -- are written by one tiny kernel at the end of the enqueued work and
  // picked up by the host without a D2H copy.  seq counts enqueued producers; mb_seen is the
  // value the mailbox carried when the host last waited for it.
  double* mb = nullptr;
  double mbv[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // the last validated snapshot of the mailbox (WaitMailbox)
  double* pin_y = nullptr;  // pinned staging of y for cxk_get_y
  long long seq = 0, mb_seen = -1, factor_seq = -1, scal_seq = -1;
  // cxk_step_scalars_async leaves its launch to the tail workgroup of the PrepareStep that normally
  // follows (StepTail, kernels_cone.hip.h); any other entry point runs it first (FlushDeferred)
  bool scal_deferred = false;
  bool no_step_tail = false;  // CXK_NO_STEP_TAIL / CXK_PREPARE_LDS in the environment at cxk_create
  bool no_device_mu = false;  // CXK_NO_DEVICE_MU likewise
  bool prepare_lds = false;   // CXK_PREPARE_LDS likewise: the workgroup PrepareStep kernels instead of the register ones
  DevBuf<double> tail_slots;  // two sets of 4 per constraint, armed with kTailSentinel, used in turn
  int tail_parity = 0;
  DevBuf<double> mu_dev;      // [1] inv_sqrt_mu as selected on the device (cxk_select_mu_async)
};

#define CXK_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                         \
      fprintf(stderr, "%s line %d: %s\n", __FILE__, __LINE__, ctx->err.c_str());            \
      return CXK_FAILURE;                                                                   \
    }                                                                                       \
  } while (0)

#define CXK_DEMAND(cond, msg)                                                    \
  do {                                                                           \
    if (!(cond)) {                                                               \
      ctx->err = msg;                                                            \
      fprintf(stderr, "%s line %d: %s\n", __FILE__, __LINE__, msg);              \
      return CXK_FAILURE;                                                        \
    }                                                                            \
  } while (0)

namespace cxk_host {

constexpr size_t kLdsLimit = 160 * 1024 - 512;
constexpr int kChainMaxLevels = 1 << 30;  // no limit: the kernel keeps the last kChainRing records in LDS and re-reads the rest
constexpr int kSplitTopLevels = 8;  // tops of at most this many levels may be swept level by level

int Fail(cxk_context* ctx, const char* msg);
// two-shape chains tree_chain_lean is compiled for (kkt_context.hip, LaunchChain)
bool ChainPairCompiled(int sa, int sb);
// dynamic-LDS limit of the tree_top_dense instances, raised on the current device (kkt_context.hip)
hipError_t RaiseTopDenseLimits();

// kkt_plans.hip
void ComputeTreeStructure(cxk_context* ctx);
void PartitionTree(cxk_context* ctx);
int BuildPlans(cxk_context* ctx);

}  // namespace cxk_host
