/*
 * conex_kkt_hip.h -- C-ABI of the MI355X-native Newton-step KKT path.
 *
 * This is the "inner" drop-in boundary: the device-resident replacement for
 * what conex/cone_program.cc::Solve calls per IPM iteration.  The "outer"
 * boundary (the reference's interfaces/conex.h, 21 CONEX_* functions) is in
 * include/conex.h and is implemented on top of these entry points.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross the boundary
 *   - all matrices column-major fp64, all indices 32-bit int (as the reference)
 *   - every function returns 0 on success (CONEX_SUCCESS polarity,
 *     conex/error_codes.h) unless documented otherwise; no exception escapes
 *   - a context is bound to ONE HIP device and ONE stream and is not re-entrant
 *     (the reference's Program is not thread-safe either: kkt_solver.h:58,62)
 *   - host arrays are copied at call time (interfaces/conex.cc:143-159)
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef CONEX_KKT_HIP_H
#define CONEX_KKT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cxk_context cxk_context;

enum { CXK_SUCCESS = 0, CXK_FAILURE = 1 };

/* cone types; the plugin surface of conex/constraint.h:51-197 */
enum {
  CXK_LMI = 0,    /* DenseLMIConstraint  dense_lmi_constraint.h:24-41 */
  CXK_LINEAR = 1, /* LinearConstraint    linear_constraint.h:14-84 */
  CXK_SOC = 2,    /* SOCConstraint       soc_constraint.h:6-54 */
  CXK_STATIC = 3, /* constant Schur block: QuadraticFunction quadratic_cost.cc:18-69,
                     SupernodalAssemblerStatic supernodal_assembler.h:122-129 */
  CXK_QUAD = 4,   /* QuadraticConstraint quadratic_cone_constraint.h:11-86: Lorentz cone
                     x0 >= sqrt(x1' Q x1) with an inner-product matrix Q on the vector part */
  CXK_OCT = 5     /* HermitianPsdConstraint<Octonions> hermitian_psd.cc:116-168, 171-230: Hermitian
                     matrices of order <= 3 over the octonions (cxk_add_hermitian with d = 8) */
};

/* ---- lifetime --------------------------------------------------------- */
/* Program(int number_of_variables) cone_program.h:101-104.
 * device < 0: host-only context (symbolic analysis works, numeric calls fail loudly).
 * stream: a hipStream_t (NULL = the device's null stream). */
int cxk_create(int num_vars, int device, void* stream, cxk_context** out);
void cxk_destroy(cxk_context* ctx);
const char* cxk_last_error(const cxk_context* ctx);

/* ---- constraint plugins: Program::AddConstraint(T, variables) cone_program.h:192-218,
 *      ConstraintManager::AddConstraint constraint_manager.h:50-64 (IsUnique check).
 *      Return the constraint id (>= 0) or -1 when rejected. vars == NULL means 0..num_vars-1. */
int cxk_add_lmi(cxk_context* ctx, int n, int m, const double* A /* m x (n x n) */,
                const double* C /* n x n */, const int* vars);
/* HermitianPsdConstraint<Real|Complex|Quaternions>(n, a, c) (hermitian_psd.h:41-52; what
 * CONEX_NewLinearMatrixInequality creates, interfaces/conex.cc:286-318).  d = 1, 2 or 4 real
 * planes; A = m x d planes of n x n (column-major, plane 0 symmetric, the others skew),
 * C = d planes.  W / dual variables are exchanged as d planes (cxk_dual_size = d n^2).  On the
 * device the cone is held through its real representation of order d n and runs on the LMI
 * kernels with the reference's Hermitian step rules (Taylor-squaring exponential, random-start
 * Lanczos, hermitian_psd.cc:10-91).  d = 8 (octonions, n <= 3): no real representation exists --
 * a cone type of its own (CXK_OCT) with the reference's octonion rules (hermitian_psd.cc:108-168:
 * quadratic representations in place of W A W, its heuristic step norms, GeodesicUpdateScaled). */
int cxk_add_hermitian(cxk_context* ctx, int n, int d, int m, const double* A, const double* C,
                      const int* vars);
/* Program::AddConstraint(EqualityConstraints{A, b}, vars) -> ConstraintManager::
 * AddEqualityConstraint (constraint_manager.h:66-90, equality_constraint.{h,cc}): A y[vars] = b,
 * A rows x m column-major.  Appends `rows` multipliers to the KKT system (cxk_system_size grows);
 * the Schur block is the constant [0 A^T; A 0] with AQc = [0; b].  Any equality switches
 * cxk_factor / the solves to the LDLT path (BlockLDLTInPlace block_triangular_operations.cc:
 * 315-349 over Eigen::RLDLT, RLDLT.h:298-431; kkt_solver.cc:180-193): factor always reports ok,
 * cxk_factor_regularized tells whether a pivot was clamped to +-1e-9.  cxk_get_W returns the
 * multipliers latched by the last cxk_prepare_step (lambda_, equality_constraint.cc:32-37). */
int cxk_add_equality(cxk_context* ctx, int rows, int m, const double* A /* rows x m */,
                     const double* b /* rows */, const int* vars);
int cxk_factor_regularized(cxk_context* ctx, int* flag);
int cxk_add_linear(cxk_context* ctx, int rows, int m, const double* A /* rows x m */,
                   const double* c /* rows */, const int* vars);
int cxk_add_soc(cxk_context* ctx, int n, int m, const double* A /* (n+1) x m */,
                const double* c /* n+1 */, const int* vars);
int cxk_add_static(cxk_context* ctx, int m, const double* G /* m x m */, const int* vars);
/* QuadraticConstraint(Q, A, c): c - A y in { (x0, x1) : x0 >= sqrt(x1' Q x1) }.  Q: n x n column-major,
 * symmetric positive definite, or NULL for the identity (the reference's two-argument constructor) */
int cxk_add_quadratic(cxk_context* ctx, int n, int m, const double* Q /* n x n or NULL */,
                      const double* A /* (n+1) x m */, const double* c /* n+1 */, const int* vars);
int cxk_num_constraints(const cxk_context* ctx);

/* Multi-GPU sharding (no reference counterpart -- the reference is single
 * process).  Must precede cxk_finalize.  This context then assembles/updates only the
 * constraints it owns (round-robin over elimination subtrees) and exposes a contiguous
 * exchange slab that the caller sum-reduces across ranks (RCCL all-reduce). */
int cxk_set_shard(cxk_context* ctx, int rank, int world_size);

/* Chain-shaped elimination trees (every step updates the next one only: BASELINE config 3 as the
 * reference's tests arrange it -- clique k = {8k .. 8k+9}, 5000 strictly dependent steps).  What the
 * getters below report is always the reference's order, supernodes, separators and permutation
 * (bit-identical: tests/test_symbolic_parity.py).  The FACTORIZATION of such a tree runs in a
 * segment-parallel order: the chain is cut into P pieces, the variables that carry an update across
 * a cut are eliminated last (in the root), the pieces become independent subtrees swept side by side
 * (symbolic.h, SegmentChain).  Same matrix, another elimination order: every result -- direction,
 * residuals, scaling points -- is the reference's to rounding (<= 1e-10 against the oracle); only the
 * stored factor differs, which is why cxk_get_slab / cxk_set_slab refuse such a context.
 * segments: 0 = the reference's order, P >= 2 = that many pieces; default (no call): the environment's
 * CXK_CHAIN_SEGMENTS, else automatic for chains of 256 steps or more.  Before cxk_finalize. */
int cxk_set_chain_segments(cxk_context* ctx, int segments);
int cxk_chain_segments(const cxk_context* ctx);

/* Reference identity: ON by default -- the library computes what the reference as written computes.
 * cxk_set_reference_identity(ctx, 0) (before cxk_finalize; or CXK_REFERENCE_QUIRKS=0 in the
 * environment) opts into two corrections of the reference:
 *   (i)  a constraint's Schur block is scattered by variable position also on fill-in supernodes,
 *        where BindDiagonalBlock's unchecked direct_update test (supernodal_assembler.cc:72-91)
 *        puts it one row and column off (a defect on rare structures);
 *   (ii) the Lanczos Ritz values of PrepareStep / GetWeightedSlackEigenvalues
 *        (approximate_eigenvalues.cc:178-239) are clamped to Samuelson's bound on the spectrum of
 *        W S (a noise-dominated Lanczos run near convergence otherwise yields an eigenvalue far
 *        outside the spectrum and a needlessly small step).
 * With identity on, CONEX_Maximize reproduces the iteration count and mu sequence of the reference
 * algorithm (tests/test_gpu_solver.py::test_reference_identity_reproduces_the_oracle_trajectory). */
int cxk_set_reference_identity(cxk_context* ctx, int on);

/* Initialize(): symbolic analysis (SupernodalKKTSolver ctor kkt_solver.cc:104-116),
 * Bind (kkt_solver.h:26-33), workspace carve + SetIdentity (cone_program.cc:78-112),
 * upload of constant data, construction of device index tables and level schedule.
 * Environment read here: CXK_SPARSE_LMI=0/1 (force the dense / sparse LMI evaluation),
 * CXK_REFERENCE_QUIRKS=0 (= cxk_set_reference_identity(ctx, 0) unless that was called). */
int cxk_finalize(cxk_context* ctx);

/* ---- symbolic results (MatrixData supernodal_solver.h:18-29; bit-exact vs reference) */
int cxk_system_size(const cxk_context* ctx); /* N */
int cxk_get_order(const cxk_context* ctx, int* order /* K */);
int cxk_get_permutation(const cxk_context* ctx, int* perm, int* perm_inv /* num_vars */);
/* which: 0 cliques(permuted) 1 supernodes_orig 2 separators_orig 3 supernodes_pos
 *        4 separators_pos ; returns length (out may be NULL) */
int cxk_get_list(const cxk_context* ctx, int which, int e, int* out);
int cxk_get_supernode_sizes(const cxk_context* ctx, int* out /* K */);
long cxk_slab_size(const cxk_context* ctx);
int cxk_get_block_offsets(const cxk_context* ctx, long* diag_off, long* offd_off /* K */);
int cxk_get_ss_index(const cxk_context* ctx, int e, long* out); /* returns count */
int cxk_num_levels(const cxk_context* ctx);

/* ---- scaling point W (workspace()->W, constraint.h:159-167) ------------ */
int cxk_set_identity(cxk_context* ctx);                       /* SetIdentity */
int cxk_dual_size(const cxk_context* ctx, int i);
int cxk_get_W(cxk_context* ctx, int i, double* out);          /* get_dual_variable */
int cxk_set_W(cxk_context* ctx, int i, const double* in);     /* warm start import */

/* ---- Newton step, device resident ------------------------------------- */
/* solver->Assemble() + AssembleSchurComplementResiduals  cone_program.cc:338-341 */
int cxk_assemble(cxk_context* ctx);
/* solver->Factor() kkt_solver.cc:172-199 (LLT mode). *ok = 1 success, 0 not PD. Syncs. */
int cxk_factor(cxk_context* ctx, int* ok);
/* The same without waiting: cxk_factor_async enqueues the factorization, cxk_factor_status
 * returns its LLT flag -- at no cost when a later blocking call (cxk_prepare_step,
 * cxk_weighted_slack_eigenvalues, cxk_step_scalars, cxk_sync) has already waited for the stream:
 * every wait brings the flag, the reduced step info and the step scalars back together through
 * one pinned host mailbox.  conex::Solve's per-iteration host round trips (cone_program.cc:360,
 * 417, 439-446) collapse into one or two this way. */
int cxk_factor_async(cxk_context* ctx);
/* factor and, in the same sweeps, solve  y <- K^-1 (cb b + cq AQc + cw AW)  (permuted b, residuals of
 * the last cxk_assemble): one upward pass instead of factor + forward substitution.  (k bs, k cs, -2)
 * gives the Newton direction of cxk_newton_direction, (-bs, cs, 0) the solve of
 * ComputeMuFromDivergence (cone_program.cc:173-214).  LLT flag through cxk_factor_status. */
int cxk_factor_solve_async(cxk_context* ctx, double cb, double cq, double cw);
/* cxk_factor_async + cxk_newton_direction in one upward pass (same right-hand side formula
 * y = k (b bs + AQc cs) - 2 AW, cone_program.cc:409-411) */
int cxk_factor_direction_async(cxk_context* ctx, double k, double bs, double cs);
int cxk_factor_status(cxk_context* ctx, int* ok);
/* enqueue the by / cx / norm reductions of cxk_step_scalars; the next cxk_step_scalars call
 * returns them (waiting only if nothing has waited since) */
int cxk_step_scalars_async(cxk_context* ctx);
/* y_dev = inv_sqrt_mu*(b*b_scaling + AQc*c_scaling) - 2 AW  cone_program.cc:409-411,
 * followed by solver->SolveInPlace(&y) kkt_solver.cc:220-263.  No host sync. */
int cxk_set_cost(cxk_context* ctx, const double* b /* num_vars, host */);
int cxk_newton_direction(cxk_context* ctx, double inv_sqrt_mu, double b_scaling,
                         double c_scaling);
/* y_dev = cb*b + cq*AQc + cw*AW, then SolveInPlace: covers the Newton right-hand side
 * (k*bs, k*cs, -2), the mu-rule solve of ComputeMuFromDivergence cone_program.cc:181
 * (-bs, cs, 0) and the dual-recovery solve cone_program.cc:504 (k*bs, 0, -1).  No host sync. */
int cxk_solve_rhs(cxk_context* ctx, double cb, double cq, double cw);
/* the scalars of one IPM iteration (cone_program.cc:343-357, 439-446), computed on device:
 * out = { b.y, AQc.y, |b|^2, |AQc|^2, <w,c>, <c,Qc> }.  Syncs. */
int cxk_step_scalars(cxk_context* ctx, double* out6);
/* whole BASELINE metric unit: assemble + factor + rhs + solve, asynchronous on the stream;
 * factor status is latched and returned by cxk_sync. */
int cxk_kkt_solve_async(cxk_context* ctx, double inv_sqrt_mu, double b_scaling,
                        double c_scaling);
int cxk_sync(cxk_context* ctx, int* factor_ok);
/* SolveInPlace on a host vector of length N (copy in, solve, copy out). */
int cxk_solve_inplace(cxk_context* ctx, double* y);
/* (sharded context with a communicator: a COLLECTIVE -- a sum all-reduce of N doubles assembles the
 * whole vector, so every rank must call it, in the same order relative to its other cxk_* calls) */
int cxk_get_y(cxk_context* ctx, double* y /* N, original variable order */);
int cxk_set_y(cxk_context* ctx, const double* y);

/* PrepareStep over all constraints (cone_program.h:69-90); info = {normsqrd, norminfd}.
 * Uses the device-resident y.  Syncs (returns two scalars). */
int cxk_prepare_step(cxk_context* ctx, int affine, double c_weight, double e_weight,
                     double* info);
/* cxk_prepare_step (affine = 0) followed by cxk_take_step with the step length of
 * cone_program.cc:417-418, step = min(1, 2 / norminfd^2), evaluated ON THE DEVICE from the norms just
 * reduced -- so TakeStep is enqueued before the host waits for `info` and no idle gap separates the
 * two.  Same arithmetic, same bits as the two calls with the host computing the step.  *took = 1
 * when TakeStep was enqueued; 0 on configurations where a kernel needs the value from the host
 * (sharded contexts, equality constraints, LMI orders beyond LDS): the caller then runs
 * cxk_take_step itself.  Only for iterations whose factorization outcome is already known. */
int cxk_prepare_take_step(cxk_context* ctx, double c_weight, double e_weight, double* info, int* took);
/* TakeStep (cone_program.h:92-97) */
int cxk_take_step(cxk_context* ctx, int affine, double e_weight, double step_size);
/* GetWeightedSlackEigenvalues (cone_program.cc:31-57) on the device-resident y;
 * out = {lambda_min, lambda_max, frobenius_norm_squared, trace} */
int cxk_weighted_slack_eigenvalues(cxk_context* ctx, double c_weight, double* out);

/* ---- The barrier parameter selected on the device: one iteration of conex::Solve without the host
 * round trip between the eigenvalue query and the Newton direction (cone_program.cc:366-413).
 * The eigenvalue query's launch evaluates ComputeMuFromDivergence's rule and the update of
 * inv_sqrt_mu that Solve makes with it (cone_program.cc:166-224, :386-392, divergence.cc:17-110;
 * the same source as the host loop compiles: csrc/mu_rule.h -- the same IEEE operations, the same
 * bits) where the four reduced eigenvalue bounds appear; the Newton direction's right-hand side
 * (:409-411) and PrepareStep's c_weight (:413) then read inv_sqrt_mu from device memory, and the
 * host learns it -- with the factorization flag, the step norms and the by / cx scalars -- from the
 * ONE mailbox at the end of the iteration.  TakeStep, enqueued before the host has seen the
 * factorization's outcome, leaves W alone when the factorization failed.
 * cxk_device_mu_supported: 1 when this program can run that way -- one GPU, Cholesky on the device
 * (not the QR mode, no equality rows), no LMI beyond LDS (its TakeStep takes the step length from the
 * host) -- else 0.  Every cone type takes part: the selection rides in the tail workgroup of the
 * eigenvalue query where all constraints run on the register LMI kernels, in its reduction launch
 * otherwise. */
int cxk_device_mu_supported(cxk_context* ctx);
/* GetWeightedSlackEigenvalues(c_weight) on the device-resident y, then
 * inv_sqrt_mu <- limits(selection > 0 ? selection : prev / 2, lb, ub) on the device.  Does not wait. */
int cxk_select_mu_async(cxk_context* ctx, double c_weight, double divergence_upper_bound, int rank,
                        double prev, double lb, double ub);
/* cxk_newton_direction with k = the device's inv_sqrt_mu */
int cxk_newton_direction_device_mu(cxk_context* ctx, double b_scaling, double c_scaling);
/* The factorization carrying THREE right-hand sides through its one whole-tree launch -- bs b, cs AQc, AW --
 * so that the solve of the mu selection (K^-1 (-bs b + cs AQc), cone_program.cc:181: left in y) and the Newton
 * direction for the mu selected afterwards (K^-1 (k (bs b + cs AQc) - 2 AW), :409-411) are combinations of its
 * three solutions: the cxk_newton_direction_device_mu that follows is 3 N multiply-adds instead of a sweep
 * over the tree -- formed inside the PrepareStep launch that reads the direction, or by one elementwise launch
 * for whoever reads y before that (cxk_get_y, cxk_step_scalars, ...): the same values.  Call order:
 * cxk_assemble, cxk_factor_solve_triple_async, cxk_select_mu_async, cxk_newton_direction_device_mu,
 * cxk_prepare_take_step_device_mu.  cxk_triple_supported: 1 where that applies (the tree in one launch on one
 * GPU, every constraint on the register LMI kernels, the barrier parameter on the device, no refinement)
 * AND the assembly just enqueued still waits to ride in the factorization (ask directly behind cxk_assemble). */
int cxk_triple_supported(cxk_context* ctx);
int cxk_factor_solve_triple_async(cxk_context* ctx, double b_scaling, double c_scaling);
/* cxk_prepare_take_step with c_weight = the device's inv_sqrt_mu * c_scaling; waits, and returns
 * the selected inv_sqrt_mu as well */
int cxk_prepare_take_step_device_mu(cxk_context* ctx, double c_scaling, double e_weight, double* info,
                                    int* took, double* inv_sqrt_mu);

/* ---- inspection (tests, KKTMatrix() kkt_solver.cc:265-269) ------------- */
int cxk_get_slab(cxk_context* ctx, double* out);
int cxk_set_slab(cxk_context* ctx, const double* in);
int cxk_get_constraint_schur(cxk_context* ctx, int i, double* G /* m*m lower */, double* AW,
                             double* AQc, double* scalars /* 2 */);
int cxk_get_residuals(cxk_context* ctx, double* AW /* N */, double* AQc /* N */,
                      double* scalars /* 2 */);

/* ---- multi-GPU (the reference is single process, so no counterpart) ----------
 * After cxk_set_shard(rank, world) + cxk_finalize every rank holds the same symbolic analysis
 * and the same deterministic partition: the elimination tree is cut at a level; everything
 * above the cut (the "top") is replicated, the subtrees below are dealt to ranks (longest
 * processing time first), each constraint lives with the subtree that eliminates it.
 * One KKT solve is then
 *     cxk_kkt_local_async   assemble own constraints, factor + forward-solve own subtrees,
 *                           fold their updates into the partial top, pack the exchange buffer
 *     all-reduce(sum) of the exchange buffer across ranks (RCCL; a few KB: latency bound)
 *     cxk_kkt_finish_async  unpack, factor/solve the top (replicated), back-substitute own
 *                           subtrees
 * after which a rank holds y for its own and for the top variables (cxk_get_valid_variables). */
/* Collectives.  With a communicator attached EVERY entry point of this header works on a sharded
 * context exactly as on a single-GPU one -- cxk_kkt_solve_async, cxk_factor*_async, cxk_solve_rhs,
 * cxk_newton_direction, cxk_solve_inplace (local sweep, ONE sum all-reduce of the top's share,
 * top, back-substitution), cxk_prepare_step / cxk_weighted_slack_eigenvalues / cxk_step_scalars /
 * cxk_line_search (scalar sum / max / min all-reduces), cxk_get_y (the whole vector) -- so the host
 * IPM loop (CONEX_Maximize) runs unchanged, every rank taking the same decisions from the same
 * reduced scalars.  The sums that cross ranks are those of supernodal_assembler.cc:103-111,162-164
 * (separator overlaps into the Schur system) and block_triangular_operations.cc:209-215 (Schur
 * updates of eliminated subtrees).
 *   cxk_comm_unique_id      ncclGetUniqueId on one rank; ship the 128 bytes to the others
 *   cxk_comm_init_rccl      ncclCommInitRank for this context's device (one process per GPU);
 *                           implies cxk_set_shard(rank, world) when called before cxk_finalize.
 *                           librccl.so is loaded on demand: single-GPU users need no RCCL
 *   cxk_comm_set_allreduce  any other transport / tests: in-place all-reduce of `count` doubles of
 *                           DEVICE memory, op 0 sum, 1 max, 2 min; work enqueued on `stream` before
 *                           the call must be seen and the result must be in place on return */
typedef int (*cxk_allreduce_fn)(void* user, double* device_buffer, long count, int op, void* stream);
int cxk_comm_unique_id(void* out128);
int cxk_comm_init_rccl(cxk_context* ctx, const void* unique_id128, int rank, int world_size);
int cxk_comm_set_allreduce(cxk_context* ctx, cxk_allreduce_fn fn, void* user);
/* diagnostic: a ONE-rank RCCL communicator on a context sharded as part of a larger (virtual) world:
 * the sharded step with real ncclAllReduce calls on a single GPU (results: one shard's only) */
int cxk_comm_init_rccl_solo(cxk_context* ctx);
/* sum / max / min all-reduces of `count` doubles through the RCCL communicator, checked */
int cxk_comm_selftest(cxk_context* ctx, int count);
/* ranks of the attached RCCL communicator as RCCL counts them (ncclCommCount); 0 without one */
int cxk_comm_count(const cxk_context* ctx);

/* SupernodalKKTSolver::SetSolverMode (kkt_solver.h:41; SolverConfiguration::kkt_solver): 0 / 1 the
 * supernodal LLT / LDLT (chosen by the structure, as kkt_solver.cc:180-193 does), 2
 * CONEX_QR_FACTORIZATION -- the reference's debugging mode for rank-deficient systems: Factor()
 * forms the dense N x N KKT matrix and takes its column-pivoted Householder QR, every solve goes
 * through it (kkt_solver.cc:175-178, 196-198, 227-231).  As there, this is dense host arithmetic on
 * one core (the assembled slab travels to the host at Factor, right-hand sides at every solve);
 * N is limited to 1500, single GPU. */
int cxk_set_solver_mode(cxk_context* ctx, int mode);

/* ---- per-phase device timers ------------------------------------------------------------
 * The reference brackets Assemble / Factor / Solve / Update of every iteration with START_TIMER /
 * END_TIMER (debug_macros.h:18-52; cone_program.cc:338-341, 359-372, 412-414, 421-437) when built
 * with CONEX_ENABLE_TIMER.  Here the caller marks the START of a phase; a hipEvent is recorded on
 * the context's stream (nothing waits), the device time up to the next mark is charged to that
 * phase, CXK_PHASE_OTHER collects what the reference leaves untimed (mu selection).
 * cxk_phase_read waits for the last mark, folds finished intervals and returns microseconds per
 * phase accumulated since the last reset.  CONEX_Maximize uses them when CONEX_ENABLE_TIMER=1 is
 * in the environment and prints the reference's "Assemble(us): .., Factor(us): .., ..." fields. */
enum { CXK_PHASE_ASSEMBLE = 0, CXK_PHASE_FACTOR = 1, CXK_PHASE_SOLVE = 2, CXK_PHASE_UPDATE = 3,
       CXK_PHASE_OTHER = 4, CXK_PHASE_COUNT = 5 };
int cxk_phase_timers(cxk_context* ctx, int on);
int cxk_phase_mark(cxk_context* ctx, int phase);
int cxk_phase_read(cxk_context* ctx, double* us /* CXK_PHASE_COUNT */, int reset);

/* The two halves of a sharded KKT solve, for callers that run the all-reduce themselves: */
int cxk_exchange_buffer(cxk_context* ctx, void** dev_ptr, long* count /* doubles */);
int cxk_exchange_download(cxk_context* ctx, double* out);   /* host copy, tests */
int cxk_exchange_upload(cxk_context* ctx, const double* in);
int cxk_kkt_local_async(cxk_context* ctx, double inv_sqrt_mu, double b_scaling, double c_scaling);
int cxk_kkt_finish_async(cxk_context* ctx, double inv_sqrt_mu, double b_scaling, double c_scaling);
int cxk_owns_constraint(const cxk_context* ctx, int i);
int cxk_get_valid_variables(const cxk_context* ctx, unsigned char* mask /* N, original order */);
int cxk_shard_info(const cxk_context* ctx, int* cut_level, int* num_levels, long* exchange_count);
/* single-GPU building blocks kept for symmetry with the reference call sequence */
int cxk_assemble_local(cxk_context* ctx);
int cxk_finish_assemble(cxk_context* ctx);

/* per-constraint {normsqrd, norminfd} of the last cxk_prepare_step (the StepInfo info_i of
 * cone_program.h:69-90 before the reduction); diagnostics and tests */
int cxk_get_step_info(cxk_context* ctx, double* out2k);

/* ComputeMuFromLineSearch (cone_program.cc:118-160) with PerformLineSearch / FindMinimumMu of the
 * linear cone (linear_constraint.cc:48-103): two solves with the current factorization (right-hand
 * sides -2 AW and AQc c_s + b b_s - 2 AW), per-row admissible interval of the step, reduced over
 * constraints.  *result = inv_sqrt_mu candidate, or -1 (a cone without line-search support --
 * constraint.h:24-28 -- or an empty interval).  Overwrites y. */
int cxk_line_search(cxk_context* ctx, double dinf_upper_bound, double b_scaling, double c_scaling,
                    double* result);

/* ---- isolated batched fp64 GEMM on the matrix pipe ----------------------------------------
 * C[b] = alpha op(A[b]) op(B[b]) + beta C[b], packed column-major host buffers (A is M x K, or
 * K x M when ta; B is K x N, or N x K when tb).  This is the kernel behind the large-order LMI
 * assembly (Eigen GEMM call sites dense_lmi_constraint.cc:72-103, psd_constraint.cc:13-28,45-84)
 * and the blocked supernode updates (block_triangular_operations.cc:184-219: LLT trailing update
 * and off^T off); exported so tests can check it and bench tools can time it against the fp64
 * MFMA roofline at the supernode sizes SURVEY 8(d) names.  lower_only: SYRK-shaped output
 * (m >= n).  splits > 1: split-K with an ordered reduction.  avg_ms: device time per launch. */
int cxk_gemm_f64(int device, int ta, int tb, int M, int N, int K, int batch, const double* A,
                 const double* B, double* C, double alpha, double beta, int lower_only, int splits,
                 int reps, double* avg_ms);

/* Number of (owned) LMI / Hermitian constraints evaluated from their nonzeros instead of dense
 * matrices (kernels_lmi_sparse.hip.h).  The C-ABI fills matrix inequalities
 * entry by entry (CONEX_UpdateLinearOperator, hermitian_psd.cc:249-275), so their A_i are
 * typically very sparse; cxk_initialize picks the sparse evaluation per constraint when its
 * (sum nnz)^2 / 2 terms cost less than the dense formula 4 n^3 (m+1) + n^2 m^2 (measured
 * break-even ratios in kernels_lmi_sparse.hip.h; CXK_SPARSE_LMI=0/1 in the environment forces
 * never/always).  Results equal the dense path's to rounding. */
int cxk_count_sparse_lmi(const cxk_context* ctx);

/* Number of (owned) LMI / Hermitian constraints whose Schur complement
 * (ConstructSchurComplementSystem, dense_lmi_constraint.cc:72-103) is evaluated by kernel `which`:
 * 0 literal LDS kernel (lmi_schur_generic; also every constraint with non-symmetric data, which
 * the reference accepts and evaluates as written), 1 row-per-lane DPP + MFMA kernel
 * (lmi_schur_fused), 2 persistent MFMA producer/consumer kernel (lmi_schur_mfma), 3 batched GEMM
 * pipeline (orders beyond LDS and mid-size orders), 4 sparse evaluation. */
int cxk_count_lmi_kernel(const cxk_context* ctx, int which);

/* Columns of the dense range at the top of the elimination tree (0 = none): when the last levels
 * hold a supernode of 33..64 columns and at most 64 columns in total, BlockCholeskyInPlace and
 * the block solves (block_triangular_operations.cc:114-219) restricted to those levels run as one
 * dense register factorization (kernels_kkt_top.hip.h) instead of supernode by supernode. */
int cxk_dense_top_columns(const cxk_context* ctx);

/* 1 when cxk_kkt_solve_async folds the assembly (UpdateBlocks / Scatter of
 * supernodal_assembler.cc:93-181 and the right-hand side of cone_program.cc:409-411) into the
 * launch of the first factor level: the leaves of the elimination tree read their panels straight
 * from the Schur blocks, the gather of everything else rides beside them as extra workgroups
 * (tree_factor_level_asm).  Decided per structure at cxk_finalize (Cholesky, the first
 * level one register shape, every supernode in it fed by exactly one constraint); results are the
 * bits of the separate assembly launch.  CXK_NO_FUSED_ASM=1 keeps the separate launch. */
int cxk_fused_assembly(const cxk_context* ctx);
/* 1 when a KKT solve (assembly gather + factorization + solve) runs as ONE launch over the whole
 * elimination tree (tree_fused.hip); CXK_NO_FUSED_TREE=1 in the environment turns it off */
int cxk_fused_tree(const cxk_context* ctx);
/* The whole-tree launch keeps every supernode's wavefront resident and lets it wait, inside the
 * kernel, for its descendants' values: deadlock-free while the launch has the device to itself
 * (grid <= resident slots, workgroups dispatched in index order), every wait bounded.  On a device
 * shared with other streams or processes a wait can run out.  That is not a failed factorization:
 * the context then gives the whole-tree launch up for good and sweeps level by level (kernel
 * boundaries instead of in-kernel waits).  cxk_sync redoes the pending factor-and-solve that way
 * itself; cxk_factor_status reports failure and this function returns 1 ONCE, so that an
 * interior-point loop (program.cc) redoes its iteration instead of giving up. */
int cxk_fused_tree_timed_out(cxk_context* ctx);
/* test hook: pretend the latest whole-tree launch reported a wait that ran out */
int cxk_debug_force_fused_timeout(cxk_context* ctx);

/* ---- timing / roofline accounting -------------------------------------- */
/* algorithmic bytes and flops of one dense-LMI assembly launch (SURVEY 8d formulas) */
int cxk_assembly_work(const cxk_context* ctx, double* bytes, double* flops);
/* average device time (ms) of the dominant assembly kernel since the last reset, measured
 * with hipEvents on the context's stream; returns number of samples */
int cxk_kernel_time(cxk_context* ctx, int reset, double* avg_ms);
/* the same for the other kernels of a Newton step (bench.py's `roofline_tree` and `newton_step`):
 * which = CXK_CLOCK_ASSEMBLY (what cxk_kernel_time reads), _TREE the tree launch(es) of a
 * factor-and-solve, _SOLVE a solve-only sweep, _QUERY the eigenvalue query, _PREPARE, _TAKE.  A slot
 * that is one kernel carries the event pair on its dispatch (the kernel's own begin / end time
 * stamps); a slot of several launches is bracketed and includes their boundaries
 * (DESIGN.md section 6 lists which slot is which at every configuration). */
enum { CXK_CLOCK_ASSEMBLY = 0, CXK_CLOCK_TREE = 1, CXK_CLOCK_SOLVE = 2, CXK_CLOCK_QUERY = 3,
       CXK_CLOCK_PREPARE = 4, CXK_CLOCK_TAKE = 5, CXK_CLOCK_COUNT = 6 };
int cxk_kernel_clock(cxk_context* ctx, int which, int reset, double* avg_ms);
/* on = 0: off; on = 1: bracket every launch of that kernel with a hipEvent pair; on = P > 1:
 * every P-th launch (an event pair costs a few us of stream time, sampling keeps the timed
 * region representative) */
/* SupernodalKKTSolver::SetIterativeRefinementIterations (kkt_solver.h:37, loop kkt_solver.cc:233-261):
 * the next factorization keeps the assembled matrix, and every solve after it runs `iterations`
 * steps y <- y + K^-1 (b - K y) on the device (supernodal mat-vec; the reference multiplies a
 * dense N x N copy).  0 switches it off.  Single GPU. */
int cxk_set_iterative_refinement(cxk_context* ctx, int iterations);

int cxk_enable_timing(cxk_context* ctx, int on);

#ifdef __cplusplus
}
#endif
#endif
