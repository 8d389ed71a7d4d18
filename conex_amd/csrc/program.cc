// CONEX_* outer C-ABI (include/conex.h) and the IPM driver on top of the device-resident
// Newton-step path (cxk_*, include/conex_kkt_hip.h).
//
// Host side only: constraint builders, argument checks, the interior-point control loop and the
// scalar mu rules.  Restates
//   interfaces/conex.cc:20-407        argument checks, copies, status codes
//   conex/cone_program.cc:78-112      Initialize
//   conex/cone_program.cc:166-224     MinimizeNormInf, ComputeMuFromDivergence, ApplyLimits
//   conex/cone_program.cc:235-552     Solve
//   conex/divergence.cc:17-110        DivergenceUpperBoundInverse and helpers
//   conex/linear_constraint.cc:14-46  PreprocessLinearInequality
//   conex/hermitian_psd.cc:249-313, linear_constraint.cc:207-228, soc_constraint.cc:305-335
//                                     UpdateLinearOperator / UpdateAffineTerm checks
// Every fp64 vector/matrix operation of the loop is a cxk_* call; per iteration only the
// scalars of cxk_step_scalars / cxk_prepare_step / cxk_weighted_slack_eigenvalues cross PCIe.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../include/conex.h"
#include "../../include/conex_kkt_hip.h"
#include "mu_rule.h"

namespace {

#define CONEX_DEMAND(x, msg)                                         \
  if (!(x)) {                                                        \
    fprintf(stderr, "%s line %d: %s\n", __FILE__, __LINE__, msg);    \
    return 1;                                                        \
  }

int g_verbose = -1;
bool Verbose() {
  if (g_verbose < 0) {
    const char* e = getenv("CONEX_VERBOSE");
    g_verbose = (e && *e && *e != '0') ? 1 : 0;
  }
  return g_verbose == 1;
}

enum ConeKind { kLmi, kLinear, kSoc, kQuadCost, kEquality, kQuadCone };

struct Cone {
  ConeKind kind = kLmi;
  int order = 0;        // LMI n ; linear rows ; SOC n (vectors in R^{n+1})
  int hyper = 1;        // LMI only: 1, 2, 4, 8
  // true: HermitianPsdConstraint<T> (CONEX_NewLinearMatrixInequality, conex.cc:286-318);
  // false: DenseLMIConstraint (CONEX_Add{Dense,Sparse}LMIConstraint, conex.cc:137-188)
  bool hermitian = false;
  std::vector<int> vars;                   // clique (variable ids) fixed at creation time
  // LMI: mats[v] = hyper planes of n*n (col-major); affine likewise. Linear/SOC: dense A (rows x cols)
  std::vector<std::vector<double>> mats;
  std::vector<double> affine;
  int cols = 0;         // linear / SOC: number of columns currently allocated in `A`
  std::vector<double> A;
  std::vector<double> c;
  std::vector<double> Q;  // kQuadCone: inner-product matrix (order x order), empty = identity
};

struct Program {
  unsigned magic = 0xC0DEC0DEu;
  int num_vars = 0;
  std::vector<Cone> cones;
  bool contains_quadratic_costs = false;
  std::vector<double> linear_cost;
  // solver state
  cxk_context* ctx = nullptr;
  bool initialized = false;
  bool dirty = true;
  int device = 0;
  int reference_identity = -1;  // -1: environment (CXK_REFERENCE_QUIRKS); see CONEX_HIP_SetReferenceIdentity
  // multi-GPU (one process per GPU, every rank builds the same program): see CONEX_HIP_SetCommunicator
  int shard_rank = 0, shard_world = 1;
  bool have_unique_id = false;
  unsigned char unique_id[128] = {0};
  cxk_allreduce_fn allreduce_fn = nullptr;
  void* allreduce_user = nullptr;
  double b_scaling = 1, c_scaling = 1;
  std::vector<double> sqrt_inv_mu;
  int num_iter = 0;
  bool stats_ready = false;
  int solved = 0, primal_infeasible = 0, dual_infeasible = 0;
  std::string why;
  ~Program() {
    if (ctx) cxk_destroy(ctx);
  }
};

// SAFER_CAST_TO_Program interfaces/conex.cc:20-33
#define CAST_PROGRAM(x, prog)                                       \
  CONEX_DEMAND(x, "Program pointer is null.");                      \
  Program* prog = static_cast<Program*>(x);                         \
  CONEX_DEMAND(prog->magic == 0xC0DEC0DEu, "Program corrupted or invalid pointer.");

std::vector<int> AllVars(const Program& p) {
  std::vector<int> v(p.num_vars);
  for (int i = 0; i < p.num_vars; i++) v[i] = i;
  return v;
}

int AddCone(Program* p, Cone&& c) {
  p->cones.push_back(std::move(c));
  p->dirty = true;
  return static_cast<int>(p->cones.size()) - 1;
}

// ---------------------------------------------------------------- divergence.cc: mu_rule.h
using cxk_mu::Wse;

int RankOf(const Cone& c) {
  switch (c.kind) {
    case kLmi: return c.order;
    case kLinear: return c.order;
    case kSoc: return 2;
    case kQuadCone: return 2;  // quadratic_cone_constraint.h:38
    default: return 0;
  }
}

// Upload the builder state into a fresh device context (Initialize, cone_program.cc:78-112).
int BuildContext(Program* p) {
  if (p->ctx) {
    cxk_destroy(p->ctx);
    p->ctx = nullptr;
  }
  if (cxk_create(p->num_vars, p->device, nullptr, &p->ctx) != CXK_SUCCESS) {
    fprintf(stderr, "conex: no HIP device available; this library has no CPU fallback.\n");
    return 1;
  }
  if (p->reference_identity >= 0) cxk_set_reference_identity(p->ctx, p->reference_identity);
  if (p->shard_world > 1) {
    if (cxk_set_shard(p->ctx, p->shard_rank, p->shard_world)) return 1;
    if (p->allreduce_fn) {
      if (cxk_comm_set_allreduce(p->ctx, p->allreduce_fn, p->allreduce_user)) return 1;
    } else if (p->have_unique_id) {
      if (cxk_comm_init_rccl(p->ctx, p->unique_id, p->shard_rank, p->shard_world)) return 1;
    } else {
      fprintf(stderr, "conex: a sharded program needs CONEX_HIP_SetCommunicator or CONEX_HIP_SetAllReduce\n");
      return 1;
    }
  }
  for (const Cone& c : p->cones) {
    int id = -1;
    const int m = static_cast<int>(c.vars.size());
    switch (c.kind) {
      case kLmi: {
        if (c.hermitian) {
          const size_t sz = (size_t)c.order * c.order * c.hyper;
          std::vector<double> A((size_t)m * sz, 0.0), C(sz, 0.0);
          for (int v = 0; v < m && v < (int)c.mats.size(); v++)
            if (!c.mats[v].empty()) std::copy(c.mats[v].begin(), c.mats[v].begin() + sz, A.begin() + v * sz);
          if (!c.affine.empty()) std::copy(c.affine.begin(), c.affine.begin() + sz, C.begin());
          id = cxk_add_hermitian(p->ctx, c.order, c.hyper, m, A.data(), C.data(), c.vars.data());
          break;
        }
        const size_t nn = (size_t)c.order * c.order;
        std::vector<double> A((size_t)m * nn, 0.0), C(nn, 0.0);
        for (int v = 0; v < m && v < (int)c.mats.size(); v++)
          if (!c.mats[v].empty()) std::copy(c.mats[v].begin(), c.mats[v].begin() + nn, A.begin() + v * nn);
        if (!c.affine.empty()) std::copy(c.affine.begin(), c.affine.begin() + nn, C.begin());
        id = cxk_add_lmi(p->ctx, c.order, m, A.data(), C.data(), c.vars.data());
        break;
      }
      case kLinear: {
        std::vector<double> A((size_t)c.order * m, 0.0);
        for (int j = 0; j < m && j < c.cols; j++)
          std::copy(c.A.begin() + (size_t)j * c.order, c.A.begin() + (size_t)(j + 1) * c.order,
                    A.begin() + (size_t)j * c.order);
        id = cxk_add_linear(p->ctx, c.order, m, A.data(), c.c.data(), c.vars.data());
        break;
      }
      case kSoc: {
        const int len = c.order + 1;
        std::vector<double> A((size_t)len * m, 0.0), cc(len, 0.0);
        for (int j = 0; j < m && j < c.cols; j++)
          std::copy(c.A.begin() + (size_t)j * len, c.A.begin() + (size_t)(j + 1) * len,
                    A.begin() + (size_t)j * len);
        std::copy(c.c.begin(), c.c.end(), cc.begin());
        id = cxk_add_soc(p->ctx, c.order, m, A.data(), cc.data(), c.vars.data());
        break;
      }
      case kQuadCost:
        id = cxk_add_static(p->ctx, m, c.A.data(), c.vars.data());
        break;
      case kQuadCone:
        id = cxk_add_quadratic(p->ctx, c.order, m, c.Q.empty() ? nullptr : c.Q.data(), c.A.data(), c.c.data(),
                               c.vars.data());
        break;
      case kEquality: {
        std::vector<double> A((size_t)c.order * m, 0.0);
        for (int j = 0; j < m && j < c.cols; j++)
          std::copy(c.A.begin() + (size_t)j * c.order, c.A.begin() + (size_t)(j + 1) * c.order,
                    A.begin() + (size_t)j * c.order);
        id = cxk_add_equality(p->ctx, c.order, m, A.data(), c.c.data(), c.vars.data());
        break;
      }
    }
    CONEX_DEMAND(id >= 0, "constraint rejected while building the device program");
  }
  CONEX_DEMAND(cxk_finalize(p->ctx) == CXK_SUCCESS, cxk_last_error(p->ctx));
  p->dirty = false;
  return 0;
}

using cxk_mu::ApplyLimits;

#define REPORT(name, val) \
  if (Verbose()) printf(#name ": %.2e, ", (double)(val));
#define PRINTSTATUS(x) \
  if (Verbose()) printf("Status: %s\n\n", x);

struct Config {  // conex::SolverConfiguration
  int prepare_dual_variables, initialization_mode;
  double inv_sqrt_mu_max, minimum_mu, maximum_mu, divergence_upper_bound;
  int enable_line_search;
  double dinf_upper_bound;
  int final_centering_steps;
  double final_centering_tolerance;
  int initial_centering_steps_warmstart, initial_centering_steps_coldstart;
  double warmstart_abort_threshold;
  int max_iterations;
  double infeasibility_threshold, kkt_error_tolerance;
  int kkt_solver, enable_rescaling, iterative_refinement_iterations;
};

Config FromApi(const CONEX_SolverConfiguration* c) {  // interfaces/conex.cc:65-90
  Config o;
  o.prepare_dual_variables = c->prepare_dual_variables;
  o.initialization_mode = c->initialization_mode;
  o.inv_sqrt_mu_max = c->inv_sqrt_mu_max;
  o.minimum_mu = c->minimum_mu;
  o.maximum_mu = c->maximum_mu;
  o.divergence_upper_bound = c->divergence_upper_bound;
  o.enable_line_search = c->enable_line_search;
  o.dinf_upper_bound = c->dinf_upper_bound;
  o.final_centering_steps = c->final_centering_steps;
  o.final_centering_tolerance = c->final_centering_tolerance;
  o.initial_centering_steps_warmstart = c->initial_centering_steps_warmstart;
  o.initial_centering_steps_coldstart = c->initial_centering_steps_coldstart;
  o.warmstart_abort_threshold = c->warmstart_abort_threshold;
  o.max_iterations = c->max_iterations;
  o.iterative_refinement_iterations = c->iterative_refinement_iterations;
  o.infeasibility_threshold = c->infeasibility_threshold;
  o.kkt_error_tolerance = c->kkt_error_tolerance;
  o.enable_rescaling = c->enable_rescaling;
  o.kkt_solver = c->kkt_solver;
  return o;
}

// ComputeMuFromDivergence cone_program.cc:173-214
int MuFromDivergence(Program* p, const Config& cfg, int rankK, double* out, bool solved_already = false) {
  cxk_context* ctx = p->ctx;
  const double bs = p->b_scaling, cs = p->c_scaling;
  // y = AQc*cs - b*bs ; SolveInPlace (already done when the factorization carried this right-hand side)
  if (!solved_already && cxk_solve_rhs(ctx, -bs, cs, 0.0)) return 1;
  double e4[4];
  if (cxk_weighted_slack_eigenvalues(ctx, cs, e4)) return 1;
  Wse mp;
  mp.lmin = e4[0];
  mp.lmax = e4[1];
  mp.frob = e4[2];
  mp.trace = e4[3];
  const double inv = cxk_mu::SelectFromDivergence(cfg.divergence_upper_bound, rankK, mp);
  *out = inv;
  return 0;
}

// conex::Solve cone_program.cc:235-533. Returns the solved flag (1 = solved).
int SolveProgram(Program* p, const Config& cfg, double* yout) {
  if (p->contains_quadratic_costs && !(cfg.enable_line_search && !cfg.enable_rescaling)) {
    fprintf(stderr, "%s line %d: %s\n", __FILE__, __LINE__,
            "Must enable line search and disable rescaling for problems with quadratic costs.");
    return 1;  // the reference's CONEX_DEMAND returns 1 here too
  }
  const int m = p->num_vars;
  p->solved = 0;
  p->primal_infeasible = 0;
  p->dual_infeasible = 0;
  bool max_iters_reached = true;
  if (p->cones.empty()) {  // empty program :265-270
    for (int i = 0; i < m; i++) yout[i] = -p->linear_cost[i] * std::numeric_limits<double>::infinity();
    return 0;
  }
  // CONEX_PROFILE=1 in the environment: wall time of set-up, of the iteration loop and of the
  // host side of each device call on stderr
  const bool profile = getenv("CONEX_PROFILE") != nullptr;
  double prof_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static const char* const prof_name[8] = {"assemble", "factor_async", "mu selection", "factor_status",
                                           "newton_direction", "prepare_step", "step_scalars", "take_step"};
#define TIMED(slot, expr)                                                                  \
  [&]() {                                                                                  \
    if (!profile) return (expr);                                                           \
    const auto t0_ = std::chrono::steady_clock::now();                                     \
    const auto r_ = (expr);                                                                \
    prof_ms[slot] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count(); \
    return r_;                                                                             \
  }()
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  // Initialize :78-112
  if (!p->initialized || p->dirty || cfg.initialization_mode == 0) {
    if (p->dirty || !p->ctx) {
      // "Sparsity Analysis(us)" of cone_program.cc:95-109: symbolic analysis + upload, host wall time
      const auto t_sp = std::chrono::steady_clock::now();
      if (BuildContext(p)) return 0;
      if (getenv("CONEX_ENABLE_TIMER") && atoi(getenv("CONEX_ENABLE_TIMER")) != 0)
        printf("Sparsity Analysis(us): %.0f, \n",
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_sp).count());
    }
    if (cfg.initialization_mode == 0) {
      p->b_scaling = 1;
      p->c_scaling = 1;
      if (cxk_set_identity(p->ctx)) return 0;
    }
    p->initialized = true;
  }
  cxk_context* ctx = p->ctx;
  // CONEX_ENABLE_TIMER=1 (the reference's compile-time macro of debug_macros.h:18-52 as an
  // environment switch): device time of the four phases the reference brackets, per iteration
  const bool timers = getenv("CONEX_ENABLE_TIMER") != nullptr && atoi(getenv("CONEX_ENABLE_TIMER")) != 0;
  cxk_phase_timers(ctx, timers ? 1 : 0);
  if (cxk_set_solver_mode(ctx, cfg.kkt_solver)) return 0;  // solver->SetSolverMode(config.kkt_solver) :305
  double phase_prev[CXK_PHASE_COUNT] = {0, 0, 0, 0, 0};
  if (timers) {  // start this solve's phase totals from zero
    cxk_phase_read(ctx, phase_prev, 1);
    for (int k = 0; k < CXK_PHASE_COUNT; k++) phase_prev[k] = 0;
  }
  // solver.SetIterativeRefinementIterations(config.iterative_refinement_iterations)
  if (cxk_set_iterative_refinement(ctx, cfg.iterative_refinement_iterations > 0 ? cfg.iterative_refinement_iterations : 0))
    return 0;
  if (profile) fprintf(stderr, "conex profile: set-up (symbolic analysis, plans, upload) %.2f ms\n", since(t_start));
  const auto t_loop = std::chrono::steady_clock::now();
  p->sqrt_inv_mu.assign(std::max(cfg.max_iterations, 1), 0.0);
  p->num_iter = 0;
  p->stats_ready = true;
  if (Verbose()) printf("\n");

  const int N = cxk_system_size(ctx);
  std::vector<double> bin(m);
  for (int i = 0; i < m; i++) bin[i] = -p->linear_cost[i];
  if (cxk_set_cost(ctx, bin.data())) return 0;

  double inv_sqrt_mu_max = cfg.inv_sqrt_mu_max;
  double cx = 1, by = -1, kkt_error = 0;
  double inv_sqrt_mu = 0, e_weight = 1, c_weight = 0, step_size = 1;
  int rankK = 0;
  for (const Cone& c : p->cones) rankK += RankOf(c);
  int centering_steps = 0;
  bool warmstart_aborted = false;
  int initial_centering_steps = cfg.initial_centering_steps_coldstart;
  int initial_centering = 1;
  double& c_scaling = p->c_scaling;
  double& b_scaling = p->b_scaling;
  if (cfg.initialization_mode) {
    PRINTSTATUS("Warmstarting...");
    initial_centering_steps = cfg.initial_centering_steps_warmstart;
  }

  for (int i = 0; i < cfg.max_iterations; i++) {
    // (state an iteration changes before it knows its factorization's outcome: restored by a redo)
    const double redo_mu = inv_sqrt_mu, redo_mu_max = inv_sqrt_mu_max, redo_bs = b_scaling, redo_cs = c_scaling;
    const int redo_centering = centering_steps;
    const bool redo_aborted = warmstart_aborted;
#define REDO_ITERATION()                  \
  {                                       \
    inv_sqrt_mu = redo_mu;                \
    inv_sqrt_mu_max = redo_mu_max;        \
    b_scaling = redo_bs;                  \
    c_scaling = redo_cs;                  \
    centering_steps = redo_centering;     \
    warmstart_aborted = redo_aborted;     \
    i--;                                  \
    continue;                             \
  }
    if (i >= initial_centering_steps) initial_centering = 0;
    if (Verbose()) printf(i < 10 ? "i:  %d, " : "i: %d, ", i);
    const bool final_centering = (inv_sqrt_mu >= inv_sqrt_mu_max) ||
                                 (kkt_error > cfg.kkt_error_tolerance) ||
                                 i >= (cfg.max_iterations - cfg.final_centering_steps);
    const bool update_mu = (i == 0) || !(initial_centering || final_centering) || warmstart_aborted;
    warmstart_aborted = false;
    if (final_centering) {
      if (centering_steps >= cfg.final_centering_steps) {
        max_iters_reached = (i >= cfg.max_iterations - 1);
        break;
      }
    }
    cxk_phase_mark(ctx, CXK_PHASE_ASSEMBLE);
    if (TIMED(0, cxk_assemble(ctx))) return 0;
    if (i < 1 && cfg.enable_rescaling) {
      if (cfg.initialization_mode == 0) {
        double sc[6];
        if (cxk_step_scalars(ctx, sc)) return 0;
        b_scaling = 1.0 / (1 + std::sqrt(sc[2]));
        c_scaling = 1.0 / (1 + std::sqrt(sc[3]));
      }
      double mu_target = 1.0 / (inv_sqrt_mu_max * inv_sqrt_mu_max);
      mu_target *= (b_scaling * c_scaling);
      inv_sqrt_mu_max = 1.0 / std::sqrt(mu_target);
    }
    // solver->Factor() :360.  The LLT flag travels back with the next host round trip of this
    // iteration (mu selection or PrepareStep); everything enqueued in between only overwrites
    // scratch state (y, the step temporaries) -- the PrepareStep kernels of the cones that keep
    // w^{1/2} in W (second-order, quadratic) and every TakeStep kernel read the flag on the device
    // and leave W alone when it is set -- so acting on the flag there is equivalent.
    // One upward pass serves the factorization and the first solve of the iteration: the
    // right-hand side of the mu selection (ComputeMuFromDivergence) when mu is updated without a
    // line search, the Newton direction itself when mu stays (its value is final before Factor()).
    const bool fuse_mu_solve = update_mu && !cfg.enable_line_search;
    const bool fuse_direction = !update_mu;
    cxk_phase_mark(ctx, CXK_PHASE_FACTOR);  // (a fused first solve of the iteration rides in the factor sweep)
    if (fuse_direction) {
      if (initial_centering == 0) centering_steps++;
      ApplyLimits(&inv_sqrt_mu, std::sqrt(1.0 / (1e-15 + cfg.maximum_mu)), inv_sqrt_mu_max);
    }
    // (the barrier parameter selected on the device, below: known before the factorization is enqueued, so
    // that it can carry the three right-hand sides the Newton direction for ANY mu is a combination of)
    const bool mu_on_device = fuse_mu_solve && !p->contains_quadratic_costs && !timers &&
                              !(i == 0 && cfg.initialization_mode == 1) && cxk_device_mu_supported(ctx) == 1;
    if (fuse_mu_solve && mu_on_device && cxk_triple_supported(ctx) == 1) {
      if (TIMED(1, cxk_factor_solve_triple_async(ctx, b_scaling, c_scaling))) return 0;
    } else if (fuse_mu_solve) {
      if (TIMED(1, cxk_factor_solve_async(ctx, -b_scaling, c_scaling, 0.0))) return 0;
    } else if (fuse_direction) {
      if (TIMED(1, cxk_factor_direction_async(ctx, inv_sqrt_mu, b_scaling, c_scaling))) return 0;
    } else {
      if (TIMED(1, cxk_factor_async(ctx))) return 0;
    }
    enum { kOk, kRetry, kFailed, kRedo };
    auto factor_outcome = [&]() -> int {
      int ok = 0;
      if (TIMED(3, cxk_factor_status(ctx, &ok))) return kFailed;
      if (ok) return kOk;
      // a wait inside the whole-tree launch ran out (shared device): nothing wrong with the matrix,
      // the context has switched to its level kernels -- the same iteration again (W is untouched:
      // every kernel that would have changed it looked at the failure flag first)
      if (cxk_fused_tree_timed_out(ctx) == 1) return kRedo;
      if (i == 0 && cfg.initialization_mode == 1) {
        PRINTSTATUS("Aborting warmstart...");
        cxk_set_identity(ctx);
        warmstart_aborted = true;
        return kRetry;
      }
      p->solved = 0;
      PRINTSTATUS("Factorization failed.");
      return kFailed;
    };
    // The selection of mu on the device (cxk_select_mu_async): the eigenvalue query's launch evaluates
    // the rule below itself, the Newton direction and PrepareStep read inv_sqrt_mu from device memory,
    // and the host learns it, with everything else, from the one mailbox at the end of the iteration.
    const double mu_lb = std::sqrt(1.0 / (1e-15 + cfg.maximum_mu));
    if (mu_on_device) {
      cxk_phase_mark(ctx, CXK_PHASE_OTHER);
      if (TIMED(2, cxk_select_mu_async(ctx, c_scaling, cfg.divergence_upper_bound, rankK, inv_sqrt_mu, mu_lb,
                                       inv_sqrt_mu_max)))
        return 0;
    } else if (update_mu) {
      cxk_phase_mark(ctx, CXK_PHASE_OTHER);  // mu selection: untimed in the reference
      double temp = -1;
      if (cfg.enable_line_search) {  // cone_program.cc:376-384
        if (cxk_line_search(ctx, cfg.dinf_upper_bound, b_scaling, c_scaling, &temp)) return 0;
        if (temp < 0) temp = inv_sqrt_mu;
      }
      if (temp < 0) {
        if (p->contains_quadratic_costs) {
          fprintf(stderr, "%s line %d: Solver terminating with error: line-search failed.\n", __FILE__,
                  __LINE__);
          return 1;
        }
        if (TIMED(2, MuFromDivergence(p, cfg, rankK, &temp, fuse_mu_solve))) return 0;
      }
      {
        const int fo = factor_outcome();  // free: the mu selection above has waited for the stream
        if (fo == kRedo) REDO_ITERATION();
        if (fo == kRetry) continue;
        if (fo == kFailed) return 0;
      }
      if (temp > 0)
        inv_sqrt_mu = temp;
      else
        inv_sqrt_mu *= .5;
    }
    if (!mu_on_device) ApplyLimits(&inv_sqrt_mu, mu_lb, inv_sqrt_mu_max);  // (on the device: part of the rule)

    cxk_phase_mark(ctx, CXK_PHASE_SOLVE);
    if (mu_on_device) {
      if (TIMED(4, cxk_newton_direction_device_mu(ctx, b_scaling, c_scaling))) return 0;
    } else if (!fuse_direction && TIMED(4, cxk_newton_direction(ctx, inv_sqrt_mu, b_scaling, c_scaling))) {
      return 0;
    }
    if (TIMED(6, cxk_step_scalars_async(ctx))) return 0;  // by / cx of :439-446 need y only: same round trip
    e_weight = 1;
    c_weight = inv_sqrt_mu * c_scaling;
    double info[2];
    cxk_phase_mark(ctx, CXK_PHASE_UPDATE);
    // When the factorization's outcome is already known (mu was updated: the flag came back with the
    // mu selection) and no warm-start check is pending, TakeStep is enqueued behind PrepareStep with
    // the step rule below evaluated on the device: the host round trip no longer separates them.
    int took_step = 0;
    const bool step_on_device = update_mu && !(i == 0 && cfg.initialization_mode == 1);
    if (mu_on_device) {
      // (the first host round trip of this iteration: the factorization's outcome arrives with it)
      if (TIMED(5, cxk_prepare_take_step_device_mu(ctx, c_scaling, e_weight, info, &took_step, &inv_sqrt_mu)))
        return 0;
      c_weight = inv_sqrt_mu * c_scaling;
      const int fo = factor_outcome();
      if (fo == kRedo) REDO_ITERATION();
      if (fo == kRetry) continue;
      if (fo == kFailed) return 0;
    } else if (step_on_device) {
      if (TIMED(5, cxk_prepare_take_step(ctx, c_weight, e_weight, info, &took_step))) return 0;
    } else if (TIMED(5, cxk_prepare_step(ctx, 0, c_weight, e_weight, info))) {
      return 0;
    }
    if (!update_mu) {
      const int fo = factor_outcome();
      if (fo == kRedo) REDO_ITERATION();
      if (fo == kRetry) continue;
      if (fo == kFailed) return 0;
    }
    step_size = 2.0 / (info[1] * info[1]);
    if (step_size > 1) step_size = 1;
    double sc[6];
    if (TIMED(6, cxk_step_scalars(ctx, sc))) return 0;
    if (i == 0 && cfg.initialization_mode == 1 && info[1] >= cfg.warmstart_abort_threshold) {
      PRINTSTATUS("Aborting warmstart...");
      cxk_set_identity(ctx);
      warmstart_aborted = true;
    } else if (!took_step) {
      if (TIMED(7, cxk_take_step(ctx, 0, e_weight, step_size))) return 0;
    }
    cxk_phase_mark(ctx, CXK_PHASE_OTHER);
    if (timers) {  // "Assemble(us): 12, Factor(us): 40, ..." as the reference's END_TIMER prints them
      double us[CXK_PHASE_COUNT];
      if (cxk_phase_read(ctx, us, 0) == CXK_SUCCESS) {
        static const char* const names[4] = {"Assemble", "Factor", "Solve", "Update"};
        for (int k = 0; k < 4; k++) printf("%s(us): %.0f, ", names[k], us[k] - phase_prev[k]);
        for (int k = 0; k < CXK_PHASE_COUNT; k++) phase_prev[k] = us[k];
      }
    }
    const double d_2 = std::sqrt(std::fabs(info[0]));
    const double d_inf = std::fabs(info[1]);
    by = sc[0] * 1.0 / (inv_sqrt_mu * c_scaling);
    cx = 2 * sc[4] + sc[1] - inv_sqrt_mu * sc[5] * c_scaling;
    cx /= (inv_sqrt_mu * b_scaling);
    double mu = 1.0 / inv_sqrt_mu;
    mu *= mu;
    const double s_dot_x = mu * (rankK - d_2 * d_2) / (b_scaling * c_scaling);
    mu = mu / (c_scaling * b_scaling);
    REPORT(mu, mu);
    REPORT(d_2, d_2);
    REPORT(d_inf, d_inf);
    if (!p->contains_quadratic_costs) {
      REPORT(by, by);
      REPORT(cx, cx);
      kkt_error = std::fabs(cx - by - s_dot_x) / s_dot_x;
      REPORT(kkt_error, kkt_error);
    }
    p->num_iter = i + 1;
    p->sqrt_inv_mu[i] = inv_sqrt_mu;
    if (Verbose()) printf("\n");
    if (final_centering || inv_sqrt_mu >= inv_sqrt_mu_max) {
      if (d_inf <= cfg.final_centering_tolerance) {
        max_iters_reached = false;
        break;
      }
    }
  }
  if (profile) fprintf(stderr, "conex profile: loop left after %.2f ms\n", since(t_loop));
  std::vector<double> y(N, 0.0);
  if (cxk_get_y(ctx, y.data())) return 0;
  for (int i = 0; i < m; i++) yout[i] = y[i];
  if (profile) {
    fprintf(stderr, "conex profile: %d iterations in %.2f ms (%.3f ms each)\n", p->num_iter, since(t_loop),
            since(t_loop) / std::max(p->num_iter, 1));
    for (int k = 0; k < 8; k++)
      fprintf(stderr, "conex profile:   host time in %-18s %.3f ms per iteration\n", prof_name[k],
              prof_ms[k] / std::max(p->num_iter, 1));
  }
#undef TIMED

  double mu = 1.0 / inv_sqrt_mu;
  mu *= mu;
  if (mu > cfg.infeasibility_threshold) {
    PRINTSTATUS("Infeasible Or Unbounded!!.");
    p->solved = 0;
    p->primal_infeasible = cx * inv_sqrt_mu <= -.5;
    p->dual_infeasible = by * inv_sqrt_mu >= .5;
  } else {
    p->solved = 1;
  }
  if (cfg.prepare_dual_variables) {  // :500-516
    int ok = 0;
    if (cxk_assemble(ctx) || cxk_factor(ctx, &ok)) return 0;
    if (cxk_solve_rhs(ctx, inv_sqrt_mu * b_scaling, 0.0, -1.0)) return 0;
    double info[2];
    if (cxk_prepare_step(ctx, 1, 0.0, 0.0, info)) return 0;
  }
  if (p->solved) {
    for (int i = 0; i < m; i++) yout[i] /= inv_sqrt_mu;
    for (int i = 0; i < m; i++) yout[i] /= c_scaling;
  }
  if (p->solved) {
    if (max_iters_reached) {
      p->solved = 0;
      PRINTSTATUS("Terminating at maximum iteration limit.");
    } else {
      PRINTSTATUS("Solved.");
    }
  }
  return p->solved;
}

}  // namespace

// =========================================================================== C-ABI
extern "C" {

void* CONEX_CreateConeProgram() { return new Program(); }

void CONEX_DeleteConeProgram(void* prog) { delete static_cast<Program*>(prog); }

CONEX_STATUS CONEX_SetNumberOfVariables(void* x, int number_of_variables) {
  CONEX_DEMAND(number_of_variables >= 1, "Number of variables must be > 0.");
  CAST_PROGRAM(x, p);
  CONEX_DEMAND(p->num_vars == 0, "Number of variables already set.");
  p->num_vars = number_of_variables;
  p->linear_cost.assign(number_of_variables, 0.0);
  p->dirty = true;
  return CONEX_SUCCESS;
}

int CONEX_AddDenseLMIConstraint(void* x, const double* A, int Ar, int Ac, int m, const double* c,
                                int cr, int cc) {
  Program* p = static_cast<Program*>(x);
  if (!p || Ar != Ac || Ar != cr || cc != cr || !A || !c) return -1;
  if (p->num_vars == 0) {  // Program(0) + dense constraint: the reference sizes by the clique
    p->num_vars = m;
    p->linear_cost.assign(m, 0.0);
  }
  Cone k;
  k.kind = kLmi;
  k.order = cc;
  k.hyper = 1;
  k.vars = AllVars(*p);
  const size_t nn = (size_t)Ar * Ac;
  for (int i = 0; i < m; i++) k.mats.emplace_back(A + i * nn, A + (i + 1) * nn);
  k.affine.assign(c, c + nn);
  return AddCone(p, std::move(k));
}

int CONEX_AddSparseLMIConstraint(void* x, const double* A, int Ar, int Ac, int num_vars,
                                 const double* c, int cr, int cc, const long* vars, int vars_rows) {
  Program* p = static_cast<Program*>(x);
  if (!p || Ar != Ac || Ar != cr || cc != cr || vars_rows != num_vars || !A || !c || !vars) return -1;
  Cone k;
  k.kind = kLmi;
  k.order = cc;
  k.hyper = 1;
  std::vector<char> seen(std::max(p->num_vars, 1), 0);
  for (int i = 0; i < num_vars; i++) {  // IsUnique constraint_manager.h:11-24
    const long v = vars[i];
    if (v < 0 || v >= p->num_vars || seen[v]++) return -1;
    k.vars.push_back(static_cast<int>(v));
  }
  const size_t nn = (size_t)Ar * Ac;
  for (int i = 0; i < num_vars; i++) k.mats.emplace_back(A + i * nn, A + (i + 1) * nn);
  k.affine.assign(c, c + nn);
  return AddCone(p, std::move(k));
}

int CONEX_AddDenseLinearConstraint(void* x, const double* A, int Ar, int Ac, const double* c,
                                   int cr) {
  Program* p = static_cast<Program*>(x);
  if (!p || Ar != cr || !A || !c) return -1;
  if (p->num_vars == 0) {
    p->num_vars = Ac;
    p->linear_cost.assign(Ac, 0.0);
  }
  Cone k;
  k.kind = kLinear;
  k.order = Ar;
  k.cols = Ac;
  k.vars = AllVars(*p);
  k.A.assign(A, A + (size_t)Ar * Ac);
  k.c.assign(c, c + Ar);
  return AddCone(p, std::move(k));
}

int CONEX_AddLinearInequalities(void* x, const double* A, int Ar, int Ac, const double* lb,
                                int num_lb, const double* ub, int num_ub) {
  Program* p = static_cast<Program*>(x);
  if (!p || Ar != num_lb || Ar != num_ub) return -1;
  // PreprocessLinearInequality linear_constraint.cc:14-46
  std::vector<std::vector<double>> rows, eq_rows;
  std::vector<double> rhs, eq_rhs;
  for (int i = 0; i < Ar; i++) {
    double n2 = 0;
    for (int j = 0; j < Ac; j++) n2 += A[i + (size_t)j * Ar] * A[i + (size_t)j * Ar];
    if (lb[i] == ub[i]) {  // equality row: scaled like the others, goes to EqualityConstraints
      const double scale = 1.0 / std::sqrt(n2 + ub[i] * ub[i]);
      std::vector<double> r(Ac);
      for (int j = 0; j < Ac; j++) r[j] = scale * A[i + (size_t)j * Ar];
      eq_rows.push_back(r);
      eq_rhs.push_back(scale * ub[i]);
      continue;
    }
    if (ub[i] < 1e8) {
      const double scale = 1.0 / std::sqrt(n2 + ub[i] * ub[i]);
      std::vector<double> r(Ac);
      for (int j = 0; j < Ac; j++) r[j] = scale * A[i + (size_t)j * Ar];
      rows.push_back(r);
      rhs.push_back(scale * ub[i]);
    }
    if (lb[i] > -1e8) {
      const double scale = 1.0 / std::sqrt(n2 + lb[i] * lb[i]);
      std::vector<double> r(Ac);
      for (int j = 0; j < Ac; j++) r[j] = -scale * A[i + (size_t)j * Ar];
      rows.push_back(r);
      rhs.push_back(-scale * lb[i]);
    }
  }
  if (!rows.empty()) {
    if (p->num_vars == 0) {
      p->num_vars = Ac;
      p->linear_cost.assign(Ac, 0.0);
    }
    Cone k;
    k.kind = kLinear;
    k.order = static_cast<int>(rows.size());
    k.cols = Ac;
    k.vars = AllVars(*p);
    k.A.assign((size_t)k.order * Ac, 0.0);
    for (int i = 0; i < k.order; i++)
      for (int j = 0; j < Ac; j++) k.A[i + (size_t)j * k.order] = rows[i][j];
    k.c = rhs;
    AddCone(p, std::move(k));
  }
  if (!eq_rows.empty()) {  // program.AddConstraint(EqualityConstraints(Aeq, beq)) conex.cc:209-211
    if (p->num_vars == 0) {
      p->num_vars = Ac;
      p->linear_cost.assign(Ac, 0.0);
    }
    Cone k;
    k.kind = kEquality;
    k.order = static_cast<int>(eq_rows.size());
    k.cols = Ac;
    k.vars = AllVars(*p);
    k.A.assign((size_t)k.order * Ac, 0.0);
    for (int i = 0; i < k.order; i++)
      for (int j = 0; j < Ac; j++) k.A[i + (size_t)j * k.order] = eq_rows[i][j];
    k.c = eq_rhs;
    AddCone(p, std::move(k));
  }
  return -1;  // interfaces/conex.cc:213-214
}

CONEX_STATUS CONEX_AddQuadraticCost(void* x, const double* A, int Ar, int Ac) {
  CAST_PROGRAM(x, p);
  CONEX_DEMAND(A && Ar == Ac, "Quadratic cost matrix must be square.");
  // NonZeroSubMat interfaces/conex.cc:44-64
  std::vector<int> vars;
  for (int i = 0; i < Ar; i++)
    if (A[i + (size_t)i * Ar] > 0) vars.push_back(i);
  const int m = static_cast<int>(vars.size());
  Cone k;
  k.kind = kQuadCost;
  k.vars = vars;
  k.A.assign((size_t)m * m, 0.0);
  for (int r = 0; r < m; r++)
    for (int c = 0; c < m; c++) {
      const double v = (A[vars[r] + (size_t)vars[c] * Ar] + A[vars[c] + (size_t)vars[r] * Ar]) / 2.0;
      k.A[r + (size_t)c * m] = v;
      k.A[c + (size_t)r * m] = v;
    }
  p->contains_quadratic_costs = true;
  AddCone(p, std::move(k));
  return CONEX_SUCCESS;  // AddQuadraticCost returns "failure = false"
}

CONEX_STATUS CONEX_NewQuadraticCost(void* x, int* constraint_id) {
  CONEX_DEMAND(constraint_id, "Received output null pointer.");
  CAST_PROGRAM(x, p);
  Cone k;
  k.kind = kQuadCost;
  k.vars = AllVars(*p);
  k.A.assign((size_t)p->num_vars * p->num_vars, 0.0);
  p->contains_quadratic_costs = true;
  *constraint_id = AddCone(p, std::move(k));
  return CONEX_SUCCESS;
}

CONEX_STATUS CONEX_NewLinearMatrixInequality(void* x, int order, int hyper_complex_dim,
                                             int* constraint_id) {
  CONEX_DEMAND(order >= 1, "Invalid LMI dimensions.");
  CONEX_DEMAND(constraint_id, "Received output null pointer.");
  CONEX_DEMAND(hyper_complex_dim == 1 || hyper_complex_dim == 2 || hyper_complex_dim == 4 ||
                   hyper_complex_dim == 8,
               "Hypercomplex dimension must be 1, 2, 4, or 8.");
  CAST_PROGRAM(x, p);
  if (hyper_complex_dim == 8)
    CONEX_DEMAND(order <= 3, "Order of octonion algebra cannot be greater than 3.");
  // (octonions: no real matrix representation -- a cone type of its own on the device, with the
  // reference's rules for it, hermitian_psd.cc:108-168: kernels_oct.hip.h)
  Cone k;
  k.kind = kLmi;
  k.order = order;
  k.hyper = hyper_complex_dim;
  k.hermitian = true;
  k.vars = AllVars(*p);
  *constraint_id = AddCone(p, std::move(k));
  return CONEX_SUCCESS;
}

CONEX_STATUS CONEX_NewLinearInequality(void* x, int num_rows, int* constraint_id) {
  CONEX_DEMAND(constraint_id, "Received output null pointer.");
  CAST_PROGRAM(x, p);
  CONEX_DEMAND(num_rows >= 1, "Number of rows must be positive.");
  Cone k;
  k.kind = kLinear;
  k.order = num_rows;
  k.cols = p->num_vars;
  k.vars = AllVars(*p);
  k.A.assign((size_t)num_rows * p->num_vars, 0.0);
  k.c.assign(num_rows, 0.0);
  *constraint_id = AddCone(p, std::move(k));
  return CONEX_SUCCESS;
}

CONEX_STATUS CONEX_NewLorentzConeConstraint(void* x, int order, int* constraint_id) {
  CONEX_DEMAND(order >= 1, "Received invalid n. Second order cone must have order (n + 1) >= 2.");
  CONEX_DEMAND(constraint_id, "Received output null pointer.");
  CAST_PROGRAM(x, p);
  Cone k;
  k.kind = kSoc;
  k.order = order;
  k.cols = 0;
  k.vars = AllVars(*p);
  k.c.assign(order + 1, 0.0);
  *constraint_id = AddCone(p, std::move(k));
  return CONEX_SUCCESS;
}

CONEX_STATUS CONEX_UpdateLinearOperator(void* x, int constraint, double value, int variable,
                                        int row, int col, int hyper_complex_dim) {
  CAST_PROGRAM(x, p);
  CONEX_DEMAND(constraint >= 0 && constraint < (int)p->cones.size(), "Invalid Constraint.");
  Cone& k = p->cones[constraint];
  const int dim = hyper_complex_dim;
  switch (k.kind) {
    case kLmi: {  // hermitian_psd.cc:249-275
      CONEX_DEMAND(dim >= 0 && dim < k.hyper, "Complex dimension out of bounds.");
      CONEX_DEMAND(row >= 0 && col >= 0 && row < k.order && col < k.order,
                   "Matrix dimension out of bounds.");
      CONEX_DEMAND(!(value != 0 && row == col && dim > 0),
                   "Imaginary components must be skew-symmetric.");
      CONEX_DEMAND(variable >= 0, "Indices cannot be negative.");
      // (hermitian_psd.cc:257-262 means to refuse dim >= 3 of an octonion matrix, but tests
      //  is_same<HermitianPsdConstraint<H>, Octonions>, which never holds: every plane is accepted)
      const size_t nn = (size_t)k.order * k.order;
      if ((int)k.mats.size() <= variable) k.mats.resize(variable + 1);
      if (k.mats[variable].empty()) k.mats[variable].assign(nn * k.hyper, 0.0);
      double* M = k.mats[variable].data() + nn * dim;
      M[row + (size_t)col * k.order] = value;
      M[col + (size_t)row * k.order] = dim == 0 ? value : -value;
      break;
    }
    case kLinear: {  // linear_constraint.cc:207-217
      CONEX_DEMAND(dim == 0, "Complex linear constraints not supported.");
      CONEX_DEMAND(col == 0, "Linear constraint is not matrix valued.");
      CONEX_DEMAND(row < k.order, "Row index out of bounds.");
      CONEX_DEMAND(variable >= 0 && row >= 0, "Indices cannot be negative.");
      CONEX_DEMAND(variable < k.cols, "Variable index out of bounds.");
      k.A[row + (size_t)variable * k.order] = value;
      break;
    }
    case kSoc: {  // soc_constraint.cc:314-324
      CONEX_DEMAND(dim == 0, "Complex second-order cone not supported.");
      CONEX_DEMAND(col == 0, "Second-order constraint is not matrix valued.");
      CONEX_DEMAND(row <= k.order, "Row index out of bounds.");
      CONEX_DEMAND(variable >= 0 && row >= 0, "Indices cannot be negative.");
      const int len = k.order + 1;
      if (variable >= k.cols) {  // ConservativeResizeHelper
        k.A.resize((size_t)len * (variable + 1), 0.0);
        k.cols = variable + 1;
      }
      k.A[row + (size_t)variable * len] = value;
      break;
    }
    case kQuadCost:
    case kEquality:  // constraint.h:13-18 default: not supported
      CONEX_DEMAND(false, "Constraint does not support updates of linear operator.");
  }
  p->dirty = true;
  return CONEX_SUCCESS;
}

CONEX_STATUS CONEX_UpdateAffineTerm(void* x, int constraint, double value, int row, int col,
                                    int hyper_complex_dim) {
  CAST_PROGRAM(x, p);
  CONEX_DEMAND(constraint >= 0 && constraint < (int)p->cones.size(), "Invalid Constraint.");
  Cone& k = p->cones[constraint];
  const int dim = hyper_complex_dim;
  switch (k.kind) {
    case kLmi: {  // hermitian_psd.cc:286-313
      CONEX_DEMAND(dim >= 0 && dim < k.hyper, "Complex dimension out of bounds.");
      CONEX_DEMAND(row >= 0 && col >= 0 && row < k.order && col < k.order,
                   "Matrix dimension out of bounds.");
      CONEX_DEMAND(!(value != 0 && row == col && dim > 0),
                   "Imaginary components must be skew-symmetric.");
      const size_t nn = (size_t)k.order * k.order;
      if (k.affine.empty()) k.affine.assign(nn * k.hyper, 0.0);
      double* M = k.affine.data() + nn * dim;
      M[row + (size_t)col * k.order] = value;
      M[col + (size_t)row * k.order] = dim == 0 ? value : -value;
      break;
    }
    case kLinear:  // linear_constraint.cc:219-228
      CONEX_DEMAND(dim == 0, "Complex linear cone not supported.");
      CONEX_DEMAND(col == 0, "Linear constraint is not matrix valued.");
      CONEX_DEMAND(row < k.order, "Row index out of bounds.");
      CONEX_DEMAND(row >= 0, "Indices cannot be negative.");
      k.c[row] = value;
      break;
    case kSoc:  // soc_constraint.cc:326-335
      CONEX_DEMAND(dim == 0, "Complex second-order cone not supported.");
      CONEX_DEMAND(col == 0, "Second-order constraint is not matrix valued.");
      CONEX_DEMAND(row <= k.order, "Row index out of bounds.");
      CONEX_DEMAND(row >= 0, "Indices cannot be negative.");
      k.c[row] = value;
      break;
    case kQuadCost:  // quadratic_cost.cc:33-39 (dim must be 0: "hyper complex dimension")
      CONEX_DEMAND(dim == 0, "Quadratic cost must be real valued matrix.");
      {
        const int m = static_cast<int>(k.vars.size());
        CONEX_DEMAND(row >= 0 && col >= 0 && row < m && col < m, "Index out of bounds");
        k.A[row + (size_t)col * m] = value;
      }
      break;
    case kEquality:  // constraint.h:20-24 default: not supported
      CONEX_DEMAND(false, "Constraint does not support updates of affine term.");
  }
  p->dirty = true;
  return CONEX_SUCCESS;
}

CONEX_STATUS CONEX_UpdateQuadraticCostMatrix(void* x, int constraint, double value, int row,
                                             int col) {
  return CONEX_UpdateAffineTerm(x, constraint, value, row, col, 0);
}

void CONEX_SetDefaultOptions(CONEX_SolverConfiguration* c) {  // cone_program.h:17-38
  if (c == NULL) {
    fprintf(stderr, "Received null pointer.");
    return;
  }
  c->prepare_dual_variables = 0;
  c->initialization_mode = 0;
  c->inv_sqrt_mu_max = 1000;
  c->minimum_mu = 1e-15;
  c->maximum_mu = 1e4;
  c->divergence_upper_bound = 1;
  c->enable_line_search = 0;
  c->dinf_upper_bound = 1;
  c->final_centering_steps = 5;
  c->final_centering_tolerance = .01;
  c->initial_centering_steps_warmstart = 0;
  c->initial_centering_steps_coldstart = 0;
  c->warmstart_abort_threshold = 2;
  c->max_iterations = 25;
  c->iterative_refinement_iterations = 0;
  c->infeasibility_threshold = 1e5;
  c->kkt_error_tolerance = 1e10;
  c->enable_rescaling = 1;
  c->kkt_solver = 0;
}

int CONEX_Solve(void* x, const CONEX_SolverConfiguration* config, double* y, int yr) {
  Program* p = static_cast<Program*>(x);
  if (!p || !config || !y || yr < p->num_vars) return 0;
  return SolveProgram(p, FromApi(config), y);
}

int CONEX_Maximize(void* x, const double* b, int br, const CONEX_SolverConfiguration* config,
                   double* y, int yr) {
  Program* p = static_cast<Program*>(x);
  if (!p || !config || !y || !b) return 0;
  if (br != p->num_vars) {  // Program::AddLinearCost CONEX_DEMAND
    fprintf(stderr, "%s line %d: %s\n", __FILE__, __LINE__,
            "Cost vector dimension does not equal number of variables");
  }
  // Solve(b, prog, ...): ClearLinearCosts(); AddLinearCost(-b)
  p->linear_cost.assign(p->num_vars, 0.0);
  if (br == p->num_vars)
    for (int i = 0; i < br; i++) p->linear_cost[i] = -b[i];
  if (yr < p->num_vars) return 0;
  return SolveProgram(p, FromApi(config), y);
}

int CONEX_GetDualVariableSize(void* x, int i) {
  Program* p = static_cast<Program*>(x);
  if (!p || i < 0 || i >= (int)p->cones.size()) {
    fprintf(stderr, "%s line %d: Invalid Constraint\n", __FILE__, __LINE__);
    return 1;
  }
  const Cone& k = p->cones[i];
  switch (k.kind) {
    case kLmi: return k.order * k.order;
    case kLinear: return k.order;
    case kSoc: return k.order + 1;
    case kQuadCone: return k.order + 1;
    default: return 0;
  }
}

void CONEX_GetDualVariable(void* x, int i, double* out, int xr, int xc) {
  Program* p = static_cast<Program*>(x);
  if (!p || !p->ctx || !out) return;
  const int n = CONEX_GetDualVariableSize(x, i);
  if (n != xr * xc || n == 0) return;
  if (p->cones[i].kind == kLmi && p->cones[i].hermitian) {
    // the reference exposes the real plane of W only (hermitian_psd.cc:24-29)
    std::vector<double> planes((size_t)n * p->cones[i].hyper);
    if (cxk_get_W(p->ctx, i, planes.data())) return;
    std::copy(planes.begin(), planes.begin() + n, out);
  } else if (cxk_get_W(p->ctx, i, out)) {
    return;
  }
  // Program::GetDualVariable cone_program.h:120-134
  if (!p->primal_infeasible && p->num_iter > 0) {
    const double s = p->sqrt_inv_mu[p->num_iter - 1] * p->b_scaling;
    for (int q = 0; q < n; q++) out[q] /= s;
  }
}

void CONEX_GetIterationStats(void* x, CONEX_IterationStats* stats, int iter_num_circular) {
  if ((x == NULL) || (stats == NULL)) {
    fprintf(stderr, "Received null pointer.");
    return;
  }
  Program* p = static_cast<Program*>(x);
  if (!p->stats_ready) {
    fprintf(stderr, "No statistics available.");
    return;
  }
  int iter_num = iter_num_circular;
  if (iter_num_circular < 0) iter_num = p->num_iter + iter_num_circular;
  if ((p->num_iter <= iter_num) || (iter_num < 0)) {
    fprintf(stderr, "Specified iteration is out of bounds.");
    return;
  }
  stats->mu = 1.0 / (p->sqrt_inv_mu[iter_num] * p->sqrt_inv_mu[iter_num]);
  stats->iteration_number = iter_num;
}

/* not part of conex.h (the reference reaches these cones through its C++ API only):
 *   CONEX_HIP_AddQuadraticConstraint   prog.AddConstraint(QuadraticConstraint(Q, A, c), vars)
 *       c - A y in { (x0, x1) : x0 >= sqrt(x1' Q x1) }   (quadratic_cone_constraint.h:11-86);
 *       Q: n x n column-major or NULL (identity), A: (n + 1) x num_vars column-major, c: n + 1,
 *       vars: num_vars variable ids (NULL: all variables in order)
 *   CONEX_HIP_AddQuadraticCostEpigraph  AddQuadraticCostEpigraph(&prog, Qi, z, epigraph)
 *       (quadratic_cone_constraint.h:88-117): y[epigraph] >= 1/2 y[z]' Qi y[z] as such a cone
 * Both return the constraint id, -1 on invalid arguments. */
int CONEX_HIP_AddQuadraticConstraint(void* x, const double* Q, int n, const double* A, int Ar, int Ac,
                                     const double* c, int cr, const long* vars, int num_vars) {
  Program* p = static_cast<Program*>(x);
  if (!p || p->magic != 0xC0DEC0DEu || n < 1 || !A || !c || Ar != n + 1 || cr != n + 1 || Ac < 1) return -1;
  if (vars ? num_vars != Ac : Ac != p->num_vars) return -1;
  Cone k;
  k.kind = kQuadCone;
  k.order = n;
  k.cols = Ac;
  k.A.assign(A, A + (size_t)(n + 1) * Ac);
  k.c.assign(c, c + n + 1);
  if (Q) k.Q.assign(Q, Q + (size_t)n * n);
  if (vars) {
    for (int i = 0; i < Ac; i++) {
      if (vars[i] < 0 || vars[i] >= p->num_vars) return -1;
      k.vars.push_back((int)vars[i]);
    }
  } else {
    k.vars = AllVars(*p);
  }
  return AddCone(p, std::move(k));
}

int CONEX_HIP_AddQuadraticCostEpigraph(void* x, const double* Qi, int nz, const long* z, long epigraph) {
  if (!Qi || !z || nz < 1) return -1;
  // inner-product matrix Q = diag(1, Qi); (A, b) with b - A (z, t) in L  <=>  t >= 1/2 z' Qi z:
  //   (.5 t + 1)^2 >= (.5 t - 1)^2 + z' Qi z
  const int n = nz + 1, len = nz + 2, cols = nz + 1;
  std::vector<double> Q((size_t)n * n, 0.0), A((size_t)len * cols, 0.0), b((size_t)len, 0.0);
  Q[0] = 1;
  for (int j = 0; j < nz; j++)
    for (int i = 0; i < nz; i++) Q[(size_t)(j + 1) * n + (i + 1)] = Qi[(size_t)j * nz + i];
  A[(size_t)nz * len + 0] = -0.5;  // topRightCorner(2, 1) << -.5, -.5
  A[(size_t)nz * len + 1] = -0.5;
  for (int j = 0; j < nz; j++) A[(size_t)j * len + 2 + j] = 1;  // bottomLeftCorner = I
  b[0] = 1;
  b[1] = -1;
  std::vector<long> vars(z, z + nz);
  vars.push_back(epigraph);
  return CONEX_HIP_AddQuadraticConstraint(x, Q.data(), n, A.data(), len, cols, b.data(), len, vars.data(), cols);
}

/* not part of conex.h: lets a host pick the HIP device ordinal before the first solve */
int CONEX_HIP_SetDevice(void* x, int device) {
  Program* p = static_cast<Program*>(x);
  if (!p) return CONEX_FAILURE;
  p->device = device;
  p->dirty = true;
  return CONEX_SUCCESS;
}

/* not part of conex.h: device microseconds per phase {Assemble, Factor, Solve, Update, other}
 * accumulated by the solves of this program while CONEX_ENABLE_TIMER=1 */
int CONEX_HIP_GetPhaseTimes(void* x, double* us5) {
  Program* p = static_cast<Program*>(x);
  if (!p || !p->ctx || !us5) return CONEX_FAILURE;
  return cxk_phase_read(p->ctx, us5, 0) == CXK_SUCCESS ? CONEX_SUCCESS : CONEX_FAILURE;
}

/* not part of conex.h (bench.py's `newton_step`): hipEvent clocks on the kernels of the program's
 * device context (cxk_enable_timing / cxk_kernel_clock); the context must exist (after a first solve) */
int CONEX_HIP_KernelClocks(void* x, int period) {
  Program* p = static_cast<Program*>(x);
  if (!p || !p->ctx) return CONEX_FAILURE;
  return cxk_enable_timing(p->ctx, period) == CXK_SUCCESS ? CONEX_SUCCESS : CONEX_FAILURE;
}
/* avg_ms[CXK_CLOCK_COUNT], samples[CXK_CLOCK_COUNT] since the last call (the slots are reset) */
int CONEX_HIP_ReadKernelClocks(void* x, double* avg_ms, int* samples) {
  Program* p = static_cast<Program*>(x);
  if (!p || !p->ctx || !avg_ms || !samples) return CONEX_FAILURE;
  if (cxk_sync(p->ctx, nullptr) != CXK_SUCCESS) return CONEX_FAILURE;  // folds the finished event pairs
  for (int k = 0; k < CXK_CLOCK_COUNT; k++) samples[k] = cxk_kernel_clock(p->ctx, k, 1, &avg_ms[k]);
  return CONEX_SUCCESS;
}

/* not part of conex.h: 1 (the default) = the reference as written, 0 = with the two corrections of
 * conex_kkt_hip.h (cxk_set_reference_identity); unset: CXK_REFERENCE_QUIRKS in the environment */
int CONEX_HIP_SetReferenceIdentity(void* x, int on) {
  Program* p = static_cast<Program*>(x);
  if (!p) return CONEX_FAILURE;
  p->reference_identity = on != 0;
  p->dirty = true;
  return CONEX_SUCCESS;
}

/* Multi-GPU, not part of conex.h (the reference is single process).  One process per GPU; every
 * rank builds the SAME program and calls CONEX_Maximize / CONEX_Solve with the same arguments;
 * constraints are dealt to the ranks by elimination subtree, the Schur sums that cross ranks go
 * through one RCCL all-reduce per factorization / solve, the step scalars through small ones
 * (conex_kkt_hip.h, "Collectives").  Every rank returns the same y; a dual variable
 * (CONEX_GetDualVariable) is current on the rank that owns its constraint.
 *   CONEX_HIP_GetUniqueId      128 bytes (ncclGetUniqueId) made by one rank, shipped to all
 *   CONEX_HIP_SetCommunicator  rank / world and the unique id: RCCL over xGMI
 *   CONEX_HIP_SetAllReduce     rank / world and a caller-supplied all-reduce (other transports, tests) */
int CONEX_HIP_GetUniqueId(void* out128) { return cxk_comm_unique_id(out128) == CXK_SUCCESS ? CONEX_SUCCESS : CONEX_FAILURE; }

int CONEX_HIP_SetCommunicator(void* x, const void* unique_id128, int rank, int world_size) {
  Program* p = static_cast<Program*>(x);
  if (!p || !unique_id128 || world_size < 1 || rank < 0 || rank >= world_size) return CONEX_FAILURE;
  p->shard_rank = rank;
  p->shard_world = world_size;
  memcpy(p->unique_id, unique_id128, 128);
  p->have_unique_id = true;
  p->allreduce_fn = nullptr;
  p->dirty = true;
  return CONEX_SUCCESS;
}

int CONEX_HIP_SetAllReduce(void* x, int rank, int world_size, cxk_allreduce_fn fn, void* user) {
  Program* p = static_cast<Program*>(x);
  if (!p || !fn || world_size < 1 || rank < 0 || rank >= world_size) return CONEX_FAILURE;
  p->shard_rank = rank;
  p->shard_world = world_size;
  p->allreduce_fn = fn;
  p->allreduce_user = user;
  p->dirty = true;
  return CONEX_SUCCESS;
}

}  // extern "C"
