/*
 * conex oracle (TEST INFRASTRUCTURE ONLY) -- constraints, Schur assembly,
 * KKT solver facade, cone updates and the IPM driver.
 *
 * Restates:
 *   conex/dense_lmi_constraint.cc:8-103      DenseLMI slack + Schur system
 *   conex/psd_constraint.cc:13-128           GeodesicUpdate/AffineUpdate/PrepareStep/TakeStep/eigs
 *   conex/linear_constraint.cc:108-205       linear cone
 *   conex/soc_constraint.cc:14-303           second-order cone (spin factor)
 *   conex/quadratic_cone_constraint.cc:14-297  Lorentz cone with inner-product matrix Q
 *   conex/supernodal_assembler.cc:23-165     Set/SetLowerTri/Scatter/GetCoeff/UpdateBlocks/Bind
 *   conex/supernodal_solver.h:36-62          DoBind
 *   conex/kkt_solver.cc:133-269              Assemble/Factor/SolveInPlace/KKTMatrix (LLT mode)
 *   conex/constraint_manager.h:11-124        IsUnique/AddConstraint/AssembleSchurComplementResiduals
 *   conex/cone_program.{h,cc}                PrepareStep/TakeStep fan-out, Initialize, Solve
 *   conex/divergence.cc:17-120               mu selection
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conex_oracle.h"
#include "cxo_internal.h"

static int g_verbose = 0;
void cxo_set_verbose(int v) { g_verbose = v; }

/* Reference defect, restated by default.  BindDiagonalBlock (supernodal_assembler.cc:72-91) turns
 * on direct_update -- the constraint's Schur block G aliased onto the supernode's diagonal block
 * -- when the supernode has as many variables as the constraint and their positions inside the
 * constraint are strictly increasing.  It does not check that the first position is 0: with a
 * fill-in variable (position -1, a variable the running-intersection fix assigned to this
 * supernode although the constraint does not contain it) followed by positions 0, 1, .., m-2 the
 * test passes, G lands one row/column off, and the constraint's last variable -- which sits in
 * the separator -- never reaches its place.  The assembled matrix is then NOT sum_c P_c^T G_c P_c.
 * cxo_set_strict_direct_update(1) adds the missing check (positions exactly 0..m-1); tests use it
 * to show that the HIP path, which always scatters by position, equals the corrected reference. */
static int g_strict_direct_update = 0;
void cxo_set_strict_direct_update(int on) { g_strict_direct_update = on; }

typedef struct {
  int type;
  int n;  /* LMI order / linear rows / SOC n (vector in R^{n+1}) */
  int m;  /* number of variables (clique size) */
  double* A;
  double* C;
  /* state */
  double* W;      /* LMI n*n ; linear n ; SOC: W[0]=W0, W[1..n]=W1 */
  double* temp1;  /* LMI n*n ; linear n ; SOC temp1_1 (n) */
  double* temp2;
  double* wa;     /* linear weighted_constraints n*m ; SOC scratch */
  double d0;      /* SOC, quadratic cone */
  double* Q;      /* quadratic cone: n*n or NULL (identity) */
  double* Agram;  /* quadratic cone: A1' Q A1 (m*m), QuadraticConstraintBase::Initialize */
  double wsq_norm_sqr; /* quadratic cone: workspace_.wsqrt_q1_norm_sqr */
  int hd;         /* Hermitian: number of real planes d (1 real, 2 complex, 4 quaternion) */
  /* Schur workspace (newton_step.h:55-109) */
  double* G_own;  /* m*m */
  double* G;      /* == G_own or slab diag block when direct_update */
  double* AW;
  double* AQc;
  double ip_wc;
  double ip_cQc;
  int direct_update;
  int epos;       /* elimination position of this constraint's clique */
} cxo_constraint;

struct cxo_program {
  int num_vars;
  int K;
  int cap;
  cxo_constraint* c;
  ivec* cliques;
  ivec* dual_vars;
  cxo_matrix_data* md;
  cxo_workspace* ws;
  double* sysAW;
  double* sysAQc;
  double sys_wc;
  double sys_cQc;
  double* b_permuted;
  /* SetIterativeRefinementIterations kkt_solver.h; kkt_matrix_ = KKTMatrix() kept by Factor when
   * iterations > 0 (kkt_solver.cc:177-179) */
  int refinement_iterations;
  double* kkt_matrix; /* N*N col-major, original variable order */
  int initialized;
  /* WorkspaceStats workspace.h:73-117 */
  double* sqrt_inv_mu;
  int max_iter;
  int num_iter;
  double b_scaling;
  double c_scaling;
  int solved;
  int primal_infeasible;
  int dual_infeasible;
  int dual_variable_start; /* constraint_manager.h:30,80: next multiplier id (>= num_vars) */
  unsigned long lanczos_calls; /* index of the next PrepareStep / eigenvalue query (start vectors) */
};

/* ------------------------------------------------------------------ helpers */
static void mm(int m, int n, int k, const double* A, const double* B, double* C) {
  for (int j = 0; j < n; j++) {
    double* c = C + (size_t)j * m;
    for (int i = 0; i < m; i++) c[i] = 0;
    for (int p = 0; p < k; p++) {
      double b = B[(size_t)j * k + p];
      const double* a = A + (size_t)p * m;
      for (int i = 0; i < m; i++) c[i] += a[i] * b;
    }
  }
}
static double dotn(size_t n, const double* a, const double* b) {
  double s = 0;
  for (size_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
static double trace_n(int n, const double* a) {
  double s = 0;
  for (int i = 0; i < n; i++) s += a[(size_t)i * n + i];
  return s;
}

void cxo_default_config(cxo_config* c) { /* cone_program.h:17-38 */
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
  c->infeasibility_threshold = 1e5;
  c->kkt_error_tolerance = 1e10;
  c->kkt_solver = 0;
  c->enable_rescaling = 1;
  c->iterative_refinement_iterations = 0;
}

/* ------------------------------------------------------------ construction */
cxo_program* cxo_program_new(int num_vars) {
  cxo_program* p = (cxo_program*)calloc(1, sizeof(cxo_program));
  p->num_vars = num_vars;
  p->dual_variable_start = num_vars;
  p->b_scaling = 1;
  p->c_scaling = 1;
  return p;
}

static void constraint_free(cxo_constraint* c) {
  free(c->A);
  free(c->C);
  free(c->W);
  free(c->temp1);
  free(c->temp2);
  free(c->wa);
  free(c->Q);
  free(c->Agram);
  free(c->G_own);
  free(c->AW);
  free(c->AQc);
}

void cxo_program_free(cxo_program* p) {
  if (!p) return;
  for (int i = 0; i < p->K; i++) constraint_free(&p->c[i]);
  free(p->c);
  ivs_free(p->cliques, p->cap);
  ivs_free(p->dual_vars, p->cap);
  cxo_matrix_data_free(p->md);
  cxo_workspace_free(p->ws);
  free(p->sysAW);
  free(p->sysAQc);
  free(p->b_permuted);
  free(p->kkt_matrix);
  free(p->sqrt_inv_mu);
  free(p);
}

/* IsUnique constraint_manager.h:11-24 */
static int is_unique(int N, int m, const int* x) {
  int* cnt = (int*)calloc((size_t)(N > 0 ? N : 1), sizeof(int));
  int ok = 1;
  for (int i = 0; i < m; i++) {
    if (x[i] >= N || x[i] < 0) {
      ok = 0;
      break;
    }
    if (++cnt[x[i]] > 1) {
      ok = 0;
      break;
    }
  }
  free(cnt);
  return ok;
}

static cxo_constraint* new_constraint(cxo_program* p, int m, const int* vars) {
  if (vars && !is_unique(p->num_vars, m, vars)) return NULL;
  if (p->K == p->cap) {
    int ncap = p->cap ? 2 * p->cap : 16;
    p->c = (cxo_constraint*)realloc(p->c, sizeof(cxo_constraint) * (size_t)ncap);
    ivec* nc = ivs_new(ncap);
    ivec* nd = ivs_new(ncap);
    for (int i = 0; i < p->cap; i++) {
      nc[i] = p->cliques[i];
      nd[i] = p->dual_vars[i];
    }
    free(p->cliques);
    free(p->dual_vars);
    p->cliques = nc;
    p->dual_vars = nd;
    p->cap = ncap;
  }
  cxo_constraint* c = &p->c[p->K];
  memset(c, 0, sizeof(*c));
  c->m = m;
  for (int i = 0; i < m; i++) iv_push(&p->cliques[p->K], vars ? vars[i] : i);
  c->G_own = (double*)calloc((size_t)m * m, sizeof(double));
  c->G = c->G_own;
  c->AW = (double*)calloc((size_t)m, sizeof(double));
  c->AQc = (double*)calloc((size_t)m, sizeof(double));
  p->K++;
  p->initialized = 0;
  return c;
}

static double* dupd(const double* src, size_t n) {
  double* d = (double*)malloc(sizeof(double) * (n ? n : 1));
  memcpy(d, src, sizeof(double) * n);
  return d;
}

int cxo_add_lmi(cxo_program* p, int n, int m, const double* A, const double* C, const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  c->type = CXO_LMI;
  c->n = n;
  c->A = dupd(A, (size_t)m * n * n);
  c->C = dupd(C, (size_t)n * n);
  c->W = (double*)calloc((size_t)n * n, sizeof(double));
  c->temp1 = (double*)calloc((size_t)n * n, sizeof(double));
  c->temp2 = (double*)calloc((size_t)n * n, sizeof(double));
  return p->K - 1;
}

/* HermitianPsdConstraint<T>(n, a, c) hermitian_psd.h:46-51; A: m x d planes of n x n, C: d planes */
int cxo_add_hermitian(cxo_program* p, int n, int d, int m, const double* A, const double* C,
                      const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  if (d != 1 && d != 2 && d != 4 && d != 8) return -1;
  if (d == 8 && n > 3) return -1; /* interfaces/conex.cc:310-311: "Order of octonion algebra cannot be greater than 3." */
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  size_t sz = (size_t)n * n * d;
  c->type = CXO_HERMITIAN;
  c->n = n;
  c->hd = d;
  c->A = dupd(A, (size_t)m * sz);
  c->C = dupd(C, sz);
  c->W = (double*)calloc(sz, sizeof(double));
  c->temp1 = (double*)calloc(sz, sizeof(double)); /* WS */
  c->temp2 = (double*)calloc(sz, sizeof(double)); /* minus_s */
  return p->K - 1;
}

/* Program::AddConstraint(EqualityConstraints{A, b}, vars) -> ConstraintManager::
 * AddEqualityConstraint (constraint_manager.h:66-90): the clique is vars followed by r fresh
 * multiplier ids; A is r x m (column-major), the constraint is A y[vars] = b. */
int cxo_add_equality(cxo_program* p, int r, int m, const double* A, const double* b,
                     const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  int k = p->K - 1, mt = m + r;
  for (int i = 0; i < r; i++) {
    iv_push(&p->cliques[k], p->dual_variable_start + i);
    iv_push(&p->dual_vars[k], p->dual_variable_start + i);
  }
  p->dual_variable_start += r;
  free(c->G_own);
  free(c->AW);
  free(c->AQc);
  c->G_own = (double*)calloc((size_t)mt * mt, sizeof(double));
  c->G = c->G_own;
  c->AW = (double*)calloc((size_t)mt, sizeof(double));
  c->AQc = (double*)calloc((size_t)mt, sizeof(double));
  c->type = CXO_EQUALITY;
  c->n = r;       /* number of equality rows = multipliers */
  c->m = mt;      /* clique size seen by the KKT system */
  c->A = dupd(A, (size_t)r * m);
  c->C = dupd(b, (size_t)r);
  c->W = (double*)calloc((size_t)(r > 0 ? r : 1), sizeof(double)); /* lambda_ */
  return k;
}

int cxo_add_linear(cxo_program* p, int r, int m, const double* A, const double* cc,
                   const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  c->type = CXO_LINEAR;
  c->n = r;
  c->A = dupd(A, (size_t)r * m);
  c->C = dupd(cc, (size_t)r);
  c->W = (double*)calloc((size_t)r, sizeof(double));
  c->temp1 = (double*)calloc((size_t)r, sizeof(double));
  c->temp2 = (double*)calloc((size_t)r, sizeof(double));
  c->wa = (double*)calloc((size_t)r * m, sizeof(double));
  return p->K - 1;
}

int cxo_add_soc(cxo_program* p, int n, int m, const double* A, const double* cc, const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  c->type = CXO_SOC;
  c->n = n;
  c->A = dupd(A, (size_t)(n + 1) * m);
  c->C = dupd(cc, (size_t)(n + 1));
  c->W = (double*)calloc((size_t)(n + 1), sizeof(double));
  c->temp1 = (double*)calloc((size_t)(n + 1), sizeof(double));
  c->temp2 = (double*)calloc((size_t)(n + 1), sizeof(double));
  c->wa = (double*)calloc((size_t)(n + 1) * (m + 1), sizeof(double));
  return p->K - 1;
}

/* QuadraticConstraintBase ctor + Initialize, quadratic_cone_constraint.h:15-30, .cc:216-219 */
static void quad_apply_Q(const cxo_constraint* o, const double* x, double* out);
int cxo_add_quadratic(cxo_program* p, int n, int m, const double* Q, const double* A, const double* cc,
                      const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  const int len = n + 1;
  c->type = CXO_QUADRATIC;
  c->n = n;
  c->A = dupd(A, (size_t)len * m);
  c->C = dupd(cc, (size_t)len);
  c->Q = Q ? dupd(Q, (size_t)n * n) : NULL;
  c->W = (double*)calloc((size_t)len, sizeof(double));
  c->temp1 = (double*)calloc((size_t)len, sizeof(double));  /* d_q1 (temp2_1 of the reference) */
  c->temp2 = (double*)calloc((size_t)len, sizeof(double));  /* wsqrt_q1 (temp3_1) */
  c->wa = (double*)calloc((size_t)len * 4 + (size_t)m * 2, sizeof(double));
  c->Agram = (double*)calloc((size_t)m * m, sizeof(double));
  double* qa = (double*)malloc(sizeof(double) * (size_t)n);
  for (int j = 0; j < m; j++) {  /* A_gram = A1' (Q A1) */
    quad_apply_Q(c, c->A + (size_t)j * len + 1, qa);
    for (int i = 0; i < m; i++) c->Agram[(size_t)j * m + i] = dotn((size_t)n, c->A + (size_t)i * len + 1, qa);
  }
  free(qa);
  return p->K - 1;
}

int cxo_add_static(cxo_program* p, int m, const double* G, const int* vars) {
  if (!vars && m != p->num_vars) return -1;
  cxo_constraint* c = new_constraint(p, m, vars);
  if (!c) return -1;
  c->type = CXO_STATIC;
  c->n = 0;
  c->A = dupd(G, (size_t)m * m);
  return p->K - 1;
}

int cxo_num_constraints(const cxo_program* p) { return p->K; }

/* ----------------------------------------------------------------- identity */
static void set_identity_one(cxo_constraint* c) {
  switch (c->type) {
    case CXO_LMI: /* psd_constraint.cc:92-95 */
      memset(c->W, 0, sizeof(double) * (size_t)c->n * c->n);
      for (int i = 0; i < c->n; i++) c->W[(size_t)i * c->n + i] = 1;
      break;
    case CXO_HERMITIAN: /* hermitian_psd.h:54-56: T::Identity(rank) */
      memset(c->W, 0, sizeof(double) * (size_t)c->n * c->n * c->hd);
      for (int i = 0; i < c->n; i++) c->W[(size_t)i * c->n + i] = 1;
      break;
    case CXO_LINEAR: /* linear_constraint.cc:105 */
      for (int i = 0; i < c->n; i++) c->W[i] = 1;
      break;
    case CXO_SOC: /* soc_constraint.h:25-28 */
      memset(c->W, 0, sizeof(double) * (size_t)(c->n + 1));
      c->W[0] = 1;
      break;
    case CXO_QUADRATIC: /* quadratic_cone_constraint.cc:292-295 */
      memset(c->W, 0, sizeof(double) * (size_t)(c->n + 1));
      c->W[0] = 1;
      break;
    default:
      break;
  }
}
void cxo_set_identity(cxo_program* p) {
  for (int i = 0; i < p->K; i++) set_identity_one(&p->c[i]);
}

/* --------------------------------------------------------------- Initialize */
int cxo_initialize(cxo_program* p) {
  if (p->K == 0) return 0;
  cxo_matrix_data_free(p->md);
  cxo_workspace_free(p->ws);
  p->md = cxo_matrix_data_build(p->K, p->cliques, p->dual_vars);
  p->ws = cxo_workspace_new(p->K, p->md->cliques, p->md->supernode_size);
  int N = p->md->N;
  free(p->sysAW);
  free(p->sysAQc);
  free(p->b_permuted);
  p->sysAW = (double*)calloc((size_t)N, sizeof(double));
  p->sysAQc = (double*)calloc((size_t)N, sizeof(double));
  p->b_permuted = (double*)calloc((size_t)N, sizeof(double));
  /* DoBind + BindDiagonalBlock: direct_update detection supernodal_assembler.cc:72-91 */
  for (int e = p->K - 1; e >= 0; e--) {
    int i = p->md->clique_order[e];
    cxo_constraint* c = &p->c[i];
    c->epos = e;
    c->direct_update = 0;
    c->G = c->G_own;
    const ivec* sn = &p->md->supernodes_pos[e];
    if (sn->n > 0 && c->m == sn->n) {
      int direct = 1;
      if (g_strict_direct_update && sn->d[0] != 0) direct = 0;
      for (int q = 1; q < sn->n; q++)
        if (sn->d[q] <= sn->d[q - 1]) {
          direct = 0;
          break;
        }
      if (direct) {
        c->direct_update = 1;
        c->G = p->ws->slab + p->ws->diag_off[e];
      }
    }
  }
  cxo_set_identity(p);
  p->b_scaling = 1;
  p->c_scaling = 1;
  p->initialized = 1;
  return 1;
}

/* -------------------------------------------------------------- getters */
int cxo_system_size(const cxo_program* p) { return p->md ? p->md->N : 0; }
int cxo_get_order(const cxo_program* p, int* order) {
  memcpy(order, p->md->clique_order, sizeof(int) * (size_t)p->K);
  return p->K;
}
int cxo_get_permutation(const cxo_program* p, int* perm, int* perm_inv) {
  memcpy(perm, p->md->permutation, sizeof(int) * (size_t)p->md->num_vars);
  memcpy(perm_inv, p->md->permutation_inverse, sizeof(int) * (size_t)p->md->num_vars);
  return p->md->num_vars;
}
int cxo_get_list(const cxo_program* p, int which, int e, int* out) {
  const ivec* v = NULL;
  switch (which) {
    case 0: v = &p->md->cliques[e]; break;
    case 1: v = &p->md->supernodes_orig[e]; break;
    case 2: v = &p->md->separators_orig[e]; break;
    case 3: v = &p->md->supernodes_pos[e]; break;
    case 4: v = &p->md->separators_pos[e]; break;
    case 5: v = &p->md->pc_supernodes[e]; break;
    case 6: v = &p->md->pc_separators[e]; break;
    default: return -1;
  }
  if (out) memcpy(out, v->d, sizeof(int) * (size_t)v->n);
  return v->n;
}
int cxo_get_supernode_sizes(const cxo_program* p, int* out) {
  memcpy(out, p->md->supernode_size, sizeof(int) * (size_t)p->K);
  return p->K;
}
long cxo_slab_size(const cxo_program* p) { return p->ws->slab_size; }
int cxo_get_block_offsets(const cxo_program* p, long* diag_off, long* offd_off) {
  memcpy(diag_off, p->ws->diag_off, sizeof(long) * (size_t)p->K);
  memcpy(offd_off, p->ws->offd_off, sizeof(long) * (size_t)p->K);
  return p->K;
}
int cxo_get_ss_index(const cxo_program* p, int e, long* out) {
  if (out) memcpy(out, p->ws->ss_index[e], sizeof(long) * (size_t)p->ws->ss_count[e]);
  return p->ws->ss_count[e];
}
int cxo_dual_size(const cxo_program* p, int i) {
  const cxo_constraint* c = &p->c[i];
  switch (c->type) {
    case CXO_LMI: return c->n * c->n;
    case CXO_HERMITIAN: return c->n * c->n * c->hd;
    case CXO_EQUALITY: return c->n;
    case CXO_LINEAR: return c->n;
    case CXO_SOC: return c->n + 1;
    case CXO_QUADRATIC: return c->n + 1;
    default: return 0;
  }
}
void cxo_get_W(const cxo_program* p, int i, double* out) {
  memcpy(out, p->c[i].W, sizeof(double) * (size_t)cxo_dual_size(p, i));
}
void cxo_set_W(cxo_program* p, int i, const double* in) {
  memcpy(p->c[i].W, in, sizeof(double) * (size_t)cxo_dual_size(p, i));
}

/* ------------------------------------------------------ Schur complements */
/* dense_lmi_constraint.cc:72-103 (initialize == true) */
static void schur_lmi(cxo_constraint* o) {
  int n = o->n, m = o->m;
  size_t nn = (size_t)n * n;
  double* AW = o->temp1;
  double* WAW = o->temp2;
  for (int i = 0; i < m; i++) {
    const double* Ai = o->A + (size_t)i * nn;
    mm(n, n, n, Ai, o->W, AW);   /* ComputeAW :29-33 */
    mm(n, n, n, o->W, AW, WAW);
    for (int j = 0; j <= i; j++) /* G.row(i).head(i+1) = vec(WAW)^T Avect[:,0:i] */
      o->G[(size_t)j * m + i] = dotn(nn, WAW, o->A + (size_t)j * nn);
    o->AW[i] = trace_n(n, AW);
    o->AQc[i] = dotn(nn, o->C, WAW); /* EvalDualObjective(WAW) :68-70 */
  }
  o->ip_wc = dotn(nn, o->C, o->W);
  mm(n, n, n, o->C, o->W, AW); /* ComputeWCW :35-39 */
  mm(n, n, n, o->W, AW, WAW);
  o->ip_cQc = dotn(nn, o->C, WAW);
}

/* linear_constraint.cc:177-205 */
static void schur_linear(cxo_constraint* o) {
  int r = o->n, m = o->m;
  double* WA = o->wa;
  double* WC = o->temp1;
  for (int j = 0; j < m; j++)
    for (int i = 0; i < r; i++) WA[(size_t)j * r + i] = o->W[i] * o->A[(size_t)j * r + i];
  for (int i = 0; i < r; i++) WC[i] = o->W[i] * o->C[i];
  double s = 0, s2 = 0;
  for (int i = 0; i < r; i++) {
    s += WC[i];
    s2 += WC[i] * WC[i];
  }
  o->ip_wc = s;
  o->ip_cQc = s2;
  for (int j = 0; j < m; j++)
    for (int i = 0; i < m; i++)
      o->G[(size_t)j * m + i] = dotn((size_t)r, WA + (size_t)i * r, WA + (size_t)j * r);
  for (int i = 0; i < m; i++) {
    o->AW[i] = dotn((size_t)r, o->A + (size_t)i * r, o->W);
    o->AQc[i] = dotn((size_t)r, WA + (size_t)i * r, WC);
  }
}

/* --- spin factor algebra, soc_constraint.cc:14-163 --- */
/* QuadraticRepresentation(x, y) :130-143 ; vectors of length len */
static void soc_quadrep(int len, const double* x, const double* y, double* out) {
  double tail2 = 0;
  for (int i = 1; i < len; i++) tail2 += x[i] * x[i];
  double det_x = x[0] * x[0] - tail2;
  double xy = dotn((size_t)len, x, y);
  for (int i = 0; i < len; i++) out[i] = (2 * xy) * x[i] + (i == 0 ? -det_x * y[i] : det_x * y[i]);
}
/* spectral map f applied through SpectralDecompSpinFactor :14-69 + Idempotents :57-69 */
static void soc_spectral(int n, double x0, const double* x1, int op /*0 sqrt,1 exp*/, double* z) {
  double nq = sqrt(dotn((size_t)n, x1, x1));
  double e0 = x0 + nq, e1 = x0 - nq;
  double f0 = op == 0 ? sqrt(e0) : exp(e0);
  double f1 = op == 0 ? sqrt(e1) : exp(e1);
  if (nq > 0) {
    z[0] = f0 * .5 + f1 * .5;
    for (int i = 0; i < n; i++) {
      double q = x1[i] / nq;
      z[1 + i] = f0 * (.5 * q) + f1 * (-.5 * q);
    }
  } else {
    z[0] = f0 * .5 + f1 * .5;
    for (int i = 0; i < n; i++) z[1 + i] = 0;
  }
}

/* soc_constraint.cc:272-303 */
static void schur_soc(cxo_constraint* o) {
  int n = o->n, m = o->m, len = n + 1;
  double* Wsqrt = o->temp2;
  soc_spectral(n, o->W[0], o->W + 1, 0, Wsqrt);
  double* WA = o->wa;                     /* len x m */
  double* WsqrtC = o->wa + (size_t)len * m; /* len */
  soc_quadrep(len, Wsqrt, o->C, WsqrtC);
  for (int i = 0; i < m; i++) soc_quadrep(len, Wsqrt, o->A + (size_t)i * len, WA + (size_t)i * len);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < m; i++)
      o->G[(size_t)j * m + i] = 2 * dotn((size_t)len, WA + (size_t)i * len, WA + (size_t)j * len);
  for (int i = 0; i < m; i++) {
    o->AW[i] = 2 * dotn((size_t)len, o->A + (size_t)i * len, o->W);
    o->AQc[i] = 2 * dotn((size_t)len, WA + (size_t)i * len, WsqrtC);
  }
  o->ip_wc = 2 * WsqrtC[0];
  o->ip_cQc = 2 * dotn((size_t)len, WsqrtC, WsqrtC);
}

/* --- Lorentz cone with inner-product matrix Q on the vector part, quadratic_cone_constraint.cc --- */
static void quad_apply_Q(const cxo_constraint* o, const double* x, double* out) { /* Q x, or x when Q = I */
  const int n = o->n;
  if (!o->Q) {
    memcpy(out, x, sizeof(double) * (size_t)n);
    return;
  }
  for (int i = 0; i < n; i++) out[i] = 0;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) out[i] += o->Q[(size_t)j * n + i] * x[j];
}
static double quad_ip(const cxo_constraint* o, const double* x, const double* y) { /* InnerProduct :35-44, SquaredNorm :14-22 */
  double* qy = (double*)malloc(sizeof(double) * (size_t)o->n);
  quad_apply_Q(o, y, qy);
  const double r = dotn((size_t)o->n, x, qy);
  free(qy);
  return r;
}
static double quad_norm(const cxo_constraint* o, const double* x) { return sqrt(fabs(quad_ip(o, x, x))); } /* :24-33 */
/* QuadraticRepresentation :46-61 */
static void quad_quadrep(int n, double x1_norm_sq, double ip_x1_y1, double x0, const double* x1, double y0,
                         const double* y1, double* z0, double* z1) {
  const double det_x = x0 * x0 - x1_norm_sq;
  const double scale = 2 * (x0 * y0 + ip_x1_y1);
  *z0 = scale * x0 - det_x * y0;
  for (int i = 0; i < n; i++) z1[i] = scale * x1[i] + det_x * y1[i];
}
static void quad_exp(int n, double k, double* x0, double* x1) { /* :63-70 */
  if (k > 0) {
    const double f = .5 * (exp(*x0 + k) - exp(*x0 - k)) / k;
    for (int i = 0; i < n; i++) x1[i] *= f;
  }
  *x0 = .5 * (exp(*x0 + k) + exp(*x0 - k));
}
static void quad_sqrt(int n, double k, double* x0, double* x1) { /* :72-79, square_root = sqrt(fabs(.)) */
  if (k > 0) {
    const double f = .5 * (sqrt(fabs(*x0 + k)) - sqrt(fabs(*x0 - k))) / k;
    for (int i = 0; i < n; i++) x1[i] *= f;
  }
  *x0 = .5 * (sqrt(fabs(*x0 + k)) + sqrt(fabs(*x0 - k)));
}
static void quad_negative_slack(const cxo_constraint* o, double k, const double* y, double* s0, double* s1) { /* :131-139 */
  const int len = o->n + 1;
  double a0 = 0;
  for (int j = 0; j < o->m; j++) a0 += o->A[(size_t)j * len] * y[j];
  *s0 = a0 - o->C[0] * k;
  for (int i = 0; i < o->n; i++) s1[i] = 0;
  for (int j = 0; j < o->m; j++)
    for (int i = 0; i < o->n; i++) s1[i] += o->A[(size_t)j * len + 1 + i] * y[j];
  for (int i = 0; i < o->n; i++) s1[i] -= o->C[1 + i] * k;
}
/* ConstructSchurComplementSystem(QuadraticConstraintBase*, initialize = true) :240-290, SchurComplement :88-100 */
static void schur_quadratic(cxo_constraint* o) {
  const int n = o->n, m = o->m, len = n + 1;
  const double W0 = o->W[0];
  const double* W1 = o->W + 1;
  const double C0 = o->C[0];
  const double* C1 = o->C + 1;
  double* QW1 = o->wa;             /* n */
  double* QC1 = o->wa + len;       /* n */
  double* A_dot_x = o->wa + 2 * len;       /* m */
  double* v = A_dot_x + m;                 /* m : A_dot_x + A0 W0 */
  quad_apply_Q(o, W1, QW1);
  quad_apply_Q(o, C1, QC1);
  const double c_dot_x = dotn((size_t)n, C1, QW1);
  for (int i = 0; i < m; i++) A_dot_x[i] = dotn((size_t)n, o->A + (size_t)i * len + 1, QW1);
  const double det_w = W0 * W0 - dotn((size_t)n, W1, QW1);
  for (int i = 0; i < m; i++) v[i] = A_dot_x[i] + o->A[(size_t)i * len] * W0;
  for (int j = 0; j < m; j++)
    for (int i = 0; i < m; i++) {
      const double a0a0 = o->A[(size_t)i * len] * o->A[(size_t)j * len];
      double g = (a0a0 - o->Agram[(size_t)j * m + i]) * -det_w;
      g += v[i] * v[j];
      g += v[i] * v[j];
      o->G[(size_t)j * m + i] = g;
    }
  for (int i = 0; i < m; i++) {
    o->AW[i] = v[i];
    o->AQc[i] = det_w * (dotn((size_t)n, o->A + (size_t)i * len + 1, QC1) - o->A[(size_t)i * len] * C0);
  }
  o->ip_cQc = det_w * (dotn((size_t)n, C1, QC1) - C0 * C0);
  const double scale = dotn((size_t)n, QW1, C1) + C0 * W0; /* (Q = I: W1 . C1 + C0 W0, the same value) */
  for (int i = 0; i < m; i++) o->AQc[i] += 2 * v[i] * scale;
  o->ip_cQc += 2 * (c_dot_x + C0 * W0) * scale;
  o->ip_wc = scale;
  /* "Account for Jordan inner-product <x, y> := 2 x^T y" */
  for (int i = 0; i < m; i++) {
    o->AQc[i] *= 2;
    o->AW[i] *= 2;
  }
  o->ip_cQc *= 2;
  o->ip_wc *= 2;
  for (size_t q = 0; q < (size_t)m * m; q++) o->G[q] *= 2;
}
/* PrepareStep(QuadraticConstraintBase*) :176-214.  wsqrt_q0 is a REFERENCE to *W0: the scalar part of
 * w^{1/2} overwrites W0 here (W1 stays), TakeStep reads it from there. */
static void quad_prepare_step(cxo_constraint* o, double c_weight, const double* y, double* normsqrd, double* norminfd) {
  const int n = o->n;
  double* ms1 = (double*)malloc(sizeof(double) * (size_t)n);
  double ms0;
  quad_negative_slack(o, c_weight, y, &ms0, ms1);
  double* wsq1 = o->temp2;
  memcpy(wsq1, o->W + 1, sizeof(double) * (size_t)n);
  quad_sqrt(n, quad_norm(o, wsq1), &o->W[0], wsq1);
  o->wsq_norm_sqr = quad_ip(o, wsq1, wsq1);
  double* d1 = o->temp1;
  quad_quadrep(n, o->wsq_norm_sqr, quad_ip(o, wsq1, ms1), o->W[0], wsq1, ms0, ms1, &o->d0, d1);
  o->d0 += 1;
  const double nd = quad_norm(o, d1);
  const double e0 = o->d0 + nd, e1 = o->d0 - nd;
  *norminfd = fabs(e0);
  if (*norminfd < fabs(e1)) *norminfd = fabs(e1);
  *normsqrd = e0 * e0 + e1 * e1;
  free(ms1);
}
/* TakeStep(QuadraticConstraintBase*) :221-243 */
static void quad_take_step(cxo_constraint* o, double step_size) {
  const int n = o->n;
  double* d1 = o->temp1;
  if (step_size != 1) {
    o->d0 = step_size * o->d0;
    for (int i = 0; i < n; i++) d1[i] = step_size * d1[i];
  }
  quad_exp(n, quad_norm(o, d1), &o->d0, d1);
  const double* wsq1 = o->temp2;
  const double ip = quad_ip(o, wsq1, d1);
  double w0;
  double* w1 = (double*)malloc(sizeof(double) * (size_t)n);
  quad_quadrep(n, o->wsq_norm_sqr, ip, o->W[0], wsq1, o->d0, d1, &w0, w1);
  o->W[0] = w0;
  memcpy(o->W + 1, w1, sizeof(double) * (size_t)n);
  free(w1);
}
/* GetWeightedSlackEigenvalues(QuadraticConstraintBase*) :142-174 */
static void quad_weighted_eigs(cxo_constraint* o, const double* y, double c_weight, double* lmin, double* lmax,
                               double* frob, double* tr) {
  const int n = o->n;
  double* ms1 = (double*)malloc(sizeof(double) * (size_t)n);
  double* wsq1 = (double*)malloc(sizeof(double) * (size_t)n);
  double* Ws1 = (double*)malloc(sizeof(double) * (size_t)n);
  double ms0, wsq0 = o->W[0], Ws0;
  quad_negative_slack(o, c_weight, y, &ms0, ms1);
  memcpy(wsq1, o->W + 1, sizeof(double) * (size_t)n);
  quad_sqrt(n, quad_norm(o, wsq1), &wsq0, wsq1);
  quad_quadrep(n, quad_ip(o, wsq1, wsq1), quad_ip(o, wsq1, ms1), wsq0, wsq1, ms0, ms1, &Ws0, Ws1);
  const double nq = quad_norm(o, Ws1);
  const double e0 = Ws0 + nq, e1 = Ws0 - nq;
  const double mn = e0 < e1 ? e0 : e1, mx = e0 < e1 ? e1 : e0;
  *lmax = -mn;
  *lmin = -mx;
  *frob = (*lmax) * (*lmax) + (*lmin) * (*lmin);
  *tr = (*lmax) + (*lmin);
  free(ms1);
  free(wsq1);
  free(Ws1);
}

static void schur_static(cxo_constraint* o) { /* supernodal_assembler.h:127 G = A_ */
  memcpy(o->G, o->A, sizeof(double) * (size_t)o->m * o->m);
  memset(o->AW, 0, sizeof(double) * (size_t)o->m);
  memset(o->AQc, 0, sizeof(double) * (size_t)o->m);
  o->ip_wc = 0;
  o->ip_cQc = 0;
}

static void schur_hermitian(cxo_constraint* o);
static void schur_equality(cxo_constraint* o);
static void set_dense_data(cxo_constraint* o) {
  switch (o->type) {
    case CXO_LMI: schur_lmi(o); break;
    case CXO_HERMITIAN: schur_hermitian(o); break;
    case CXO_EQUALITY: schur_equality(o); break;
    case CXO_LINEAR: schur_linear(o); break;
    case CXO_SOC: schur_soc(o); break;
    case CXO_QUADRATIC: schur_quadratic(o); break;
    case CXO_STATIC: schur_static(o); break;
  }
}

/* GetCoeff supernodal_assembler.cc:59-70 */
static double get_coeff(const cxo_constraint* o, int i, int j) {
  if (i < 0 || j < 0) return 0;
  return (i >= j) ? o->G[(size_t)j * o->m + i] : o->G[(size_t)i * o->m + j];
}

/* UpdateBlocks supernodal_assembler.cc:113-165 */
static void update_blocks(cxo_program* p, cxo_constraint* o) {
  set_dense_data(o);
  int e = o->epos;
  cxo_workspace* w = p->ws;
  const ivec* r = &p->md->supernodes_pos[e];
  const ivec* s = &p->md->separators_pos[e];
  int ns = r->n, nsep = s->n;
  double* D = w->slab + w->diag_off[e];
  double* B = w->slab + w->offd_off[e];
  if (o->direct_update) {
    /* G aliases the diagonal block, which descendants scatter into later in this pass; keep a
     * pristine copy of the constraint's own Schur block for inspection (test hook only). */
    memcpy(o->G_own, o->G, sizeof(double) * (size_t)o->m * o->m);
    if (ns > 0 && nsep > 0) memset(B, 0, sizeof(double) * (size_t)ns * nsep);
    return;
  }
  if (ns > 0) { /* SetLowerTri */
    for (int j = 0; j < ns; j++)
      for (int i = j; i < ns; i++) D[(size_t)j * ns + i] = get_coeff(o, r->d[i], r->d[j]);
  }
  if (ns > 0 && nsep > 0) { /* Set */
    for (int j = 0; j < nsep; j++)
      for (int i = 0; i < ns; i++) B[(size_t)j * ns + i] = get_coeff(o, r->d[i], s->d[j]);
  }
  if (w->ss_count[e] > 0) { /* Scatter */
    int cnt = 0;
    for (int j = 0; j < nsep; j++)
      for (int i = j; i < nsep; i++) w->slab[w->ss_index[e][cnt++]] += get_coeff(o, s->d[i], s->d[j]);
  }
}

/* kkt_solver.cc:164-170 + constraint_manager.h:107-124 */
void cxo_assemble(cxo_program* p) {
  for (int e = p->K - 1; e >= 0; e--) update_blocks(p, &p->c[p->md->clique_order[e]]);
  int N = p->md->N;
  memset(p->sysAW, 0, sizeof(double) * (size_t)N);
  memset(p->sysAQc, 0, sizeof(double) * (size_t)N);
  p->sys_wc = 0;
  p->sys_cQc = 0;
  for (int i = 0; i < p->K; i++) {
    cxo_constraint* c = &p->c[i];
    p->sys_wc += c->ip_wc;
    p->sys_cQc += c->ip_cQc;
    for (int q = 0; q < p->cliques[i].n; q++) {
      int k = p->cliques[i].d[q];
      p->sysAW[k] += c->AW[q];
      p->sysAQc[k] += c->AQc[q];
    }
  }
}

void cxo_get_slab(const cxo_program* p, double* out) {
  memcpy(out, p->ws->slab, sizeof(double) * (size_t)p->ws->slab_size);
}
void cxo_get_constraint_schur(const cxo_program* p, int i, double* G, double* AW, double* AQc,
                              double* scalars) {
  const cxo_constraint* c = &p->c[i];
  if (G) memcpy(G, c->G_own, sizeof(double) * (size_t)c->m * c->m);
  if (AW) memcpy(AW, c->AW, sizeof(double) * (size_t)c->m);
  if (AQc) memcpy(AQc, c->AQc, sizeof(double) * (size_t)c->m);
  if (scalars) {
    scalars[0] = c->ip_wc;
    scalars[1] = c->ip_cQc;
  }
}
void cxo_get_residuals(const cxo_program* p, double* AW, double* AQc, double* scalars) {
  int N = p->md->N;
  if (AW) memcpy(AW, p->sysAW, sizeof(double) * (size_t)N);
  if (AQc) memcpy(AQc, p->sysAQc, sizeof(double) * (size_t)N);
  if (scalars) {
    scalars[0] = p->sys_wc;
    scalars[1] = p->sys_cQc;
  }
}

/* Factor kkt_solver.cc:172-199 (LLT branch) */
/* SupernodalKKTSolver::Factor kkt_solver.cc:172-199: Cholesky unless some constraint carries
 * multipliers, then block LDLT (which always "succeeds"; regularisation is only recorded) */
void cxo_kkt_matrix(const cxo_program* p, double* out);

/* SetIterativeRefinementIterations (kkt_solver.h): takes effect at the next Factor */
void cxo_set_refinement(cxo_program* p, int iterations) { p->refinement_iterations = iterations; }

int cxo_factor(cxo_program* p) {
  if (p->refinement_iterations > 0) { /* kkt_solver.cc:177-179: dense copy before factoring */
    size_t N = (size_t)p->md->N;
    free(p->kkt_matrix);
    p->kkt_matrix = (double*)malloc(sizeof(double) * N * N);
    cxo_kkt_matrix(p, p->kkt_matrix);
  }
  int use_cholesky = 1;
  for (int i = 0; i < p->K; i++)
    if (p->dual_vars[i].n > 0) {
      use_cholesky = 0;
      break;
    }
  if (use_cholesky) {
    p->ws->factored_ldlt = 0;
    return cxo_block_cholesky(p->ws);
  }
  cxo_block_ldlt(p->ws);
  return 1;
}

/* factorization_regularized_ kkt_solver.cc:192 */
int cxo_factor_regularized(const cxo_program* p) { return p->ws ? p->ws->regularized : 0; }

/* SolveInPlace kkt_solver.cc:220-263 (no refinement): b_perm = Pt^T b ; solves ; b = Pt b_perm.
 * Pt.indices() = permutation_inverse, so (Pt^T b)(i) = b(permutation_inverse[i]). */
static void solve_once(cxo_program* p, double* y) {
  int N = p->md->N;
  const int* pinv = p->md->permutation_inverse;
  for (int i = 0; i < N; i++) p->b_permuted[i] = y[pinv[i]];
  if (p->ws->factored_ldlt) {
    cxo_solve_ldlt(p->ws, p->b_permuted);
  } else {
    cxo_apply_block_inverse(p->ws, p->b_permuted);
    cxo_apply_block_inverse_of_transpose(p->ws, p->b_permuted);
  }
  for (int i = 0; i < N; i++) y[pinv[i]] = p->b_permuted[i];
}

/* With refinement (kkt_solver.cc:233-261): y <- y + K^-1 (b - kkt_matrix_ y), `iterations` times,
 * kkt_matrix_ being the dense copy Factor took of the assembled matrix. */
void cxo_solve_inplace(cxo_program* p, double* y) {
  int N = p->md->N;
  int iters = p->kkt_matrix ? p->refinement_iterations : 0;
  double* total = NULL;
  if (iters > 0) {
    total = (double*)malloc(sizeof(double) * (size_t)N);
    memcpy(total, y, sizeof(double) * (size_t)N);
  }
  solve_once(p, y);
  if (iters > 0) {
    double* res = (double*)malloc(sizeof(double) * (size_t)N);
    for (int it = 0; it < iters; it++) {
      for (int i = 0; i < N; i++) res[i] = 0.0;
      for (int j = 0; j < N; j++) { /* kkt_matrix_ * y, column sweep */
        const double* col = p->kkt_matrix + (size_t)j * N;
        double yj = y[j];
        for (int i = 0; i < N; i++) res[i] += col[i] * yj;
      }
      for (int i = 0; i < N; i++) res[i] = total[i] - res[i];
      solve_once(p, res);
      for (int i = 0; i < N; i++) y[i] += res[i];
    }
    free(res);
    free(total);
  }
}

/* KKTMatrix kkt_solver.cc:265-269 : Pt * selfadjoint(ToDense) * Pt^T */
void cxo_kkt_matrix(const cxo_program* p, double* out) {
  int N = p->md->N;
  double* L = (double*)malloc(sizeof(double) * (size_t)N * N);
  cxo_workspace_to_dense(p->ws, L);
  const int* pinv = p->md->permutation_inverse;
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) {
      double v = (i >= j) ? L[(size_t)j * N + i] : L[(size_t)i * N + j];
      out[(size_t)pinv[j] * N + pinv[i]] = v;
    }
  free(L);
}

/* ------------------------------------------------------------ cone updates */
/* DenseLMIConstraint::ComputeNegativeSlack dense_lmi_constraint.cc:8-27 */
static void lmi_negative_slack(const cxo_constraint* o, double k, const double* y, double* s) {
  size_t nn = (size_t)o->n * o->n;
  memset(s, 0, sizeof(double) * nn);
  for (int i = 0; i < o->m; i++) {
    const double* Ai = o->A + (size_t)i * nn;
    double yi = y[i];
    for (size_t q = 0; q < nn; q++) s[q] += yi * Ai[q];
  }
  for (size_t q = 0; q < nn; q++) s[q] -= k * o->C[q];
}

static int argmax_diag(int n, const double* a) { /* maxCoeff(&index): first max */
  int idx = 0;
  for (int i = 1; i < n; i++)
    if (a[(size_t)i * n + i] > a[(size_t)idx * n + idx]) idx = i;
  return idx;
}

/* PrepareStep(PsdConstraint*) psd_constraint.cc:45-84 */
static void lmi_prepare_step(cxo_constraint* o, int affine, double c_weight, double e_weight,
                             const double* y, double* normsqrd, double* norminfd) {
  int n = o->n;
  size_t nn = (size_t)n * n;
  double* minus_s = o->temp1; /* aliases WS */
  double* WS = o->temp1;
  double* WSWS = o->temp2;
  double* tmp = (double*)malloc(sizeof(double) * nn);
  lmi_negative_slack(o, c_weight, y, minus_s);
  mm(n, n, n, o->W, minus_s, tmp); /* WS = W * minus_s (Eigen evaluates into a temporary) */
  memcpy(WS, tmp, sizeof(double) * nn);
  if (affine) { /* AffineUpdate :33-43 */
    double* WSW = o->temp2;
    mm(n, n, n, WS, o->W, WSW);
    if (e_weight == 0) {
      for (size_t q = 0; q < nn; q++) o->W[q] += WSW[q];
    } else {
      for (size_t q = 0; q < nn; q++) o->W[q] = o->W[q] * (1 + e_weight) + WSW[q];
    }
    free(tmp);
    return;
  }
  int index = argmax_diag(n, WS);
  double* eigs = (double*)malloc(sizeof(double) * (size_t)(n > 1 ? n : 2));
  /* minus_s aliases WS here, so r = WS.col(index) */
  int ne = cxo_asymmetric_lanczos(n, WS, o->W, minus_s + (size_t)index * n, n / 2, eigs);
  double mn = eigs[0], mx = eigs[0];
  for (int i = 1; i < ne; i++) {
    if (eigs[i] < mn) mn = eigs[i];
    if (eigs[i] > mx) mx = eigs[i];
  }
  double l1 = fabs(e_weight + mn), l2 = fabs(e_weight + mx);
  double norminf = l1 < l2 ? l2 : l1;
  mm(n, n, n, WS, WS, WSWS);
  *normsqrd = trace_n(n, WSWS) + 2 * trace_n(n, WS) + n;
  *norminfd = norminf;
  free(eigs);
  free(tmp);
}

/* TakeStep(PsdConstraint*) -> GeodesicUpdate psd_constraint.cc:13-28, 86-90 */
static void lmi_take_step(cxo_constraint* o, double e_weight, double step_size) {
  int n = o->n;
  size_t nn = (size_t)n * n;
  double* WS = o->temp1;
  double* expWS = o->temp2;
  for (int i = 0; i < n; i++) WS[(size_t)i * n + i] += e_weight;
  if (step_size != 1.0)
    for (size_t q = 0; q < nn; q++) WS[q] *= step_size;
  cxo_pade_expm(n, WS, expWS);
  double* tmp = (double*)malloc(sizeof(double) * nn);
  mm(n, n, n, expWS, o->W, tmp);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) {
      WS[(size_t)j * n + i] = tmp[(size_t)i * n + j]; /* WS = W^T */
    }
  for (size_t q = 0; q < nn; q++) o->W[q] = (tmp[q] + WS[q]) * 0.5;
  free(tmp);
}

/* GetWeightedSlackEigenvalues(PsdConstraint*) psd_constraint.cc:97-128 */
static void lmi_weighted_eigs(cxo_constraint* o, const double* y, double c_weight, double* lmin,
                              double* lmax, double* frob, double* tr) {
  int n = o->n;
  double* minus_s = o->temp1;
  double* WS = o->temp2;
  lmi_negative_slack(o, c_weight, y, minus_s);
  mm(n, n, n, o->W, minus_s, WS);
  int index = argmax_diag(n, WS);
  double* eigs = (double*)malloc(sizeof(double) * (size_t)(n > 1 ? n : 2));
  int ne = cxo_asymmetric_lanczos(n, WS, o->W, minus_s + (size_t)index * n, n / 2, eigs);
  double mn = eigs[0], mx = eigs[0];
  for (int i = 1; i < ne; i++) {
    if (eigs[i] < mn) mn = eigs[i];
    if (eigs[i] > mx) mx = eigs[i];
  }
  *lmax = -mn;
  *lmin = -mx;
  double* WSWS = o->temp1;
  mm(n, n, n, WS, WS, WSWS);
  *frob = trace_n(n, WSWS);
  *tr = -trace_n(n, WS);
  free(eigs);
}

/* linear_constraint.cc:164-168 ; note topRows(number_of_variables) */
static void lin_negative_slack(const cxo_constraint* o, double k, const double* y, double* s) {
  int r = o->n;
  for (int i = 0; i < r; i++) s[i] = 0;
  for (int j = 0; j < o->m; j++)
    for (int i = 0; i < r; i++) s[i] += o->A[(size_t)j * r + i] * y[j];
  for (int i = 0; i < r; i++) s[i] -= o->C[i] * k;
}

static void lin_take_step(cxo_constraint* o, int affine, double step_size) { /* :130-145 */
  int r = o->n;
  if (!affine) {
    double* d = o->temp2;
    if (step_size != 1)
      for (int i = 0; i < r; i++) d[i] *= step_size;
    for (int i = 0; i < r; i++) d[i] = exp(d[i]);
    for (int i = 0; i < r; i++) o->W[i] *= d[i];
  } else { /* AffineUpdate :170-175 */
    double* ms = o->temp1;
    for (int i = 0; i < r; i++) {
      double sw = ms[i] * o->W[i];
      ms[i] = sw;
      o->W[i] += o->W[i] * sw;
    }
  }
}

static void lin_prepare_step(cxo_constraint* o, int affine, double c_weight, double e_weight,
                             double step_size, const double* y, double* normsqrd,
                             double* norminfd) { /* :108-128 */
  int r = o->n;
  if (!affine) {
    double* d = o->temp2;
    lin_negative_slack(o, c_weight, y, d);
    double mx = 0, s2 = 0;
    for (int i = 0; i < r; i++) {
      d[i] = d[i] * o->W[i] + e_weight;
      if (fabs(d[i]) > mx || i == 0) mx = fabs(d[i]) > mx ? fabs(d[i]) : mx;
      s2 += d[i] * d[i];
    }
    *norminfd = mx;
    *normsqrd = s2;
  } else {
    lin_negative_slack(o, 0, y, o->temp1);
    lin_take_step(o, 1, step_size);
  }
}

static void lin_weighted_eigs(cxo_constraint* o, const double* y, double c_weight, double* lmin,
                              double* lmax, double* frob, double* tr) { /* :147-162 */
  int r = o->n;
  double* ms = o->temp1;
  double* Ws = o->temp2;
  lin_negative_slack(o, c_weight, y, ms);
  double mn = 0, mx = 0, s2 = 0, s = 0;
  for (int i = 0; i < r; i++) {
    Ws[i] = o->W[i] * ms[i];
    if (i == 0 || Ws[i] < mn) mn = Ws[i];
    if (i == 0 || Ws[i] > mx) mx = Ws[i];
    s2 += Ws[i] * Ws[i];
    s += Ws[i];
  }
  *lmax = -mn;
  *lmin = -mx;
  *frob = s2;
  *tr = -s;
}

static void soc_negative_slack(const cxo_constraint* o, double k, const double* y, double* s) {
  int len = o->n + 1; /* soc_constraint.cc:193-197 */
  for (int i = 0; i < len; i++) s[i] = 0;
  for (int j = 0; j < o->m; j++)
    for (int i = 0; i < len; i++) s[i] += o->A[(size_t)j * len + i] * y[j];
  for (int i = 0; i < len; i++) s[i] -= o->C[i] * k;
}

/* PrepareStep(SOCConstraint*) soc_constraint.cc:251-270 : overwrites W with w^{1/2} */
static void soc_prepare_step(cxo_constraint* o, double c_weight, const double* y,
                             double* normsqrd, double* norminfd) {
  int n = o->n, len = n + 1;
  double* minus_s = (double*)malloc(sizeof(double) * (size_t)len);
  double* wsqrt = (double*)malloc(sizeof(double) * (size_t)len);
  double* d = (double*)malloc(sizeof(double) * (size_t)len);
  soc_negative_slack(o, c_weight, y, minus_s);
  soc_spectral(n, o->W[0], o->W + 1, 0, wsqrt);
  memcpy(o->W, wsqrt, sizeof(double) * (size_t)len);
  soc_quadrep(len, wsqrt, minus_s, d);
  d[0] += 1;
  memcpy(o->temp1, d + 1, sizeof(double) * (size_t)n);
  o->d0 = d[0];
  double nq = sqrt(dotn((size_t)n, d + 1, d + 1)); /* NormInf :182-191 */
  double e0 = d[0] + nq, e1 = d[0] - nq;
  *norminfd = fabs(e0) > fabs(e1) ? fabs(e0) : fabs(e1);
  *normsqrd = 2 * dotn((size_t)len, d, d);
  free(minus_s);
  free(wsqrt);
  free(d);
}

/* TakeStep(SOCConstraint*) soc_constraint.cc:225-249 */
static void soc_take_step(cxo_constraint* o, double step_size) {
  int n = o->n, len = n + 1;
  double* d1 = (double*)malloc(sizeof(double) * (size_t)n);
  double* expd = (double*)malloc(sizeof(double) * (size_t)len);
  double* wn = (double*)malloc(sizeof(double) * (size_t)len);
  memcpy(d1, o->temp1, sizeof(double) * (size_t)n); /* `auto d1 = temp1_1` copies the Map, aliasing */
  double d0 = o->d0;
  if (step_size != 1.0) {
    d0 *= step_size;
    for (int i = 0; i < n; i++) d1[i] *= step_size;
    memcpy(o->temp1, d1, sizeof(double) * (size_t)n); /* Map copy aliases temp1_1 storage */
  }
  soc_spectral(n, d0, d1, 1, expd);
  soc_quadrep(len, o->W, expd, wn);
  memcpy(o->W, wn, sizeof(double) * (size_t)len);
  free(d1);
  free(expd);
  free(wn);
}

/* GetWeightedSlackEigenvalues(SOCConstraint*) soc_constraint.cc:200-223 */
static void soc_weighted_eigs(cxo_constraint* o, const double* y, double c_weight, double* lmin,
                              double* lmax, double* frob, double* tr) {
  int n = o->n, len = n + 1;
  double* minus_s = (double*)malloc(sizeof(double) * (size_t)len);
  double* wsqrt = (double*)malloc(sizeof(double) * (size_t)len);
  double* Ws = (double*)malloc(sizeof(double) * (size_t)len);
  soc_negative_slack(o, c_weight, y, minus_s);
  soc_spectral(n, o->W[0], o->W + 1, 0, wsqrt);
  soc_quadrep(len, wsqrt, minus_s, Ws);
  double nq = sqrt(dotn((size_t)n, Ws + 1, Ws + 1));
  double e0 = Ws[0] + nq, e1 = Ws[0] - nq;
  double mn = e0 < e1 ? e0 : e1, mx = e0 < e1 ? e1 : e0;
  *lmax = -mn;
  *lmin = -mx;
  *frob = (*lmax) * (*lmax) + (*lmin) * (*lmin);
  *tr = (*lmax) + (*lmin);
  free(minus_s);
  free(wsqrt);
  free(Ws);
}

/* Vars() cone_program.h:59-67 */
/* ConstructSchurComplementSystem(EqualityConstraints*) equality_constraint.cc:14-30 (initialize):
 * G = [0 A^T; A 0] over (clique variables, multipliers), AQc = [0; b], everything else zero */
static void schur_equality(cxo_constraint* o) {
  int r = o->n, mt = o->m, m = mt - r;
  memset(o->G, 0, sizeof(double) * (size_t)mt * mt);
  memset(o->AW, 0, sizeof(double) * (size_t)mt);
  memset(o->AQc, 0, sizeof(double) * (size_t)mt);
  for (int j = 0; j < m; j++)
    for (int i = 0; i < r; i++) {
      double a = o->A[(size_t)j * r + i];
      o->G[(size_t)j * mt + (m + i)] = a; /* bottom-left  */
      o->G[(size_t)(m + i) * mt + j] = a; /* top-right    */
    }
  for (int i = 0; i < r; i++) o->AQc[m + i] = o->C[i];
  o->ip_wc = 0;
  o->ip_cQc = 0;
}

/* ------------------------------------------------------------- Hermitian PSD over R / C / H */
/* ConstructSchurComplementSystem(HermitianPsdConstraint<T>*) hermitian_psd.cc:171-230, initialize */
static void schur_hermitian(cxo_constraint* o) {
  int n = o->n, m = o->m, d = o->hd;
  size_t sz = (size_t)n * n * d;
  double* AW = (double*)malloc(sizeof(double) * sz);
  double* WAW = (double*)malloc(sizeof(double) * sz);
  for (int i = 0; i < m; i++) {
    if (d == 8) { /* octonions are not associative: W A W is Q(W) A, <A, W> stands in for tr(A W)  :181-196 */
      cxo_hc_quadratic_representation(d, n, o->W, o->A + (size_t)i * sz, WAW);
    } else {
      cxo_hc_multiply(d, n, n, n, o->A + (size_t)i * sz, o->W, AW);
      cxo_hc_multiply(d, n, n, n, o->W, AW, WAW);
    }
    for (int j = i; j < m; j++)
      o->G[(size_t)i * m + j] = cxo_hc_trace_inner_product(d, n, o->A + (size_t)j * sz, WAW);
    o->AW[i] = d == 8 ? cxo_hc_trace_inner_product(d, n, o->A + (size_t)i * sz, o->W) : trace_n(n, AW); /* AW.at(0).trace() */
    o->AQc[i] = cxo_hc_trace_inner_product(d, n, o->C, WAW);
  }
  o->ip_wc = 0;
  o->ip_wc += cxo_hc_trace_inner_product(d, n, o->C, o->W);
  cxo_hc_quadratic_representation(d, n, o->W, o->C, WAW);
  o->ip_cQc = cxo_hc_trace_inner_product(d, n, o->C, WAW);
  free(AW);
  free(WAW);
}

/* ComputeNegativeSlack hermitian_psd.h:110-115 */
static void herm_negative_slack(const cxo_constraint* o, double k, const double* y, double* s) {
  size_t sz = (size_t)o->n * o->n * o->hd;
  for (size_t q = 0; q < sz; q++) s[q] = o->C[q] * -k;
  for (int i = 0; i < o->m; i++) {
    const double* Ai = o->A + (size_t)i * sz;
    for (size_t q = 0; q < sz; q++) s[q] = s[q] + Ai[q] * y[i];
  }
}

static void herm_start_vector(const cxo_constraint* o, int id, unsigned long call, double* r) {
  for (int q = 0; q < o->n * o->hd; q++) r[q] = cxo_hc_random((uint64_t)id, call, (uint64_t)q);
}

/* PrepareStep hermitian_psd.cc:34-71 */
static void herm_prepare_step(cxo_constraint* o, int id, unsigned long call, int affine,
                              double c_weight, double e_weight, const double* y, double* normsqrd,
                              double* norminfd) {
  int n = o->n, d = o->hd;
  size_t sz = (size_t)n * n * d;
  double* WS = o->temp1;
  double* minus_s = o->temp2;
  herm_negative_slack(o, c_weight, y, minus_s);
  if (d == 8) { /* PrepareStep(HermitianPsdConstraint<Octonions>*) hermitian_psd.cc:129-145: no affine branch,
                   no e_weight; the infinity norm is the reference's "heuristic approximation" */
    const double trace_ws = cxo_hc_trace_inner_product(d, n, o->W, minus_s);
    cxo_hc_quadratic_representation(d, n, o->W, minus_s, WS);
    *normsqrd = cxo_hc_trace_inner_product(d, n, WS, minus_s) + 2 * trace_ws + n;
    *norminfd = 1.0 / 3.0 * (trace_ws + n);
    return;
  }
  cxo_hc_multiply(d, n, n, n, o->W, minus_s, WS);
  if (affine) {
    double* WSW = (double*)malloc(sizeof(double) * sz);
    cxo_hc_multiply(d, n, n, n, WS, o->W, WSW);
    if (e_weight != 0)
      for (size_t q = 0; q < sz; q++) o->W[q] = o->W[q] * (1 + e_weight);
    for (size_t q = 0; q < sz; q++) o->W[q] = o->W[q] + WSW[q];
    free(WSW);
    return;
  }
  double* r = (double*)malloc(sizeof(double) * (size_t)n * d);
  double* eigs = (double*)malloc(sizeof(double) * (size_t)(n + 2));
  herm_start_vector(o, id, call, r);
  int ne = cxo_hc_approximate_eigenvalues(d, n, WS, o->W, r, n / 2 + 1, eigs);
  double mn = eigs[0], mx = eigs[0];
  for (int i = 1; i < ne; i++) {
    if (eigs[i] < mn) mn = eigs[i];
    if (eigs[i] > mx) mx = eigs[i];
  }
  double l1 = fabs(e_weight + mn), l2 = fabs(e_weight + mx);
  double norminf = l1;
  if (norminf < l2) norminf = l2;
  double* WSWS = (double*)malloc(sizeof(double) * sz);
  cxo_hc_multiply(d, n, n, n, WS, WS, WSWS);
  *norminfd = norminf;
  *normsqrd = trace_n(n, WSWS) + 2 * trace_n(n, WS) + n;
  free(WSWS);
  free(eigs);
  free(r);
}

/* TakeStep hermitian_psd.cc:10-31 */
static void herm_take_step(cxo_constraint* o, double e_weight, double step_size) {
  int n = o->n, d = o->hd;
  size_t nn = (size_t)n * n, sz = nn * d;
  double* WS = o->temp1;
  if (d == 8) { /* TakeStep(HermitianPsdConstraint<Octonions>*) hermitian_psd.cc:116-127 with
                   DoGeodesicUpdateScaled exponential_map.cc:131-144:
                   W <- herm(c^2 W + 2 c k Q(W) s + k^2 Q(W) (Q(s) W)), c = 1.5, k = 0.5 */
    double* ms = o->temp2;
    if (step_size != 1)
      for (size_t q = 0; q < sz; q++) ms[q] = ms[q] * step_size;
    double* wn = (double*)malloc(sizeof(double) * sz);
    cxo_hc_geodesic_update_scaled(d, n, o->W, ms, wn);
    memcpy(o->W, wn, sizeof(double) * sz);
    free(wn);
    return;
  }
  for (int i = 0; i < n; i++) WS[(size_t)i * n + i] += e_weight;
  if (step_size != 1.0)
    for (size_t q = 0; q < sz; q++) WS[q] = WS[q] * step_size;
  double* E = (double*)malloc(sizeof(double) * sz);
  double* T = (double*)malloc(sizeof(double) * sz);
  double* Tc = (double*)malloc(sizeof(double) * sz);
  cxo_hc_exponential_map(d, n, WS, E);
  cxo_hc_multiply(d, n, n, n, E, o->W, T);
  cxo_hc_conj_transpose(d, n, n, T, Tc);
  for (size_t q = 0; q < sz; q++) o->W[q] = (T[q] + Tc[q]) * .5;
  free(E);
  free(T);
  free(Tc);
}

/* GetWeightedSlackEigenvalues hermitian_psd.cc:73-91 */
static void herm_weighted_eigs(cxo_constraint* o, int id, unsigned long call, const double* y,
                               double c_weight, double* lmin, double* lmax, double* frob, double* tr) {
  int n = o->n, d = o->hd;
  size_t sz = (size_t)n * n * d;
  double* minus_s = (double*)malloc(sizeof(double) * sz);
  double* WS = (double*)malloc(sizeof(double) * sz);
  double* WSWS = (double*)malloc(sizeof(double) * sz);
  double* r = (double*)malloc(sizeof(double) * (size_t)n * d);
  double* eigs = (double*)malloc(sizeof(double) * (size_t)(n + 2));
  herm_negative_slack(o, c_weight, y, minus_s);
  if (d == 8) { /* GetWeightedSlackEigenvalues(HermitianPsdConstraint<Octonions>*) hermitian_psd.cc:147-168 */
    cxo_hc_quadratic_representation(d, n, o->W, minus_s, WS);
    const double normsqrd = cxo_hc_trace_inner_product(d, n, WS, minus_s);
    const double tws = cxo_hc_trace_inner_product(d, n, o->W, minus_s);
    *lmax = fabs(normsqrd) / (1e-15 + fabs(tws));
    *lmin = *lmax * .01;
    *tr = -tws;
    *frob = normsqrd;
    free(minus_s);
    free(WS);
    free(WSWS);
    free(r);
    free(eigs);
    return;
  }
  cxo_hc_multiply(d, n, n, n, o->W, minus_s, WS);
  herm_start_vector(o, id, call, r);
  int ne = cxo_hc_approximate_eigenvalues(d, n, WS, o->W, r, n / 2 + 1, eigs);
  double mn = eigs[0], mx = eigs[0];
  for (int i = 1; i < ne; i++) {
    if (eigs[i] < mn) mn = eigs[i];
    if (eigs[i] > mx) mx = eigs[i];
  }
  *lmax = -mn;
  *lmin = -mx;
  cxo_hc_multiply(d, n, n, n, WS, WS, WSWS);
  *frob = trace_n(n, WSWS);
  *tr = -trace_n(n, WS);
  free(minus_s);
  free(WS);
  free(WSWS);
  free(r);
  free(eigs);
}

static void gather_vars(const cxo_program* p, int i, const double* y, double* z) {
  for (int q = 0; q < p->cliques[i].n; q++) z[q] = y[p->cliques[i].d[q]];
}

/* PrepareStep(ConstraintManager*) cone_program.h:69-90 */
void cxo_prepare_step(cxo_program* p, int affine, double c_weight, double e_weight,
                      const double* y, double* info) {
  double normsqrd = 0, norminfd = -1;
  double ni_sq = 0, ni_inf = 0; /* info_i persists across constraints (StepInfo info_i) */
  int maxm = 1;
  for (int i = 0; i < p->K; i++)
    if (p->c[i].m > maxm) maxm = p->c[i].m;
  double* z = (double*)malloc(sizeof(double) * (size_t)maxm);
  for (int i = 0; i < p->K; i++) {
    cxo_constraint* c = &p->c[i];
    gather_vars(p, i, y, z);
    switch (c->type) {
      case CXO_LMI: lmi_prepare_step(c, affine, c_weight, e_weight, z, &ni_sq, &ni_inf); break;
      case CXO_HERMITIAN:
        herm_prepare_step(c, i, p->lanczos_calls, affine, c_weight, e_weight, z, &ni_sq, &ni_inf);
        break;
      case CXO_LINEAR: lin_prepare_step(c, affine, c_weight, e_weight, 1.0, z, &ni_sq, &ni_inf); break;
      case CXO_SOC: soc_prepare_step(c, c_weight, z, &ni_sq, &ni_inf); break;
      case CXO_QUADRATIC: quad_prepare_step(c, c_weight, z, &ni_sq, &ni_inf); break;
      case CXO_EQUALITY: /* equality_constraint.cc:32-37: lambda_ = y.tail(rows) */
        for (int q = 0; q < c->n; q++) c->W[q] = z[c->m - c->n + q];
        ni_sq = 0;
        ni_inf = 0;
        break;
      default: ni_sq = 0; ni_inf = 0; break;
    }
    if (ni_inf > norminfd) norminfd = ni_inf;
    normsqrd += ni_sq;
  }
  p->lanczos_calls++;
  info[0] = normsqrd;
  info[1] = norminfd;
  free(z);
}

void cxo_take_step(cxo_program* p, int affine, double e_weight, double step_size) {
  for (int i = 0; i < p->K; i++) {
    cxo_constraint* c = &p->c[i];
    switch (c->type) {
      case CXO_LMI: lmi_take_step(c, e_weight, step_size); break;
      case CXO_HERMITIAN: herm_take_step(c, e_weight, step_size); break;
      case CXO_LINEAR: lin_take_step(c, affine, step_size); break;
      case CXO_SOC: soc_take_step(c, step_size); break;
      case CXO_QUADRATIC: quad_take_step(c, step_size); break;
      default: break;
    }
  }
}

/* GetWeightedSlackEigenvalues(ConstraintManager*) cone_program.cc:31-57 */
void cxo_weighted_slack_eigenvalues(cxo_program* p, const double* y, double c_weight,
                                    double* out) {
  double frob = 0, tr = 0, lmax = -30000, lmin = 30000;
  int maxm = 1;
  for (int i = 0; i < p->K; i++)
    if (p->c[i].m > maxm) maxm = p->c[i].m;
  double* z = (double*)malloc(sizeof(double) * (size_t)maxm);
  for (int i = 0; i < p->K; i++) {
    cxo_constraint* c = &p->c[i];
    gather_vars(p, i, y, z);
    double t_min = DBL_MAX, t_max = -DBL_MAX, t_frob = 0, t_tr = 0;
    switch (c->type) {
      case CXO_LMI: lmi_weighted_eigs(c, z, c_weight, &t_min, &t_max, &t_frob, &t_tr); break;
      case CXO_HERMITIAN:
        herm_weighted_eigs(c, i, p->lanczos_calls, z, c_weight, &t_min, &t_max, &t_frob, &t_tr);
        break;
      case CXO_LINEAR: lin_weighted_eigs(c, z, c_weight, &t_min, &t_max, &t_frob, &t_tr); break;
      case CXO_SOC: soc_weighted_eigs(c, z, c_weight, &t_min, &t_max, &t_frob, &t_tr); break;
      case CXO_QUADRATIC: quad_weighted_eigs(c, z, c_weight, &t_min, &t_max, &t_frob, &t_tr); break;
      default: break;
    }
    if (lmax < t_max) lmax = t_max;
    if (lmin > t_min) lmin = t_min;
    frob += t_frob;
    tr += t_tr;
  }
  p->lanczos_calls++;
  out[0] = lmin;
  out[1] = lmax;
  out[2] = frob;
  out[3] = tr;
  free(z);
}

/* ------------------------------------------------------- divergence.cc */
typedef struct {
  double frob, trace, lmin, lmax, rank;
} wse_t;

static double solve_rational(double a, double b, double c, double d, double k) { /* :17-22 */
  double ur = b * b - 4 * a * c + 8 * a * k + 2 * b * d * k + pow(d * k, 2);
  return -(b + d * k - sqrt(ur)) / (2 * a);
}
static double inv_lmax_branch(double bound, const wse_t* p) { /* :25-40 */
  double x = solve_rational(p->frob, -2 * p->trace, p->rank, p->lmax, bound);
  double lower = 2.0 / (p->lmax + p->lmin);
  return x >= lower ? x : -1;
}
static int in_limits(double x, double lo, double hi) { return x >= lo && x <= hi; }
static double inv_lmin_branch(double bound, const wse_t* p) { /* :46-82 */
  double lower = 0, upper = 2.0 / (p->lmax + p->lmin), k = -1;
  double a = p->frob / p->lmin, b = 2 * p->trace / p->lmin, n = p->rank / p->lmin, c = bound;
  double ur = b * b + 2 * b * c + c * c - 4 * a * n;
  double f = (b + c + sqrt(ur)) / (2 * a), s = (b + c - sqrt(ur)) / (2 * a);
  if (!(ur < 0)) {
    if (in_limits(f, lower, upper)) k = f;
    if (in_limits(s, lower, upper))
      if (s > k) k = s;
  }
  return k;
}
static int bound_is_finite(double k, const wse_t* p) { /* :84-93 */
  double ni = fabs(k * p->lmax - 1);
  if (ni < fabs(k * p->lmin - 1)) ni = fabs(k * p->lmin - 1);
  return ni < 1;
}
static double div_ub_inverse(double bound, const wse_t* p) { /* :95-110 */
  double k = -1;
  double k1 = inv_lmin_branch(bound, p);
  double k2 = inv_lmax_branch(bound, p);
  if (bound_is_finite(k1, p)) k = k1;
  if (k2 > k && bound_is_finite(k2, p)) k = k2;
  return k;
}
static double div_ub(double k, const wse_t* p) { /* :112-120 */
  double num = k * k * p->frob - 2 * k * p->trace + p->rank;
  double ni = fabs(k * p->lmax - 1);
  if (ni < fabs(k * p->lmin - 1)) ni = fabs(k * p->lmin - 1);
  return num / (1 - ni);
}
/* p5 = {frobenius_norm_squared, trace, lambda_min, lambda_max, rank} */
double cxo_divergence_upper_bound_inverse(double bound, const double* p5) {
  wse_t p = {p5[0], p5[1], p5[2], p5[3], p5[4]};
  return div_ub_inverse(bound, &p);
}
double cxo_divergence_upper_bound(double k, const double* p5) {
  wse_t p = {p5[0], p5[1], p5[2], p5[3], p5[4]};
  return div_ub(k, &p);
}

/* ------------------------------------------------------------- IPM driver */
int cxo_kkt_solve(cxo_program* p, const double* b, double k, double bs, double cs, double* y) {
  cxo_assemble(p);
  if (!cxo_factor(p)) return 0;
  int N = p->md->N;
  for (int i = 0; i < N; i++) y[i] = k * (b[i] * bs + p->sysAQc[i] * cs) - 2 * p->sysAW[i];
  cxo_solve_inplace(p, y);
  return 1;
}

static int rank_of(const cxo_constraint* c) {
  switch (c->type) {
    case CXO_LMI: return c->n;
    case CXO_HERMITIAN: return c->n;
    case CXO_EQUALITY: return 0;
    case CXO_LINEAR: return c->n;
    case CXO_SOC: return 2;
    case CXO_QUADRATIC: return 2; /* quadratic_cone_constraint.h:38 */
    default: return 0;
  }
}

/* ComputeMuFromDivergence cone_program.cc:173-214 */
static double mu_from_divergence(cxo_program* p, const double* AQc_s, double c_weight,
                                 const double* b_s, const cxo_config* cfg, int rankK, double* y) {
  int N = p->md->N;
  for (int i = 0; i < N; i++) y[i] = AQc_s[i] - b_s[i];
  cxo_solve_inplace(p, y);
  double o4[4];
  cxo_weighted_slack_eigenvalues(p, y, c_weight, o4);
  wse_t mp;
  mp.lmin = o4[0];
  mp.lmax = o4[1];
  mp.frob = o4[2];
  mp.trace = o4[3];
  mp.rank = rankK;
  double bound = cfg->divergence_upper_bound * rankK;
  double inv = div_ub_inverse(bound, &mp);
  if (inv == -1) { /* MinimizeNormInf :166-172 */
    inv = -1;
    if (mp.lmin > 0) inv = 2.0 / (mp.lmin + mp.lmax);
  }
  if (inv < 0 && mp.trace > 1e-12) {
    double kstar = mp.trace / mp.frob;
    double nb = 1.5 * (mp.frob * kstar * kstar - 2 * mp.trace * kstar + rankK);
    if (nb > rankK * .7) nb = rankK * .7;
    double a = mp.frob, bb = -2 * mp.trace, c = rankK - nb;
    if (bb * bb - 4 * a * c < 0)
      inv = mp.trace / mp.frob;
    else
      inv = (-bb + sqrt(bb * bb - 4 * a * c)) / (2 * a);
  }
  return inv;
}

#define VREPORT(name, val) \
  if (g_verbose) printf(#name ": %.2e, ", (double)(val));

/* conex::Solve cone_program.cc:235-552 */
/* PerformLineSearch(LinearConstraint*) + FindMinimumMu linear_constraint.cc:48-103.
 * Returns failure (1) when the admissible interval is empty. */
static int lin_line_search(cxo_constraint* o, double c0_weight, double c1_weight, double dinfmax,
                           const double* y0, const double* y1, double* lower, double* upper) {
  int r = o->n;
  double* d0 = o->temp1;
  double* d1 = o->temp2;
  lin_negative_slack(o, c0_weight, y0, d0);
  for (int i = 0; i < r; i++) d0[i] = d0[i] * o->W[i];
  for (int i = 0; i < r; i++) d0[i] += 1;
  lin_negative_slack(o, c1_weight, y1, d1);
  for (int i = 0; i < r; i++) d1[i] = d1[i] * o->W[i];
  for (int i = 0; i < r; i++) d1[i] += 1;
  for (int i = 0; i < r; i++) d1[i] = d1[i] - d0[i];
  double ub = *upper, lb = *lower;
  for (int i = 0; i < r; i++) {
    double ubi = (dinfmax - d0[i]) / d1[i];
    double lbi = (-dinfmax - d0[i]) / d1[i];
    if (lbi > ubi) {
      double t = ubi;
      ubi = lbi;
      lbi = t;
    }
    if (ubi < ub || i == 0) ub = ubi;
    if (lbi > lb || i == 0) lb = lbi;
  }
  *upper = ub;
  *lower = lb;
  return lb > ub;
}

/* ComputeMuFromLineSearch cone_program.cc:118-160.  y0 is the caller's iterate vector and is
 * overwritten, as in the reference. */
static double mu_from_line_search(cxo_program* p, double dinf_upper_bound, const double* AQc_s,
                                  double c_weight, const double* b_s, double* y0) {
  int N = p->md->N;
  for (int q = 0; q < N; q++) y0[q] = -2 * p->sysAW[q];
  cxo_solve_inplace(p, y0);
  double* y1 = (double*)malloc(sizeof(double) * (size_t)N);
  for (int q = 0; q < N; q++) y1[q] = AQc_s[q] + b_s[q] - 2 * p->sysAW[q];
  cxo_solve_inplace(p, y1);
  double c0 = c_weight * 0, c1 = c_weight * 1;
  double out_lb = -DBL_MAX, out_ub = DBL_MAX;
  int maxm = 1;
  for (int i = 0; i < p->K; i++)
    if (p->c[i].m > maxm) maxm = p->c[i].m;
  double* z1 = (double*)malloc(sizeof(double) * (size_t)maxm);
  double* z2 = (double*)malloc(sizeof(double) * (size_t)maxm);
  double result = 0;
  int failed = 0;
  for (int i = 0; i < p->K && !failed; i++) {
    cxo_constraint* c = &p->c[i];
    gather_vars(p, i, y0, z1);
    gather_vars(p, i, y1, z2);
    double lb = -DBL_MAX, ub = DBL_MAX; /* LineSearchOutput output_i */
    switch (c->type) {
      case CXO_LINEAR: failed = lin_line_search(c, c0, c1, dinf_upper_bound, z1, z2, &lb, &ub); break;
      case CXO_STATIC:   /* quadratic_cost.cc:59-65 */
      case CXO_EQUALITY: /* equality_constraint.h:49-54 */
        break;
      default: /* constraint.h:24-28: "Constraint does not support line search." -> failure */
        failed = 1;
        break;
    }
    if (failed) break;
    if (lb > out_lb) out_lb = lb;
    if (ub < out_ub) out_ub = ub;
  }
  if (failed)
    result = -1;
  else
    result = out_lb <= out_ub ? out_ub : -1;
  free(y1);
  free(z1);
  free(z2);
  return result;
}

int cxo_solve(cxo_program* p, const double* bin, const cxo_config* cfg, double* yout) {
  int m = p->num_vars;
  p->solved = 0;
  p->primal_infeasible = 0;
  p->dual_infeasible = 0;
  int max_iters_reached = 1;
  if (p->K == 0) {
    for (int i = 0; i < m; i++) yout[i] = bin[i] * INFINITY;
    return 0;
  }
  { /* cone_program.cc:237-240: quadratic costs need the line search and no rescaling */
    int quad = 0;
    for (int i = 0; i < p->K; i++)
      if (p->c[i].type == CXO_STATIC) quad = 1;
    if (quad && !(cfg->enable_line_search && !cfg->enable_rescaling)) {
      fprintf(stderr, "Must enable line search and disable rescaling for problems with quadratic costs.\n");
      return 0;
    }
  }
  /* Initialize :78-112 */
  if (!p->initialized || cfg->initialization_mode == 0) {
    int keepW = p->initialized && cfg->initialization_mode != 0;
    (void)keepW;
    cxo_initialize(p); /* cold start: W = identity, scalings = 1 */
  }
  free(p->sqrt_inv_mu);
  p->sqrt_inv_mu = (double*)calloc((size_t)(cfg->max_iterations > 0 ? cfg->max_iterations : 1),
                                   sizeof(double));
  p->max_iter = cfg->max_iterations;
  p->num_iter = 0;
  int N = p->md->N;
  double* y = (double*)calloc((size_t)N, sizeof(double));
  double* b = (double*)calloc((size_t)N, sizeof(double));
  double* AQc_s = (double*)calloc((size_t)N, sizeof(double));
  double* b_s = (double*)calloc((size_t)N, sizeof(double));
  for (int i = 0; i < m && i < N; i++) b[i] = bin[i];

  double inv_sqrt_mu_max = cfg->inv_sqrt_mu_max;
  double cx = 1, by = -1, kkt_error = 0;
  double inv_sqrt_mu = 0, e_weight = 1, c_weight = 0, step_size = 1;
  int rankK = 0;
  for (int i = 0; i < p->K; i++) rankK += rank_of(&p->c[i]);
  int centering_steps = 0;
  int warmstart_aborted = 0;
  int initial_centering_steps = cfg->initial_centering_steps_coldstart;
  int initial_centering = 1;
  if (cfg->initialization_mode) initial_centering_steps = cfg->initial_centering_steps_warmstart;

  for (int i = 0; i < cfg->max_iterations; i++) {
    if (i >= initial_centering_steps) initial_centering = 0;
    if (g_verbose) printf(i < 10 ? "i:  %d, " : "i: %d, ", i);
    int final_centering = (inv_sqrt_mu >= inv_sqrt_mu_max) || (kkt_error > cfg->kkt_error_tolerance) ||
                          i >= (cfg->max_iterations - cfg->final_centering_steps);
    int update_mu = (i == 0) || !(initial_centering || final_centering) || warmstart_aborted;
    warmstart_aborted = 0;
    if (final_centering) {
      if (centering_steps >= cfg->final_centering_steps) {
        max_iters_reached = (i >= cfg->max_iterations - 1);
        break;
      }
    }
    cxo_assemble(p);
    if (i < 1 && cfg->enable_rescaling) {
      if (cfg->initialization_mode == 0) {
        double nb = 0, nq = 0;
        for (int q = 0; q < N; q++) {
          nb += b[q] * b[q];
          nq += p->sysAQc[q] * p->sysAQc[q];
        }
        p->b_scaling = 1.0 / (1 + sqrt(nb));
        p->c_scaling = 1.0 / (1 + sqrt(nq));
      }
      double mu_target = 1.0 / (inv_sqrt_mu_max * inv_sqrt_mu_max);
      mu_target *= (p->b_scaling * p->c_scaling);
      inv_sqrt_mu_max = 1.0 / sqrt(mu_target);
    }
    double bs = p->b_scaling, cs = p->c_scaling;
    if (!cxo_factor(p)) {
      if (i == 0 && cfg->initialization_mode == 1) {
        cxo_set_identity(p);
        warmstart_aborted = 1;
        continue;
      }
      p->solved = 0;
      if (g_verbose) printf("Status: Factorization failed.\n\n");
      free(y);
      free(b);
      free(AQc_s);
      free(b_s);
      return 0;
    }
    if (update_mu) {
      double temp = -1;
      for (int q = 0; q < N; q++) {
        AQc_s[q] = p->sysAQc[q] * cs;
        b_s[q] = b[q] * bs;
      }
      if (cfg->enable_line_search) { /* cone_program.cc:376-384 */
        temp = mu_from_line_search(p, cfg->dinf_upper_bound, AQc_s, cs, b_s, y);
        if (temp < 0) temp = inv_sqrt_mu;
      }
      if (temp < 0) temp = mu_from_divergence(p, AQc_s, cs, b_s, cfg, rankK, y);
      if (temp > 0)
        inv_sqrt_mu = temp;
      else
        inv_sqrt_mu *= .5;
    } else {
      if (initial_centering == 0) centering_steps++;
    }
    {
      double mx = inv_sqrt_mu_max;
      double mn = sqrt(1.0 / (1e-15 + cfg->maximum_mu));
      if (inv_sqrt_mu > mx) inv_sqrt_mu = mx;
      if (inv_sqrt_mu < mn) inv_sqrt_mu = mn;
    }
    for (int q = 0; q < N; q++)
      y[q] = inv_sqrt_mu * (b[q] * bs + p->sysAQc[q] * cs) - 2 * p->sysAW[q];
    cxo_solve_inplace(p, y);
    e_weight = 1;
    c_weight = inv_sqrt_mu * cs;
    double info[2];
    cxo_prepare_step(p, 0, c_weight, e_weight, y, info);
    step_size = 2.0 / (info[1] * info[1]);
    if (step_size > 1) step_size = 1;
    if (i == 0 && cfg->initialization_mode == 1 && info[1] >= cfg->warmstart_abort_threshold) {
      cxo_set_identity(p);
      warmstart_aborted = 1;
    } else {
      cxo_take_step(p, 0, e_weight, step_size);
    }
    double d_2 = sqrt(fabs(info[0]));
    double d_inf = fabs(info[1]);
    by = dotn((size_t)N, b, y) * 1.0 / (inv_sqrt_mu * cs);
    cx = 2 * p->sys_wc + dotn((size_t)N, p->sysAQc, y) - inv_sqrt_mu * p->sys_cQc * cs;
    cx /= (inv_sqrt_mu * bs);
    double mu = 1.0 / inv_sqrt_mu;
    mu *= mu;
    double s_dot_x = mu * (rankK - d_2 * d_2) / (bs * cs);
    mu = mu / (cs * bs);
    VREPORT(mu, mu);
    VREPORT(d_2, d_2);
    VREPORT(d_inf, d_inf);
    VREPORT(by, by);
    VREPORT(cx, cx);
    kkt_error = fabs(cx - by - s_dot_x) / s_dot_x;
    VREPORT(kkt_error, kkt_error);
    p->num_iter = i + 1;
    p->sqrt_inv_mu[i] = inv_sqrt_mu;
    if (g_verbose) printf("\n");
    if (final_centering || inv_sqrt_mu >= inv_sqrt_mu_max) {
      if (d_inf <= cfg->final_centering_tolerance) {
        max_iters_reached = 0;
        break;
      }
    }
  }
  for (int i = 0; i < m; i++) yout[i] = y[i];
  double mu = 1.0 / inv_sqrt_mu;
  mu *= mu;
  if (mu > cfg->infeasibility_threshold) {
    p->solved = 0;
    p->primal_infeasible = cx * inv_sqrt_mu <= -.5;
    p->dual_infeasible = by * inv_sqrt_mu >= .5;
  } else {
    p->solved = 1;
  }
  if (cfg->prepare_dual_variables) { /* :500-516 */
    cxo_assemble(p);
    cxo_factor(p);
    double* y2 = (double*)calloc((size_t)N, sizeof(double));
    for (int q = 0; q < N; q++) y2[q] = inv_sqrt_mu * b[q] * p->b_scaling - 1 * p->sysAW[q];
    cxo_solve_inplace(p, y2);
    double info[2];
    cxo_prepare_step(p, 1, 0, 0, y2, info);
    free(y2);
  }
  if (p->solved) {
    for (int i = 0; i < m; i++) yout[i] /= inv_sqrt_mu;
    for (int i = 0; i < m; i++) yout[i] /= p->c_scaling;
  }
  if (p->solved) {
    if (max_iters_reached) p->solved = 0;
  }
  if (g_verbose) printf("Status: %s\n\n", p->solved ? "Solved." : "Not solved.");
  free(y);
  free(b);
  free(AQc_s);
  free(b_s);
  return p->solved;
}

int cxo_num_iterations(const cxo_program* p) { return p->num_iter; }
/* CONEX_GetIterationStats (interfaces/conex.cc:259-285): mu of iteration i (negative: from the end) */
double cxo_iteration_mu(const cxo_program* p, int iter) {
  if (iter < 0) iter = p->num_iter + iter;
  if (iter < 0 || iter >= p->num_iter) return -1.0;
  return 1.0 / (p->sqrt_inv_mu[iter] * p->sqrt_inv_mu[iter]);
}

/* Program::GetDualVariable cone_program.h:120-134 */
void cxo_get_dual_variable(cxo_program* p, int i, double* out) {
  int n = cxo_dual_size(p, i);
  memcpy(out, p->c[i].W, sizeof(double) * (size_t)n);
  if (!p->primal_infeasible && p->num_iter > 0) {
    double s = p->sqrt_inv_mu[p->num_iter - 1] * p->b_scaling;
    for (int q = 0; q < n; q++) out[q] /= s;
  }
}

/* --------------------------------------------------- stand-alone KAT hooks */
int cxo_path_in_tree(int x, int y, int n, const int* parent, const int* depth, int* path) {
  (void)n;
  ivec v;
  iv_init(&v);
  cxo_path_in_tree_iv(x, y, parent, depth, &v);
  int len = v.n;
  memcpy(path, v.d, sizeof(int) * (size_t)len);
  iv_free(&v);
  return len;
}

int cxo_pick_clique_order(int K, const int* ptr, const int* idx, int root, int* order,
                          int* sn_ptr, int* sn_idx, int* sep_ptr, int* sep_idx) {
  ivec* cl = ivs_new(K);
  ivec* sn = ivs_new(K);
  ivec* sp = ivs_new(K);
  for (int i = 0; i < K; i++) {
    for (int q = ptr[i]; q < ptr[i + 1]; q++) iv_push(&cl[i], idx[q]);
    iv_sort(&cl[i]);
  }
  cxo_pick_clique_order_iv(K, cl, NULL, root, order, sn, sp, NULL, NULL);
  sn_ptr[0] = 0;
  sep_ptr[0] = 0;
  for (int i = 0; i < K; i++) {
    for (int q = 0; q < sn[i].n; q++) sn_idx[sn_ptr[i] + q] = sn[i].d[q];
    sn_ptr[i + 1] = sn_ptr[i] + sn[i].n;
    for (int q = 0; q < sp[i].n; q++) sep_idx[sep_ptr[i] + q] = sp[i].d[q];
    sep_ptr[i + 1] = sep_ptr[i] + sp[i].n;
  }
  ivs_free(cl, K);
  ivs_free(sn, K);
  ivs_free(sp, K);
  return K;
}

void cxo_pade(int n, const double* arg, double* result) { cxo_pade_expm(n, arg, result); }
int cxo_lanczos_asym(int n, const double* WS, const double* W, const double* r, int iters,
                     double* eigs) {
  return cxo_asymmetric_lanczos(n, WS, W, r, iters, eigs);
}
int cxo_lanczos_sym(int n, const double* A, const double* r0, int iters, double* eigs) {
  return cxo_symmetric_lanczos(n, A, r0, iters, eigs);
}
int cxo_jacobi(int n, const double* A, const double* W, const double* r0, int iters,
               double* eigs) {
  return cxo_jacobi_eigenvalues(n, A, W, r0, iters, eigs);
}
int cxo_tridiag_eigs(int n, const double* d, const double* e, double* out) {
  return cxo_tridiagonal_eigenvalues(n, d, e, out);
}

struct cxo_ws_handle {
  cxo_workspace* w;
};
cxo_ws_handle* cxo_ws_new(int K, const int* ptr, const int* idx, const int* supernode_size) {
  ivec* path = ivs_new(K);
  for (int i = 0; i < K; i++)
    for (int q = ptr[i]; q < ptr[i + 1]; q++) iv_push(&path[i], idx[q]);
  cxo_ws_handle* h = (cxo_ws_handle*)malloc(sizeof(cxo_ws_handle));
  h->w = cxo_workspace_new(K, path, supernode_size);
  ivs_free(path, K);
  return h;
}
void cxo_ws_free(cxo_ws_handle* h) {
  if (!h) return;
  cxo_workspace_free(h->w);
  free(h);
}
int cxo_ws_N(const cxo_ws_handle* h) { return h->w->N; }
long cxo_ws_slab_size(const cxo_ws_handle* h) { return h->w->slab_size; }
double* cxo_ws_slab(cxo_ws_handle* h) { return h->w->slab; }
void cxo_ws_offsets(const cxo_ws_handle* h, long* diag_off, long* offd_off) {
  memcpy(diag_off, h->w->diag_off, sizeof(long) * (size_t)h->w->K);
  memcpy(offd_off, h->w->offd_off, sizeof(long) * (size_t)h->w->K);
}
int cxo_ws_ss_index(const cxo_ws_handle* h, int e, long* out) {
  if (out) memcpy(out, h->w->ss_index[e], sizeof(long) * (size_t)h->w->ss_count[e]);
  return h->w->ss_count[e];
}
int cxo_ws_cholesky(cxo_ws_handle* h) { return cxo_block_cholesky(h->w); }
void cxo_ws_forward(cxo_ws_handle* h, double* y) { cxo_apply_block_inverse(h->w, y); }
void cxo_ws_backward(cxo_ws_handle* h, double* y) { cxo_apply_block_inverse_of_transpose(h->w, y); }
void cxo_ws_to_dense(const cxo_ws_handle* h, double* out) { cxo_workspace_to_dense(h->w, out); }
/* BlockLDLTInPlace + ApplyBlockInverseOfMD + ApplyBlockInverseOfMTranspose on a raw workspace
 * (block_triangular_operations.cc:315-349, 222-299), as block_triangular_operations_test.cc:183-212 uses them */
int cxo_ws_ldlt(cxo_ws_handle* h) { return cxo_block_ldlt(h->w); }
void cxo_ws_solve_ldlt(cxo_ws_handle* h, double* y) { cxo_solve_ldlt(h->w, y); }
