/*
 * conex oracle -- public (ctypes-facing) API.  TEST INFRASTRUCTURE ONLY: see
 * cxo_internal.h for the rules.  Every function cites the reference code it
 * restates in the matching .c file.
 */
#ifndef CONEX_ORACLE_H
#define CONEX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
  CXO_LMI = 0,       /* DenseLMIConstraint   dense_lmi_constraint.{h,cc}, psd_constraint.{h,cc} */
  CXO_LINEAR = 1,    /* LinearConstraint     linear_constraint.{h,cc} */
  CXO_SOC = 2,       /* SOCConstraint        soc_constraint.{h,cc} */
  CXO_STATIC = 3,    /* SupernodalAssemblerStatic supernodal_assembler.h:122-129 (fixed G) */
  CXO_HERMITIAN = 4, /* HermitianPsdConstraint<Real|Complex|Quaternions> hermitian_psd.{h,cc} */
  CXO_EQUALITY = 5,  /* EqualityConstraints equality_constraint.{h,cc} (multipliers -> LDLT path) */
  CXO_QUADRATIC = 6  /* QuadraticConstraint  quadratic_cone_constraint.{h,cc}: Lorentz cone x0 >= sqrt(x1' Q x1) */
};

/* cone_program.h:17-38 */
typedef struct {
  int prepare_dual_variables;
  int initialization_mode;
  double inv_sqrt_mu_max;
  double minimum_mu;
  double maximum_mu;
  double divergence_upper_bound;
  int enable_line_search;
  double dinf_upper_bound;
  int final_centering_steps;
  double final_centering_tolerance;
  int initial_centering_steps_warmstart;
  int initial_centering_steps_coldstart;
  double warmstart_abort_threshold;
  int max_iterations;
  double infeasibility_threshold;
  double kkt_error_tolerance;
  int kkt_solver;
  int enable_rescaling;
  int iterative_refinement_iterations;
} cxo_config;

typedef struct cxo_program cxo_program;

void cxo_default_config(cxo_config* c);

/* ---- program construction (cone_program.h:99-233, constraint_manager.h) ---- */
cxo_program* cxo_program_new(int num_vars);
void cxo_program_free(cxo_program* p);
/* A: m matrices n*n col-major; C: n*n; vars: m variable ids (NULL = 0..num_vars-1).
 * returns constraint id, or -1 on rejection (IsUnique failure). */
int cxo_add_lmi(cxo_program* p, int n, int m, const double* A, const double* C, const int* vars);
/* A: r x m col-major, c: r */
int cxo_add_linear(cxo_program* p, int r, int m, const double* A, const double* c,
                   const int* vars);
/* A: (n+1) x m col-major, c: n+1 */
int cxo_add_soc(cxo_program* p, int n, int m, const double* A, const double* c, const int* vars);
/* QuadraticConstraint(Q, A, c): Q n x n col-major or NULL (identity: the two-argument constructor),
 * A: (n+1) x m col-major (row 0 = A0, the rest A1), c: n+1 */
int cxo_add_quadratic(cxo_program* p, int n, int m, const double* Q, const double* A, const double* c,
                      const int* vars);
/* A y[vars] = b with r rows: appends r multipliers to the KKT system (constraint_manager.h:66-90);
 * the factorization switches to BlockLDLTInPlace (kkt_solver.cc:180-193).  dual size = r (lambda). */
int cxo_add_equality(cxo_program* p, int r, int m, const double* A, const double* b,
                     const int* vars);
/* 1 when the last LDLT factorization clamped a pivot to +-1e-9 (kkt_solver.cc:190-192) */
int cxo_factor_regularized(const cxo_program* p);
/* Hermitian PSD cone over R (d=1), C (d=2) or H (d=4): A = m x d planes of n x n (column-major),
 * C = d planes.  W / dual variable: d planes.  hermitian_psd.h:41-116 */
int cxo_add_hermitian(cxo_program* p, int n, int d, int m, const double* A, const double* C,
                      const int* vars);
/* stand-alone hooks for the hyper-complex algebra (KATs against numpy complex / quaternion) */
void cxo_hc_multiply(int d, int r, int k, int c, const double* X, const double* Y, double* Z);
void cxo_hc_exponential_map(int d, int n, const double* x, double* y);
int cxo_hc_approximate_eigenvalues(int d, int n, const double* WS, const double* W, const double* r,
                                   int num_iter, double* eigs);
double cxo_hc_random(unsigned long id, unsigned long call, unsigned long idx);
/* fixed m x m Schur block (SupernodalAssemblerStatic) */
int cxo_add_static(cxo_program* p, int m, const double* G, const int* vars);
int cxo_num_constraints(const cxo_program* p);

/* Initialize(): symbolic analysis + Bind + workspace; W = identity (cone_program.cc:78-112) */
int cxo_initialize(cxo_program* p);

/* ---- symbolic getters (for bit-exact parity) ---- */
int cxo_system_size(const cxo_program* p);   /* N */
int cxo_get_order(const cxo_program* p, int* order /*K*/);
int cxo_get_permutation(const cxo_program* p, int* perm, int* perm_inv /*num_vars each*/);
/* which: 0 cliques(permuted) 1 supernodes_orig 2 separators_orig 3 supernodes_pos
 *        4 separators_pos 5 pc_supernodes (by original clique id) 6 pc_separators
 * returns length; out may be NULL to query */
int cxo_get_list(const cxo_program* p, int which, int e, int* out);
int cxo_get_supernode_sizes(const cxo_program* p, int* out /*K*/);
long cxo_slab_size(const cxo_program* p);
int cxo_get_block_offsets(const cxo_program* p, long* diag_off, long* offd_off /*K each*/);
int cxo_get_ss_index(const cxo_program* p, int e, long* out); /* returns count */

/* ---- numeric path ---- */
void cxo_set_identity(cxo_program* p);
int cxo_dual_size(const cxo_program* p, int i);
void cxo_get_W(const cxo_program* p, int i, double* out);
void cxo_set_W(cxo_program* p, int i, const double* in);
/* solver->Assemble() + AssembleSchurComplementResiduals (cone_program.cc:338-341) */
void cxo_assemble(cxo_program* p);
void cxo_get_slab(const cxo_program* p, double* out);
void cxo_get_constraint_schur(const cxo_program* p, int i, double* G /*m*m*/, double* AW,
                              double* AQc, double* scalars /*2*/);
void cxo_get_residuals(const cxo_program* p, double* AW /*N*/, double* AQc /*N*/,
                       double* scalars /*2: <w,c>, <c,Qc>*/);
int cxo_factor(cxo_program* p); /* 1 = success */
void cxo_solve_inplace(cxo_program* p, double* y /*N*/);
/* SetIterativeRefinementIterations: the next cxo_factor keeps the dense assembled matrix and every
 * solve then runs `iterations` refinement steps (kkt_solver.cc:177-179, 233-261) */
void cxo_set_refinement(cxo_program* p, int iterations);
void cxo_kkt_matrix(const cxo_program* p, double* out /*N*N, original variable order*/);
/* PrepareStep over all constraints (cone_program.h:69-90); info = {normsqrd, norminfd} */
void cxo_prepare_step(cxo_program* p, int affine, double c_weight, double e_weight,
                      const double* y, double* info);
void cxo_take_step(cxo_program* p, int affine, double e_weight, double step_size);
/* out = {lambda_min, lambda_max, frobenius_norm_squared, trace} (cone_program.cc:31-57) */
void cxo_weighted_slack_eigenvalues(cxo_program* p, const double* y, double c_weight,
                                    double* out);
/* one full KKT solve at the current W: assemble, factor, rhs, solve (BASELINE metric).
 * y = k*(b*b_scaling + AQc*c_scaling) - 2 AW ; returns factor status */
int cxo_kkt_solve(cxo_program* p, const double* b, double inv_sqrt_mu, double b_scaling,
                  double c_scaling, double* y);
/* conex::Solve(b, prog, config, y) cone_program.cc:235-552; returns solved flag */
int cxo_solve(cxo_program* p, const double* b, const cxo_config* cfg, double* y);
int cxo_num_iterations(const cxo_program* p);
double cxo_iteration_mu(const cxo_program* p, int iter); /* CONEX_GetIterationStats conex.cc:259-285 */
void cxo_get_dual_variable(cxo_program* p, int i, double* out);
void cxo_set_verbose(int v);
/* 1: direct_update only when the supernode's positions are exactly 0..m-1 (fixes a reference defect
 * on fill-in structures, see cxo_program.c); 0 (default): the reference's test as written */
void cxo_set_strict_direct_update(int on);

/* ---- stand-alone pieces for known-answer tests ---- */
int cxo_path_in_tree(int x, int y, int n, const int* parent, const int* depth, int* path);
/* flat clique input: ptr[K+1], idx[]; outputs flattened the same way (caller sizes them) */
int cxo_pick_clique_order(int K, const int* ptr, const int* idx, int root, int* order,
                          int* sn_ptr, int* sn_idx, int* sep_ptr, int* sep_idx);
void cxo_pade(int n, const double* arg, double* result);
int cxo_lanczos_asym(int n, const double* WS, const double* W, const double* r, int iters,
                     double* eigs);
int cxo_lanczos_sym(int n, const double* A, const double* r0, int iters, double* eigs);
int cxo_jacobi(int n, const double* A, const double* W, const double* r0, int iters,
               double* eigs);
int cxo_tridiag_eigs(int n, const double* d, const double* e, double* out);
double cxo_divergence_upper_bound_inverse(double bound, const double* p5);
double cxo_divergence_upper_bound(double k, const double* p5);

/* raw supernodal workspace (block_triangular_operations_test.cc style) */
typedef struct cxo_ws_handle cxo_ws_handle;
cxo_ws_handle* cxo_ws_new(int K, const int* ptr, const int* idx, const int* supernode_size);
void cxo_ws_free(cxo_ws_handle* h);
int cxo_ws_N(const cxo_ws_handle* h);
long cxo_ws_slab_size(const cxo_ws_handle* h);
double* cxo_ws_slab(cxo_ws_handle* h);
void cxo_ws_offsets(const cxo_ws_handle* h, long* diag_off, long* offd_off);
int cxo_ws_ss_index(const cxo_ws_handle* h, int e, long* out);
int cxo_ws_cholesky(cxo_ws_handle* h);
void cxo_ws_forward(cxo_ws_handle* h, double* y);
void cxo_ws_backward(cxo_ws_handle* h, double* y);
void cxo_ws_to_dense(const cxo_ws_handle* h, double* out);
int cxo_ws_ldlt(cxo_ws_handle* h);                  /* BlockLDLTInPlace :315-349 */
void cxo_ws_solve_ldlt(cxo_ws_handle* h, double* y); /* inv(M D) then inv(M^T) :222-299 */

#ifdef __cplusplus
}
#endif
#endif
