/*
 * conex.h -- outer C-ABI of libconex.so (MI355X build).
 *
 * Binary-compatible with the reference's interfaces/conex.h:7-99: the same 21 entry points with
 * the same argument order and types, the same CONEX_SolverConfiguration field order, the same
 * status polarity.  Existing callers -- the SWIG/numpy wrapper (interfaces/python/conex.i:17-29),
 * MATLAB loadlibrary (interfaces/matlab/util/ConexProgram.m), C programs linking -lconex
 * (interfaces/test/test_app.cc) -- relink unchanged.
 *
 * Status conventions (kept from the reference):
 *   builders / updaters   CONEX_SUCCESS = 0, CONEX_FAILURE = 1 after a "file line: msg" line on
 *                         stderr (error_checking_macros.h:15-19)
 *   CONEX_Add*            the id of the new constraint (CONEX_AddLinearInequalities: -1)
 *   CONEX_Maximize/Solve  1 = solved, 0 = not solved (cone_program.cc:532)
 *
 * Behind the boundary the per-iteration Newton step (Schur assembly, supernodal Cholesky / LDLT,
 * triangular solves, geodesic update) runs on the GPU through conex_kkt_hip.h; the IPM control
 * loop is host C++ restating cone_program.cc:235-552.  Input arrays are copied at call time
 * (interfaces/conex.cc:143-223), outputs are caller-allocated, matrices are column-major, a
 * program handle is not thread-safe, separate handles are independent.
 *
 * Default behaviour: the reference algorithm as written, results equal to rounding.  Two
 * corrections of the reference are available and OFF by default -- a constraint's Schur block
 * scattered by variable position also where the reference misplaces it (rare fill-in structures,
 * supernodal_assembler.cc:72-91), and the Lanczos eigenvalue estimates that set the step length
 * clamped to a bound that provably contains the spectrum (psd_constraint.cc:45-84,
 * approximate_eigenvalues.cc:178-239 return them unclamped).  CXK_REFERENCE_QUIRKS=0 in the
 * environment (or the extra symbol CONEX_HIP_SetReferenceIdentity(program, 0)) switches both
 * on; without them CONEX_Maximize reproduces the reference algorithm's mu sequence iteration
 * for iteration.
 */
#ifndef CONEX_API_H
#define CONEX_API_H
#ifdef __cplusplus
extern "C" {
#endif

typedef int CONEX_STATUS;
enum { CONEX_SUCCESS = 0, CONEX_FAILURE = 1 };

/* conex::SolverConfiguration (cone_program.h:17-38) in the ABI's field order
 * (interfaces/conex.h:10-30: iterative_refinement_iterations follows max_iterations,
 * kkt_solver is last). */
typedef struct {
  int prepare_dual_variables;          /* recover X = W / (sqrt_inv_mu * b_scaling) at the end */
  int initialization_mode;             /* 0 cold start (W = e), 1 warm start (keep W) */
  double inv_sqrt_mu_max;              /* target 1/sqrt(mu), default 1000 */
  double minimum_mu;
  double maximum_mu;
  double divergence_upper_bound;       /* times rank(K): bound used by the divergence mu rule */
  int enable_line_search;              /* line-search mu rule (needed for quadratic costs) */
  double dinf_upper_bound;
  int final_centering_steps;
  double final_centering_tolerance;
  int initial_centering_steps_warmstart;
  int initial_centering_steps_coldstart;
  double warmstart_abort_threshold;
  int max_iterations;
  int iterative_refinement_iterations; /* refinement steps per solve (device: supernodal mat-vec) */
  double infeasibility_threshold;
  double kkt_error_tolerance;
  int enable_rescaling;
  int kkt_solver;                      /* accepted, ignored: LLT, or LDLT when equalities exist */
} CONEX_SolverConfiguration;

typedef struct {
  double mu;
  int iteration_number;
} CONEX_IterationStats;

typedef struct {
  int iterations;
} CONEX_SolutionStats;

/* ------------------------------------------------------------------ program lifetime / options
 * interfaces/conex.cc:129-135, 399-407, 231-257 */
void* CONEX_CreateConeProgram();
void CONEX_DeleteConeProgram(void* program);
CONEX_STATUS CONEX_SetNumberOfVariables(void* program, int number_of_variables);
/* also zeroes iterative_refinement_iterations and kkt_solver, which the reference leaves unset */
void CONEX_SetDefaultOptions(CONEX_SolverConfiguration* options);

/* ------------------------------------------------------------------ whole-constraint builders
 * Dense data, copied.  interfaces/conex.cc:137-229, 343-354 */

/* c - A y >= 0 over all variables; A is rows x cols */
int CONEX_AddDenseLinearConstraint(void* program,
                                   const double* A, int A_rows, int A_cols,
                                   const double* c, int c_rows);

/* lb <= A y <= ub, every row normalised as PreprocessLinearInequality does
 * (linear_constraint.cc:14-46); rows with lb == ub become an equality block with multipliers
 * (LDLT path).  Returns -1, as the reference does. */
int CONEX_AddLinearInequalities(void* program,
                                const double* A, int A_rows, int A_cols,
                                const double* lb, int lb_rows,
                                const double* ub, int ub_rows);

/* y' Q y cost on the variables whose diagonal entry of Q is positive */
int CONEX_AddQuadraticCost(void* program, const double* Q, int Q_rows, int Q_cols);

/* C - sum_i y_i A_i >= 0 (PSD); A_stack holds num_matrices matrices of order x order */
int CONEX_AddDenseLMIConstraint(void* program,
                                const double* A_stack, int order_rows, int order_cols,
                                int num_matrices,
                                const double* C, int C_rows, int C_cols);

/* the same over the variable subset variables[0 .. num_matrices) */
int CONEX_AddSparseLMIConstraint(void* program,
                                 const double* A_stack, int order_rows, int order_cols,
                                 int num_matrices,
                                 const double* C, int C_rows, int C_cols,
                                 const long* variables, int variables_rows);

/* ------------------------------------------------------------------ entry-by-entry builders
 * interfaces/conex.cc:287-397 */

/* Hermitian PSD cone over R / C / H / O: hyper_complex_dim in {1, 2, 4, 8}
 * (octonions, hyper_complex_dim = 8: order at most 3, interfaces/conex.cc:310-311) */
CONEX_STATUS CONEX_NewLinearMatrixInequality(void* program, int order, int hyper_complex_dim,
                                             int* constraint_id);
CONEX_STATUS CONEX_NewLorentzConeConstraint(void* program, int order, int* constraint_id);
CONEX_STATUS CONEX_NewLinearInequality(void* program, int num_rows, int* constraint_id);
CONEX_STATUS CONEX_NewQuadraticCost(void* program, int* constraint_id);

CONEX_STATUS CONEX_UpdateLinearOperator(void* program, int constraint, double value,
                                        int variable, int row, int col, int hyper_complex_dim);
CONEX_STATUS CONEX_UpdateAffineTerm(void* program, int constraint, double value,
                                    int row, int col, int hyper_complex_dim);
CONEX_STATUS CONEX_UpdateQuadraticCostMatrix(void* program, int constraint, double value,
                                             int row, int col);

/* ------------------------------------------------------------------ solve
 * interfaces/conex.cc:93-112.  Maximise b'y (CONEX_Solve: the cost accumulated so far) subject to
 * the constraints; y has y_rows = number of variables entries.  Returns 1 when solved. */
int CONEX_Maximize(void* program, const double* b, int b_rows,
                   const CONEX_SolverConfiguration* options, double* y, int y_rows);
int CONEX_Solve(void* program, const CONEX_SolverConfiguration* options, double* y, int y_rows);

/* ------------------------------------------------------------------ results
 * interfaces/conex.cc:114-127, 259-285 */
int CONEX_GetDualVariableSize(void* program, int constraint);
void CONEX_GetDualVariable(void* program, int constraint, double* x, int x_rows, int x_cols);
/* iteration < 0 counts from the last one */
void CONEX_GetIterationStats(void* program, CONEX_IterationStats* stats, int iteration);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif
