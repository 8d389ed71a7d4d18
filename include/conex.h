/*
 * conex.h -- outer C-ABI of libconex.so (MI355X build).
 *
 * Binary-compatible with the reference's interfaces/conex.h:7-99: same 21 entry points, same
 * argument order and types, same CONEX_SolverConfiguration field order, same status polarity
 * (builders/updaters: CONEX_SUCCESS = 0 / CONEX_FAILURE = 1 after a "file line: msg" line on
 * stderr, error_checking_macros.h:15-19; CONEX_Maximize / CONEX_Solve: 1 = solved, 0 = not
 * solved, cone_program.cc:532).  Existing callers (the SWIG/numpy Python wrapper
 * interfaces/python/conex.i:17-29, MATLAB loadlibrary interfaces/matlab/util/ConexProgram.m,
 * or -lconex C programs like interfaces/test/test_app.cc) relink unchanged.
 *
 * What differs is behind the boundary: the per-iteration Newton step (Schur assembly,
 * supernodal Cholesky, triangular solves, geodesic update) runs on the GPU through the cxk_*
 * interface of conex_kkt_hip.h; the IPM control loop (mu selection, stopping rules) is host
 * C++ restating cone_program.cc:235-552.  All input arrays are copied at call time
 * (interfaces/conex.cc:143-223); outputs are caller-allocated; a program handle is not
 * thread-safe; separate handles are independent.
 *
 * Matrices are column-major (Fortran order) as in the reference.
 */
#ifndef CONEX_API_H
#define CONEX_API_H
#ifdef __cplusplus
extern "C" {
#endif

typedef int CONEX_STATUS;
enum { CONEX_SUCCESS = 0, CONEX_FAILURE = 1 };

/* Mirror of conex::SolverConfiguration (cone_program.h:17-38) in the ABI's field order
 * (interfaces/conex.h:10-30; note iterative_refinement_iterations sits after max_iterations
 * and kkt_solver is last). */
typedef struct {
  int prepare_dual_variables;            /* recover X = W / (sqrt_inv_mu * b_scaling) at the end */
  int initialization_mode;               /* 0 cold start (W = e), 1 warm start (keep W) */
  double inv_sqrt_mu_max;                /* target 1/sqrt(mu), default 1000 */
  double minimum_mu;
  double maximum_mu;
  double divergence_upper_bound;         /* x rank(K) = bound used by the mu rule */
  int enable_line_search;
  double dinf_upper_bound;
  int final_centering_steps;
  double final_centering_tolerance;
  int initial_centering_steps_warmstart;
  int initial_centering_steps_coldstart;
  double warmstart_abort_threshold;
  int max_iterations;
  int iterative_refinement_iterations;   /* dense-KKT refinement: not available on device (0) */
  double infeasibility_threshold;
  double kkt_error_tolerance;
  int enable_rescaling;
  int kkt_solver;                        /* 0 LLT (device); LDLT/QR modes: see DESIGN.md */
} CONEX_SolverConfiguration;

typedef struct {
  double mu;
  int iteration_number;
} CONEX_IterationStats;

typedef struct {
  int iterations;
} CONEX_SolutionStats;

/* interfaces/conex.cc:129-135 */
void* CONEX_CreateConeProgram();
void CONEX_DeleteConeProgram(void*);

/* interfaces/conex.cc:216-229 : rows Ar of A (Ar x Ac) form  c - A y >= 0 ; returns constraint id */
int CONEX_AddDenseLinearConstraint(void* prog, const double* A, int Ar, int Ac,
                                   const double* c, int cr);

/* interfaces/conex.cc:190-215 : lb <= A y <= ub, rows normalised as the reference does.
 * Returns -1 (as the reference does). Rows with lb == ub need the equality/LDLT path, which is
 * not on the device yet: such a call is rejected with a message. */
int CONEX_AddLinearInequalities(void* prog, const double* A, int Ar, int Ac,
                                const double* lb, int num_lb, const double* ub,
                                int num_ub);

/* interfaces/conex.cc:343-354 */
int CONEX_AddQuadraticCost(void* prog, const double* A, int Ar, int Ac);

/* interfaces/conex.cc:137-160 : Aarray = m matrices (n x n, column-major), cmat n x n */
int CONEX_AddDenseLMIConstraint(void* prog, const double* Aarray, int Aarrayr,
                                int Aarrayc, int m, const double* cmat, int cr,
                                int cc);

/* interfaces/conex.cc:162-188 : as above on the variable subset vars[0..m) */
int CONEX_AddSparseLMIConstraint(void* prog, const double* Aarray, int Aarrayr,
                                 int Aarrayc, int m, const double* cmat, int cr,
                                 int cc, const long* vars, int vars_c);

/* interfaces/conex.cc:93-112 : maximise b'y ; returns 1 when solved */
int CONEX_Maximize(void* prog, const double* b, int br,
                   const CONEX_SolverConfiguration* config, double* y, int yr);

int CONEX_Solve(void* prog, const CONEX_SolverConfiguration* config, double* y,
                int yr);

/* interfaces/conex.cc:114-127 */
void CONEX_GetDualVariable(void* prog, int i, double* x, int xr, int xc);

int CONEX_GetDualVariableSize(void* prog_ptr, int i);

/* interfaces/conex.cc:231-257 (this build also zeroes iterative_refinement_iterations and
 * kkt_solver, which the reference leaves indeterminate) */
void CONEX_SetDefaultOptions(CONEX_SolverConfiguration* config);

/* interfaces/conex.cc:259-285 : negative iter_num counts from the end */
void CONEX_GetIterationStats(void* prog, CONEX_IterationStats* stats,
                             int iter_num);

/* interfaces/conex.cc:365-373 */
CONEX_STATUS CONEX_UpdateLinearOperator(void* program, int constraint,
                                        double value, int variable, int row,
                                        int col, int hyper_complex_dim);

/* interfaces/conex.cc:287-316 : hyper_complex_dim in {1, 2, 4, 8} */
CONEX_STATUS CONEX_NewLinearMatrixInequality(void* program, int order,
                                             int hyper_complex_dim,
                                             int* constraint_id);

/* interfaces/conex.cc:375-382 */
CONEX_STATUS CONEX_UpdateAffineTerm(void* program, int constraint, double value,
                                    int row, int col, int hyper_complex_dim);

/* interfaces/conex.cc:384-397 */
CONEX_STATUS CONEX_NewLorentzConeConstraint(void* program, int order,
                                            int* constraint_id);

/* interfaces/conex.cc:318-329 */
CONEX_STATUS CONEX_NewLinearInequality(void* program, int num_rows,
                                       int* constraint_id);

/* interfaces/conex.cc:331-341, 356-363 */
CONEX_STATUS CONEX_NewQuadraticCost(void* p, int* constraint_id);
CONEX_STATUS CONEX_UpdateQuadraticCostMatrix(void* p, int id, double value,
                                             int row, int col);

/* interfaces/conex.cc:399-407 */
CONEX_STATUS CONEX_SetNumberOfVariables(void* program, int m);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif
