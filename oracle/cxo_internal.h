/*
 * conex oracle -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the Newton-step KKT path of
 * ToyotaResearchInstitute/conex (reference @ /root/reference).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (conex_amd/) never links or calls it.
 *
 * Parity status: the reference cannot be compiled here (Eigen 3.3.9 is an
 * un-vendored http_archive, WORKSPACE:5-13), so this restatement is pinned by
 * the reference's own literal known-answer tests (see tests/test_oracle_kat.py)
 * and by dense numpy/scipy cross-checks.  Integer outputs (clique order,
 * supernodes, separators, permutation, index tables) are bit-exact to the
 * reference algorithm; floating point is pinned to the tolerances the
 * reference's tests use (1e-12 LLT/solve, 1e-7 Pade).
 */
#ifndef CXO_INTERNAL_H
#define CXO_INTERNAL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- growable int vector ---------- */
typedef struct {
  int* d;
  int n;
  int cap;
} ivec;

void iv_init(ivec* v);
void iv_free(ivec* v);
void iv_clear(ivec* v);
void iv_push(ivec* v, int x);
void iv_copy(ivec* dst, const ivec* src);
void iv_sort(ivec* v);
/* std::set_intersection / set_union / set_difference on sorted inputs */
void iv_intersection(const ivec* a, const ivec* b, ivec* out);
void iv_union(const ivec* a, const ivec* b, ivec* out);
void iv_difference(const ivec* a, const ivec* b, ivec* out);
ivec* ivs_new(int n);
void ivs_free(ivec* v, int n);

/* ---------- symbolic analysis ---------- */
typedef struct {
  int K; /* number of cliques */
  int N; /* sum of supernode sizes */
  int num_vars; /* GetMax(cliques)+1 */
  int* clique_order;           /* [K]  position e -> original clique index */
  ivec* cliques;               /* [K]  permuted labels: supernode first, then separators */
  ivec* supernodes_orig;       /* [K]  original labels (MatrixData.supernodes_original_labels) */
  ivec* separators_orig;       /* [K]  original labels, ordered by permuted label */
  ivec* supernodes_pos;        /* [K]  after RelabelCliques: position in constraint, -1 = fill-in */
  ivec* separators_pos;        /* [K] */
  int* supernode_size;         /* [K] */
  int* permutation;            /* [num_vars] original -> eliminated position */
  int* permutation_inverse;    /* [num_vars] */
  /* raw PickCliqueOrder outputs, indexed by ORIGINAL clique index */
  ivec* pc_supernodes;
  ivec* pc_separators;
  int* tree_parent;
  int* tree_height;
} cxo_matrix_data;

void cxo_path_in_tree_iv(int x, int y, const int* parent, const int* depth, ivec* path);

/* clique_ordering.cc:307-333 ; valid_leaf may be NULL (== empty vector) */
void cxo_pick_clique_order_iv(int K, const ivec* cliques_sorted, const int* valid_leaf, int root,
                              int* order, ivec* supernodes, ivec* separators, int* tree_parent,
                              int* tree_height);

/* kkt_solver.cc:70-102 */
int cxo_get_root_node(int K, const ivec* cliques, const ivec* dual_vars);

/* supernodal_solver.cc:376-431 + kkt_solver.cc:47-68 */
cxo_matrix_data* cxo_matrix_data_build(int K, const ivec* cliques, const ivec* dual_vars);
/* kkt_solver.cc:118-131 (explicit order/supernodes/separators) */
cxo_matrix_data* cxo_matrix_data_from_supernodes(int K, const ivec* cliques, int num_vars,
                                                 const int* order, const ivec* supernodes,
                                                 const ivec* separators);
void cxo_matrix_data_free(cxo_matrix_data* d);

/* ---------- supernodal storage (row A7) ---------- */
typedef struct {
  int K;
  int N;
  int* supernode_size;   /* [K] */
  ivec* snodes;          /* [K] permuted labels */
  ivec* separators;      /* [K] permuted labels */
  long* diag_off;        /* [K] offset of n_s x n_s col-major block in slab */
  long* offd_off;        /* [K] offset of n_s x s col-major block */
  long slab_size;
  double* slab;          /* owned, zero initialised */
  int* var_to_sn;        /* [N] */
  int* var_to_pos;       /* [N] */
  /* column_intersections[sn], sn = 0..K-2 ; intersection_position[sn][k] pairs */
  ivec* col_int;         /* [K-1] list of j */
  ivec* col_int_start;   /* [K-1] start index of pair list k (len = n+1) */
  ivec* pair_first;      /* [K-1] flat */
  ivec* pair_second;     /* [K-1] flat */
  long** ss_index;       /* [K] s(s+1)/2 slab offsets (seperator_diagonal) */
  int* ss_count;         /* [K] */
  double* temporaries;   /* max separator size */
  /* LDLT mode (kkt_solver.cc:180-193): per-supernode transpositions of Eigen::RLDLT */
  int* transpositions;   /* [N], local indices inside each supernode */
  int factored_ldlt;     /* last factorization was BlockLDLTInPlace */
  int regularized;       /* a pivot was clamped to +-1e-9 */
} cxo_workspace;

cxo_workspace* cxo_workspace_new(int K, const ivec* path, const int* supernode_size);
void cxo_workspace_free(cxo_workspace* w);

/* ---------- numeric kernels ---------- */
/* block_triangular_operations.cc:184-219 ; returns 1 on success */
int cxo_block_cholesky(cxo_workspace* w);
/* BlockLDLTInPlace :315-349 with Eigen::RLDLT (RLDLT.h:294-431); returns 1 when no pivot was
 * regularised */
int cxo_block_ldlt(cxo_workspace* w);
/* SolveInPlaceLDLT: ApplyBlockInverseOfMD :265-299 then ApplyBlockInverseOfMTranspose :222-263 */
void cxo_solve_ldlt(const cxo_workspace* w, double* y);
/* rldlt_inplace<Lower>::unblocked RLDLT.h:298-431; returns 1 when no regularisation was used */
int cxo_rldlt_inplace(int n, double* a, int lda, int* transpositions);
/* :160-182 */
void cxo_apply_block_inverse(const cxo_workspace* w, double* y);
/* :114-151 */
void cxo_apply_block_inverse_of_transpose(const cxo_workspace* w, double* y);
/* supernodal_solver.cc:264-273 (lower triangle incl. structural zeros) */
void cxo_workspace_to_dense(const cxo_workspace* w, double* out /* N*N col-major */);

/* dense helpers (column major) */
int cxo_llt_inplace(int n, double* a, int lda);
void cxo_pade_expm(int n, const double* arg, double* result);
/* approximate_eigenvalues.cc:178-239 ; returns count of eigenvalues written */
int cxo_asymmetric_lanczos(int n, const double* WS, const double* W, const double* r,
                           int num_iter, double* eigs);
/* approximate_eigenvalues.cc:10-95 (JacobiSolver), returns n eigenvalues */
int cxo_jacobi_eigenvalues(int n, const double* A, const double* W, const double* r0, int iters,
                           double* eigs);
/* approximate_eigenvalues.cc:119-144 symmetric Lanczos */
int cxo_symmetric_lanczos(int n, const double* A, const double* r0, int num_iter, double* eigs);
/* eigenvalues of symmetric tridiagonal (diag d[n], offdiag e[n-1]) ascending */
int cxo_tridiagonal_eigenvalues(int n, const double* d, const double* e, double* out);

/* cxo_hermitian.c -- hyper-complex (d real planes) matrix algebra, jordan_matrix_algebra.cc */
#include <stdint.h>
double cxo_hc_random(uint64_t id, uint64_t call, uint64_t idx);
void cxo_hc_multiply(int d, int r, int k, int c, const double* X, const double* Y, double* Z);
void cxo_hc_conj_transpose(int d, int r, int c, const double* X, double* Z);
double cxo_hc_trace_inner_product(int d, int n, const double* X, const double* Y);
void cxo_hc_quadratic_representation(int d, int n, const double* x, const double* y, double* out);
void cxo_hc_geodesic_update_scaled(int d, int n, const double* w, const double* s, double* out);
int cxo_hc_approximate_eigenvalues(int d, int n, const double* WS, const double* W, const double* r,
                                   int num_iter, double* eigs);
void cxo_hc_exponential_map(int d, int n, const double* x, double* y);

#ifdef __cplusplus
}
#endif
#endif
