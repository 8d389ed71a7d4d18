/* SeDuMi-format front end of libconex.so: the reference's MATLAB entry point
 *     [x, y, info] = conex(A, b, c, K, pars)            interfaces/matlab/conex.m:2-82
 * and the preprocessing it calls, as C entry points (the reference runs this part in MATLAB and
 * reaches the solver through loadlibrary, interfaces/matlab/util/ConexProgram.m:29-100; a C / Python
 * / MATLAB caller of THIS library gets the same pipeline without MATLAB):
 *     CleanLinear                 util/CleanLinear.m:1-30          rows of [A b] that are not all zero
 *     coneBase.Symmetrize         util/coneBase.m:180-190          A, c averaged with their transposes
 *     ConexPreprocess, blkdiagPrg util/ConexPreprocess.m:19-33, util/blkdiagPrg.m:17-38
 *     BuildMask                   util/BuildMask.m:1-85            support closure: which entries of
 *                                                                  every PSD block can be nonzero
 *     BinaryPsdCompletion         util/BinaryPsdCompletion.m:1-17  connected components of that support
 *                                                                  = the blocks a PSD block splits into
 *     ExtractConstraintMatrices   util/ExtractConstraintMatrices.m:1-48  per block: its variables
 *                                                                  (rows of A) and their matrices
 * The problem is SeDuMi's dual form
 *     maximize b'y   subject to   c - A'y in K,      A: m x N (sparse), N = sum_i K.s(i)^2,
 * every PSD block stored column-major.  As in conex.m only K.s is supported (K.f, K.l, K.q, K.r must
 * be empty or zero: conex.m:7-15 raises "Cone not supported yet" for l, q, r; free variables need
 * util/EliminateFreeVars.m's sparse null space and are refused here too).
 *
 * What the preprocessing does: it is NOT a chordal conversion with overlap constraints.  It finds,
 * per PSD block, the coordinate subspace the data can reach (entries of c, closed under "a row of A
 * that touches the support brings all its entries") and splits the block into the connected
 * components of that support, each completed to a dense diagonal block.  A block-diagonal SDP given
 * as one large (arbitrarily permuted) LMI thus becomes several small LMIs, each over the variables
 * that touch it -- which is what feeds the clique path of the solver.
 *
 * One documented divergence from conex.m: options.blkdiag = 1 with a SINGLE PSD block.  conex.m
 * preprocesses the problem and then, in its length(K.s) == 1 branch (conex.m:45-49), hands the
 * ORIGINAL matrices to AddDenseLinearMatrixInequality reshaped with the REDUCED block's order --
 * a reshape error (or a wrong program) whenever the preprocessing changed anything.  This front end
 * gives the solver the preprocessed blocks in that case as in every other (the single block is
 * split into its connected components; tests/test_sedumi_frontend.py pins it).  With the default
 * (blkdiag = -1: off for one block) and with blkdiag = 0 a single block is solved as one dense LMI
 * over the cleaned rows, exactly as conex.m does.
 */
#ifndef CONEX_SEDUMI_H
#define CONEX_SEDUMI_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int blkdiag; /* pars.blkdiag: 1 block-diagonalise, 0 do not, -1 conex.m's default (iff more than one PSD block) */
  int errors;  /* pars.errors: fill info.errors (conex.m:79-82) */
  int verbose;
} CONEX_SedumiOptions;

typedef struct {
  int solved;        /* CONEX_Maximize's return value (1 = solved) */
  int pinf, dinf;    /* info.pinf = info.dinf = ~solved (conex.m:65-66) */
  double cpusec;     /* wall time of CONEX_Maximize (conex.m:62-64) */
  double errors[2];  /* |c'x - b'y| and c'x - b'y when options.errors */
  int num_blocks;    /* PSD blocks handed to the solver */
  int num_rows_kept; /* rows of A left after CleanLinear (= variables of the solver) */
} CONEX_SedumiInfo;

/* The whole of conex.m.  A as triplets (0-based row < m, col < N), K.s = Ks[0 .. num_psd).
 * x: N doubles (primal, block by block column-major), y: m doubles.  Returns CONEX_SUCCESS (0) when
 * the pipeline ran (info->solved tells how the solver ended), 1 on invalid input. */
int CONEX_SolveSedumi(long m, long N, long nnz, const long* A_row, const long* A_col, const double* A_val,
                      const double* b, const double* c, int num_psd, const long* Ks,
                      const CONEX_SedumiOptions* options_or_null, double* x, double* y,
                      CONEX_SedumiInfo* info_or_null);

/* The preprocessing alone (what conex.m has before it builds the program), for inspection and
 * tests.  blkdiag as above.  Returns a handle (NULL on invalid input), released by CONEX_SedumiFree. */
void* CONEX_SedumiPreprocess(long m, long N, long nnz, const long* A_row, const long* A_col, const double* A_val,
                             const double* b, const double* c, int num_psd, const long* Ks, int blkdiag);
void CONEX_SedumiFree(void* handle);
int CONEX_SedumiNumBlocks(const void* handle);
/* rows of the input that survive both CleanLinear passes, ascending (count = number of solver variables) */
long CONEX_SedumiKeptRows(const void* handle, long* rows_or_null);
/* blkdiagPrg.indx: the columns of the input the reduced problem keeps, in the reduced order */
long CONEX_SedumiKeptColumns(const void* handle, long* cols_or_null);
/* the reduced cost vector b (one entry per kept row) */
long CONEX_SedumiReducedB(const void* handle, double* b_or_null);
int CONEX_SedumiBlockOrder(const void* handle, int block);
int CONEX_SedumiBlockNumVariables(const void* handle, int block);
/* ExtractConstraintMatrices: the block's variables (indices into the kept rows, ascending), their
 * matrices (order x order x num_variables, column-major: matrix_conex_format) and the affine term */
int CONEX_SedumiBlockData(const void* handle, int block, long* variables, double* matrices, double* affine);

#ifdef __cplusplus
}
#endif
#endif
