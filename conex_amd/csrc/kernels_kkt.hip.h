// Supernodal KKT kernels: deterministic gather-assembly of the slab, level-scheduled
// left-looking block Cholesky, and level-scheduled block triangular solves.
//
// Reference semantics reproduced here (summation ORDER included, so results do not depend
// on scheduling):
//   SupernodalAssemblerBase::UpdateBlocks (Set/SetLowerTri/Scatter)  supernodal_assembler.cc:113-165
//   SupernodalKKTSolver::Assemble (descending elimination index)     kkt_solver.cc:164-170
//   AssembleSchurComplementResiduals                                 constraint_manager.h:107-124
//   BlockCholeskyInPlace                                             block_triangular_operations.cc:184-219
//   ApplyBlockInverseInPlace / ...OfTransposeInPlace                 block_triangular_operations.cc:114-182
//
// The reference pushes updates through tables of double*; here every target entry PULLS its
// contributions from an index list built on the host in the reference's own order.  Entries
// are owned by exactly one thread, so no atomics are needed and runs are bit-reproducible.
#pragma once
#include "device_utils.h"

namespace cxk {

// slab[dst[t]] = sum_k G[src[k]], k in [ptr[t], ptr[t+1]) ; src < 0 means structural zero.
__global__ void gather_slab(int64_t T, const int64_t* __restrict__ dst,
                            const int* __restrict__ ptr, const int64_t* __restrict__ src,
                            const double* __restrict__ G, double* __restrict__ slab) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < T;
       t += (int64_t)gridDim.x * blockDim.x) {
    double s = 0;
    for (int k = ptr[t]; k < ptr[t + 1]; k++) {
      const int64_t q = src[k];
      if (q >= 0) s += G[q];
    }
    slab[dst[t]] = s;
  }
}

// Residual vectors in permuted order + the two scalars (fixed-order sums).
__global__ void gather_residuals(int N, const int* __restrict__ ptr, const int64_t* __restrict__ src,
                                 const double* __restrict__ AWc, const double* __restrict__ AQcc,
                                 double* __restrict__ AW, double* __restrict__ AQc, int K,
                                 const double* __restrict__ sc, double* __restrict__ sys_sc) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int p = tid; p < N; p += gridDim.x * blockDim.x) {
    double a = 0, q = 0;
    for (int k = ptr[p]; k < ptr[p + 1]; k++) {
      a += AWc[src[k]];
      q += AQcc[src[k]];
    }
    AW[p] = a;
    AQc[p] = q;
  }
  if (blockIdx.x == 0) {  // <w,c> and <c,Qc>: fixed-order strided partial sums + block sum
    __shared__ double red[8];
    double s0 = 0, s1 = 0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
      s0 += sc[2 * i];
      s1 += sc[2 * i + 1];
    }
    s0 = BlockSum(s0, red);
    s1 = BlockSum(s1, red);
    if (threadIdx.x == 0) {
      sys_sc[0] = s0;
      sys_sc[1] = s1;
    }
  }
}

// y = k (b bs + AQc cs) - 2 AW   (cone_program.cc:409-411), all in permuted order
__global__ void build_rhs(int N, double k, double bs, double cs, const double* __restrict__ b,
                          const double* __restrict__ AQc, const double* __restrict__ AW,
                          double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = k * (b[p] * bs + AQc[p] * cs) - 2 * AW[p];
}

// y = AQc cs - b bs  (ComputeMuFromDivergence cone_program.cc:181)
__global__ void build_mu_rhs(int N, double bs, double cs, const double* __restrict__ b,
                             const double* __restrict__ AQc, double* __restrict__ y) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x)
    y[p] = AQc[p] * cs - b[p] * bs;
}

struct FactorPlan {
  // per supernode
  const int* ns;             // [K]
  const int* nsep;           // [K]
  const int* start;          // [K] first permuted index
  const int64_t* diag_off;   // [K]
  const int64_t* offd_off;   // [K]
  // Cholesky pull lists: targets of supernode p are [tg_ptr[p], tg_ptr[p+1])
  const int* tg_ptr;         // [K+1]
  const int64_t* tg_dst;     // slab offset of target entry
  const int* tr_ptr;         // [T+1] triples of target t
  const int64_t* tr_colk;    // slab offset of child column k
  const int64_t* tr_colj;    // slab offset of child column j
  const int* tr_len;         // child supernode size
  // forward-solve pull lists per permuted row
  const int* fs_ptr;         // [N+1]
  const int64_t* fs_col;     // slab offset of child's off-diagonal column
  const int* fs_start;       // start of the child's segment in the vector
  const int* fs_len;
  // backward: separator rows in the reference's accumulation order
  const int* bs_ptr;         // [K+1]
  const int* bs_c;           // column index c within off block
  const int* bs_row;         // permuted index of separator variable
};

// One workgroup per supernode of the level.  Phase 1 pulls the Schur updates of all
// finished descendants, phase 2 factors the diagonal block in LDS (right-looking, one column
// at a time), phase 3 solves L^{-1} * off.  If `rhs` != nullptr the forward substitution of
// the right-hand side is fused in (same dependency structure).
__global__ void __launch_bounds__(256)
chol_level(FactorPlan P, const int* __restrict__ level_sn, double* __restrict__ slab,
           double* __restrict__ rhs, int* __restrict__ fail) {
  extern __shared__ double lds[];
  const int p = level_sn[blockIdx.x];
  const int ns = P.ns[p], s = P.nsep[p];
  double* D = slab + P.diag_off[p];
  double* B = slab + P.offd_off[p];
  double* sD = lds;            // ns x ns
  double* sB = lds + ns * ns;  // ns x s
  double* sb = sB + ns * s;    // ns (rhs segment)

  // ---- phase 1: pull updates  S_S -= off_i[:,k] . off_i[:,j]  in increasing child index
  for (int t = P.tg_ptr[p] + threadIdx.x; t < P.tg_ptr[p + 1]; t += blockDim.x) {
    double acc = slab[P.tg_dst[t]];
    for (int q = P.tr_ptr[t]; q < P.tr_ptr[t + 1]; q++) {
      const double* ck = slab + P.tr_colk[q];
      const double* cj = slab + P.tr_colj[q];
      const int len = P.tr_len[q];
      double dot = 0;
      for (int r = 0; r < len; r++) dot = fma(ck[r], cj[r], dot);
      acc -= dot;
    }
    slab[P.tg_dst[t]] = acc;
  }
  if (rhs) {
    const int st = P.start[p];
    for (int r = threadIdx.x; r < ns; r += blockDim.x) {
      double acc = rhs[st + r];
      for (int q = P.fs_ptr[st + r]; q < P.fs_ptr[st + r + 1]; q++) {
        const double* col = slab + P.fs_col[q];
        const double* bi = rhs + P.fs_start[q];
        const int len = P.fs_len[q];
        double dot = 0;
        for (int k = 0; k < len; k++) dot = fma(col[k], bi[k], dot);
        acc -= dot;
      }
      sb[r] = acc;
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < ns * ns; q += blockDim.x) sD[q] = D[q];
  for (int q = threadIdx.x; q < ns * s; q += blockDim.x) sB[q] = B[q];
  __syncthreads();

  // ---- phase 2: LLT of the diagonal block (lower), column by column
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  for (int k = 0; k < ns; k++) {
    const double akk = sD[k + k * ns];
    if (!(akk > 0.0)) {
      if (threadIdx.x == 0) {
        s_bad = 1;
        atomicExch(fail, 1);
      }
      break;  // uniform: akk is the same value for every thread
    }
    const double d = sqrt(akk);
    __syncthreads();
    for (int i = k + threadIdx.x; i < ns; i += blockDim.x)
      sD[i + k * ns] = (i == k) ? d : sD[i + k * ns] / d;
    __syncthreads();
    const int rem = ns - k - 1;
    // trailing update of the lower triangle: A[i][j] -= L[i][k] L[j][k], j > k, i >= j
    for (int idx = threadIdx.x; idx < rem * rem; idx += blockDim.x) {
      const int j = k + 1 + idx / rem, i = k + 1 + idx % rem;
      if (i >= j) sD[i + j * ns] -= sD[i + k * ns] * sD[j + k * ns];
    }
    __syncthreads();
  }
  __syncthreads();
  if (s_bad) return;

  // ---- phase 3: off <- L^{-1} off ; rhs <- L^{-1} rhs  (forward substitution by columns)
  const int ncols = s + (rhs ? 1 : 0);
  for (int k = 0; k < ns; k++) {
    const double d = sD[k + k * ns];
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
      double* col = (c < s) ? sB + c * ns : sb;
      col[k] /= d;
    }
    __syncthreads();
    const int rem = ns - k - 1;
    for (int idx = threadIdx.x; idx < rem * ncols; idx += blockDim.x) {
      const int c = idx / rem, i = k + 1 + idx % rem;
      double* col = (c < s) ? sB + c * ns : sb;
      col[i] -= sD[i + k * ns] * col[k];
    }
    __syncthreads();
  }
  for (int q = threadIdx.x; q < ns * ns; q += blockDim.x) {
    const int i = q % ns, j = q / ns;
    if (i >= j) D[q] = sD[q];
  }
  for (int q = threadIdx.x; q < ns * s; q += blockDim.x) B[q] = sB[q];
  if (rhs)
    for (int r = threadIdx.x; r < ns; r += blockDim.x) rhs[P.start[p] + r] = sb[r];
}

// Forward substitution only (factor already done):  b_p <- L_p^{-1} (b_p - sum_i off_i^T b_i)
__global__ void __launch_bounds__(64)
forward_level(FactorPlan P, const int* __restrict__ level_sn, const double* __restrict__ slab,
              double* __restrict__ rhs) {
  extern __shared__ double lds[];
  const int p = level_sn[blockIdx.x];
  const int ns = P.ns[p];
  const double* D = slab + P.diag_off[p];
  double* sb = lds;
  const int st = P.start[p];
  for (int r = threadIdx.x; r < ns; r += blockDim.x) {
    double acc = rhs[st + r];
    for (int q = P.fs_ptr[st + r]; q < P.fs_ptr[st + r + 1]; q++) {
      const double* col = slab + P.fs_col[q];
      const double* bi = rhs + P.fs_start[q];
      const int len = P.fs_len[q];
      double dot = 0;
      for (int k = 0; k < len; k++) dot = fma(col[k], bi[k], dot);
      acc -= dot;
    }
    sb[r] = acc;
  }
  __syncthreads();
  for (int k = 0; k < ns; k++) {
    if (threadIdx.x == 0) sb[k] /= D[k + (size_t)k * ns];
    __syncthreads();
    const double bk = sb[k];
    for (int i = k + 1 + threadIdx.x; i < ns; i += blockDim.x) sb[i] -= D[i + (size_t)k * ns] * bk;
    __syncthreads();
  }
  for (int r = threadIdx.x; r < ns; r += blockDim.x) rhs[st + r] = sb[r];
}

// Backward substitution:  b_j <- L_j^{-T} (b_j - sum_c off_j[:,c] y[sep_j[c]])
__global__ void __launch_bounds__(64)
backward_level(FactorPlan P, const int* __restrict__ level_sn, const double* __restrict__ slab,
               double* __restrict__ rhs) {
  extern __shared__ double lds[];
  const int p = level_sn[blockIdx.x];
  const int ns = P.ns[p];
  const double* D = slab + P.diag_off[p];
  const double* B = slab + P.offd_off[p];
  double* sb = lds;
  const int st = P.start[p];
  for (int r = threadIdx.x; r < ns; r += blockDim.x) {
    double acc = rhs[st + r];
    for (int q = P.bs_ptr[p]; q < P.bs_ptr[p + 1]; q++)
      acc -= B[r + (size_t)P.bs_c[q] * ns] * rhs[P.bs_row[q]];
    sb[r] = acc;
  }
  __syncthreads();
  for (int k = ns - 1; k >= 0; k--) {
    // y_k = (b_k - sum_{i>k} L[i][k] y_i) / L[k][k]
    double part = 0;
    for (int i = k + 1 + threadIdx.x; i < ns; i += blockDim.x) part += D[i + (size_t)k * ns] * sb[i];
    part = WaveSum(part);
    if (threadIdx.x == 0) sb[k] = (sb[k] - part) / D[k + (size_t)k * ns];
    __syncthreads();
  }
  for (int r = threadIdx.x; r < ns; r += blockDim.x) rhs[st + r] = sb[r];
}

// permuted <-> original order copies
__global__ void permute_gather(int N, const int* __restrict__ idx, const double* __restrict__ in,
                               double* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
    out[i] = in[idx[i]];
}

}  // namespace cxk
