// Argument structs shared by the LMI kernel files and their host launchers (plain data, no
// device code: safe to include from several translation units).
#pragma once
#include <cstdint>

namespace cxk {

struct LmiGroup {
  int n;
  int m;
  int count;
  const double* A;  // count x a_stride: the m matrices (n*n each) of a member, col-major
  // doubles between the A blocks of consecutive members: m n^2, or (m+1) n^2 for groups whose
  // kernel wants C stored right behind the A_i (lmi_schur_mfma); C itself always also lives in `C`
  long long a_stride;
  const double* C;  // count x (n*n)
  double* W;        // count x (n*n)
  double* T1;       // count x (n*n)   temp_1 of WorkspaceDensePSD (WS between Prepare/TakeStep)
  const int* ids;   // member -> constraint id
  // lower triangles of the A_i, packed column by column (n (n + 1) / 2 doubles each, count x m of them),
  // for the kernels that only need sum_i y_i A_i of symmetric data (the slack of lmi_prepare_rows:
  // half the bytes of the full matrices); nullptr where no such copy is kept
  const double* Apk;
  // 0: DenseLMIConstraint semantics.  d in {1,2,4}: HermitianPsdConstraint over R / C / H stored
  // through its real representation (order n = d * hyper-complex order): outputs carry the
  // factor 1/d (tr over the representation = d * Re tr), TakeStep uses the reference's
  // Taylor-squaring exponential and the eigenvalue estimates its random-start Lanczos
  // (hermitian_psd.cc:10-91, exponential_map.cc:15-43, jordan_matrix_algebra.cc:386-452).
  int herm_d;
  // Sparse groups (kernels_lmi_sparse.hip.h): A is not stored densely.  Nonzeros (both triangles)
  // matrix-major -- entries of (member, i) at [sp_eptr[mem*m+i], sp_eptr[mem*m+i+1]), sp_erc =
  // row | col << 16 -- and position-major for the slack: entries of (member, position q) at
  // [sp_pptr[mem*n*n+q], ...), variable index ascending.  All null for dense groups.
  const int* sp_eptr;
  const int* sp_erc;
  const int* sp_pairs;  // pair number t = i (i + 1) / 2 + j -> i | j << 16, 0 <= j <= i <= m (row m: the list of C)
  const double* sp_eval;
  const int* sp_pptr;
  const int* sp_pvar;
  const double* sp_pval;
};

struct Arena {
  double* G;              // per-constraint m x m Schur blocks (lower triangle meaningful)
  const int64_t* g_off;   // [K]
  double* AWc;            // per-constraint AW / AQc
  double* AQcc;
  const int64_t* r_off;   // [K]
  double* sc;             // [2K] <w,c>, <c,Qc>
};

struct StepArgs {
  const double* y;        // permuted Newton direction (device)
  // not null: the direction is still the three solutions of cxk_factor_solve_triple_async and is combined where it
  // is read, y_i = k (y3[i] + y3[st + i]) - 2 y3[2 st + i] with k = y3_k[0] (StepY: the expression of
  // newton_from_three, the same bits); extra workgroups of the launch write y out (StepTail::ny)
  const double* y3;
  long long y3_stride;
  const double* y3_k;
  const int* cl_ptr;      // [K+1] clique pointer
  const int* cl_perm;     // permuted index of each clique variable
  double* info;           // per-constraint outputs (2 or 4 doubles each)
  int affine;
  double c_weight;
  double e_weight;
  double step_size;
  // not null: the step length is taken on the device from the reduced norms of the PrepareStep
  // just before, step = min(1, 2 / norminfd^2) with norminfd = step_from[1] (cone_program.cc:417-418),
  // so that TakeStep can be enqueued without a host round trip in between
  const double* step_from;
  // not null: c_weight = cw_from[0] * cw_scale, the barrier parameter the device selected
  // (cxk_select_mu_async) times c_scaling (cone_program.cc:413): CWeightOf, every PrepareStep kernel
  const double* cw_from;
  double cw_scale;
  // not null (TakeStep enqueued before the host has seen the factorization's outcome): leave W alone
  // when the factorization failed -- skip_if[0] != 0, or skip_if[1] == skip_tag != 0 (MailboxFailValue)
  const int* skip_if;
  int skip_tag;
  unsigned long long call;  // index of this PrepareStep / eigenvalue query (Hermitian start vectors)
  // reference identity (the default; off: cxk_set_reference_identity(ctx, 0) / CXK_REFERENCE_QUIRKS=0):
  // the Ritz values go out exactly as approximate_eigenvalues.cc:178-239 produces them, without the
  // Samuelson clamp
  int no_clamp;
};

#ifdef __HIPCC__
__device__ __forceinline__ bool StepSkipped(const StepArgs& sa) {
  return sa.skip_if && (sa.skip_if[0] != 0 || (sa.skip_tag != 0 && sa.skip_if[1] == sa.skip_tag));
}
__device__ __forceinline__ double YFromThree(const double* __restrict__ y3, long long st, double k, int i) {
  return k * (y3[i] + y3[st + i]) - 2.0 * y3[2 * st + i];
}
__device__ __forceinline__ double StepY(const StepArgs& sa, int i) {
  return sa.y3 ? YFromThree(sa.y3, sa.y3_stride, sa.y3_k[0], i) : sa.y[i];
}
__device__ __forceinline__ double CWeightOf(const StepArgs& sa) {
  return sa.cw_from ? sa.cw_from[0] * sa.cw_scale : sa.c_weight;
}
__device__ __forceinline__ double StepSizeOf(const StepArgs& sa) {
  if (!sa.step_from) return sa.step_size;
  const double v = sa.step_from[1];
  const double s = 2.0 / (v * v);
  return s > 1 ? 1.0 : s;
}
#endif

}  // namespace cxk
