// Hermitian matrices over the octonions (order <= 3: the exceptional Jordan algebra and its
// subalgebras), the reference's HermitianPsdConstraint<Octonions> -- hermitian_psd.cc:116-168
// (TakeStep, PrepareStep, GetWeightedSlackEigenvalues: the reference's own heuristic rules),
// :171-230 (ConstructSchurComplementSystem, the Octonions branches), jordan_matrix_algebra.cc:101-177
// (Multiply through the sign / index tables, JordanMultiply, QuadraticRepresentation), :204-210
// (TraceInnerProduct), exponential_map.cc:131-144 (DoGeodesicUpdateScaled).
//
// The octonions are not associative, so there is no real matrix representation to hand to the LMI
// kernels: W A W becomes the quadratic representation Q(W) A = 2 W o (W o A) - (W o W) o A of the
// Jordan product x o y = (x y + y x) / 2, every product formed plane by plane through the
// multiplication table.  A matrix is 8 planes of n x n (plane-major, each column-major); one
// 64-thread workgroup per cone, the planes in LDS, one lane per entry of a product (8 n^2 <= 72
// entries of 8 n multiply-adds each).  These cones are a few hundred doubles: nothing here is priced
// against a roofline.
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

struct OctGroup {
  int n, m, count;
  const double* A;  // count x m x 8 x n x n
  const double* C;  // count x 8 x n x n
  double* W;        // count x 8 x n x n
  double* S;        // count x 8 x n x n: minus_s between PrepareStep and TakeStep
  const int* ids;
};

// sign table M of jordan_matrix_algebra.cc:104-111; the plane of e_i e_j is i ^ j (:113-120)
__device__ __forceinline__ bool OctPositive(int i, int j) {
  // bit (8 i + j) set: +1
  constexpr unsigned long long kPlus =
      (0xFFull << 0) | (0x69ull << 8) | (0xC3ull << 16) | (0xA5ull << 24) | (0x0Full << 32) | (0x55ull << 40) |
      (0x99ull << 48) | (0x33ull << 56);
  return (kPlus >> (8 * i + j)) & 1ull;
}

constexpr int kOctMax = 72;  // 8 planes x 3 x 3

// out = x y   (Multiply :101-138: plane i ^ j takes +- X_i Y_j, i ascending)
__device__ inline void OctMul(int n, const double* x, const double* y, double* out) {
  const int nn = n * n, tot = 8 * nn;
  for (int e = threadIdx.x; e < tot; e += 64) {
    const int p = e / nn, rc = e - p * nn, r = rc % n, c = rc / n;
    double z = 0.0;
    for (int i = 0; i < 8; i++) {
      const int j = i ^ p;
      double t = 0.0;
      for (int k = 0; k < n; k++) t += x[i * nn + k * n + r] * y[j * nn + c * n + k];
      z = OctPositive(i, j) ? z + t : z - t;
    }
    out[e] = z;
  }
  WaveSync();
}
// out = x o y = (x y + y x) / 2   (:163-169); a, b: scratch
__device__ inline void OctJordan(int n, const double* x, const double* y, double* out, double* a, double* b) {
  OctMul(n, x, y, a);
  OctMul(n, y, x, b);
  for (int e = threadIdx.x; e < 8 * n * n; e += 64) out[e] = (a[e] + b[e]) * .5;
  WaveSync();
}
// out = Q(x) y = 2 x o (x o y) - (x o x) o y   (:171-177); t1 .. t3, a, b: scratch
__device__ inline void OctQuadRep(int n, const double* x, const double* y, double* out, double* t1, double* t2,
                                  double* t3, double* a, double* b) {
  OctJordan(n, x, y, t1, a, b);
  OctJordan(n, x, t1, t2, a, b);
  OctJordan(n, x, x, t1, a, b);
  OctJordan(n, t1, y, t3, a, b);
  for (int e = threadIdx.x; e < 8 * n * n; e += 64) out[e] = t2[e] * 2 + t3[e] * -1;
  WaveSync();
}
// <x, y> (:204-210): per plane the column sums, added up; one thread
__device__ inline double OctIp(int n, const double* x, const double* y) {
  const int nn = n * n;
  double ip = 0.0;
  for (int p = 0; p < 8; p++) {
    double tot = 0.0;
    for (int j = 0; j < n; j++) {
      double cs = 0.0;
      for (int i = 0; i < n; i++) cs += x[p * nn + j * n + i] * y[p * nn + j * n + i];
      tot += cs;
    }
    ip += tot;
  }
  return ip;
}

// SetIdentity (hermitian_psd.h:54-56): T::Identity(rank)
__global__ void oct_set_identity(OctGroup g) {
  const int nn = g.n * g.n, sz = 8 * nn;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)g.count * sz; q += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(q % sz);
    g.W[q] = (e < nn && e % g.n == e / g.n) ? 1.0 : 0.0;
  }
}

// ConstructSchurComplementSystem(HermitianPsdConstraint<Octonions>*), initialize = true
__global__ void __launch_bounds__(64) oct_schur(OctGroup g, Arena ar) {
  __shared__ double sW[kOctMax], sA[kOctMax], sQ[kOctMax], t1[kOctMax], t2[kOctMax], t3[kOctMax], ta[kOctMax],
      tb[kOctMax];
  const int n = g.n, m = g.m, sz = 8 * n * n, mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * m * sz;
  const double* C = g.C + (size_t)mem * sz;
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  for (int e = threadIdx.x; e < sz; e += 64) sW[e] = g.W[(size_t)mem * sz + e];
  WaveSync();
  for (int i = 0; i < m; i++) {
    for (int e = threadIdx.x; e < sz; e += 64) sA[e] = A[(size_t)i * sz + e];
    WaveSync();
    OctQuadRep(n, sW, sA, sQ, t1, t2, t3, ta, tb);  // W A_i W
    for (int j = i + (int)threadIdx.x; j < m; j += 64) {
      const double v = OctIp(n, A + (size_t)j * sz, sQ);
      G[j + (size_t)i * m] = v;
      G[i + (size_t)j * m] = v;
    }
    if (threadIdx.x == 0) AW[i] = OctIp(n, sA, sW);
    if (threadIdx.x == 1) AQc[i] = OctIp(n, C, sQ);
    WaveSync();
  }
  for (int e = threadIdx.x; e < sz; e += 64) sA[e] = C[e];
  WaveSync();
  OctQuadRep(n, sW, sA, sQ, t1, t2, t3, ta, tb);  // W C W
  if (threadIdx.x == 0) {
    ar.sc[2 * id] = OctIp(n, sA, sW);
    ar.sc[2 * id + 1] = OctIp(n, sA, sQ);
  }
}

// MODE 0: PrepareStep (:129-145); MODE 1: GetWeightedSlackEigenvalues (:147-168)
template <int MODE>
__global__ void __launch_bounds__(64) oct_prepare(OctGroup g, StepArgs sa) {
  sa.c_weight = CWeightOf(sa);  // (the barrier parameter may live on the device: cxk_select_mu_async)
  extern __shared__ double sy[];  // m
  __shared__ double sW[kOctMax], sS[kOctMax], sQ[kOctMax], t1[kOctMax], t2[kOctMax], t3[kOctMax], ta[kOctMax],
      tb[kOctMax];
  const int n = g.n, m = g.m, sz = 8 * n * n, mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * m * sz;
  const double* C = g.C + (size_t)mem * sz;
  for (int q = threadIdx.x; q < m; q += 64) sy[q] = sa.y[sa.cl_perm[sa.cl_ptr[id] + q]];
  for (int e = threadIdx.x; e < sz; e += 64) sW[e] = g.W[(size_t)mem * sz + e];
  WaveSync();
  for (int e = threadIdx.x; e < sz; e += 64) {  // ComputeNegativeSlack hermitian_psd.h:110-115
    double s = C[e] * -sa.c_weight;
    for (int i = 0; i < m; i++) s = s + A[(size_t)i * sz + e] * sy[i];
    sS[e] = s;
    if (MODE == 0) g.S[(size_t)mem * sz + e] = s;
  }
  WaveSync();
  OctQuadRep(n, sW, sS, sQ, t1, t2, t3, ta, tb);
  if (threadIdx.x != 0) return;
  const double tws = OctIp(n, sW, sS), nq = OctIp(n, sQ, sS);
  if (MODE == 0) {
    sa.info[2 * id] = nq + 2 * tws + n;
    sa.info[2 * id + 1] = 1.0 / 3.0 * (tws + n);  // "TODO: replace this heuristic approximation"
  } else {
    const double lmax = fabs(nq) / (1e-15 + fabs(tws));  // from |x|_1 |x|_inf >= |x|_2^2
    sa.info[4 * id] = lmax * .01;
    sa.info[4 * id + 1] = lmax;
    sa.info[4 * id + 2] = nq;
    sa.info[4 * id + 3] = -tws;
  }
}

// TakeStep (:116-127) with GeodesicUpdateScaled: W <- herm(c^2 W + 2 c k Q(W) s + k^2 Q(W) (Q(s) W)), c = 1.5, k = 0.5
__global__ void __launch_bounds__(64) oct_take_step(OctGroup g, StepArgs sa) {
  if (StepSkipped(sa)) return;  // (enqueued before the host saw the factorization fail: leave W alone)
  __shared__ double sW[kOctMax], sS[kOctMax], q1[kOctMax], q2[kOctMax], q3[kOctMax], t1[kOctMax], t2[kOctMax],
      t3[kOctMax], ta[kOctMax], tb[kOctMax];
  const int n = g.n, nn = n * n, sz = 8 * nn, mem = blockIdx.x;
  const double step = StepSizeOf(sa);
  for (int e = threadIdx.x; e < sz; e += 64) {
    sW[e] = g.W[(size_t)mem * sz + e];
    const double s = g.S[(size_t)mem * sz + e];
    sS[e] = step != 1.0 ? s * step : s;
  }
  WaveSync();
  OctQuadRep(n, sW, sS, q1, t1, t2, t3, ta, tb);
  OctQuadRep(n, sS, sW, q2, t1, t2, t3, ta, tb);
  OctQuadRep(n, sW, q2, q3, t1, t2, t3, ta, tb);
  const double c = 1.5, k = 1.0 / 2.0;
  for (int e = threadIdx.x; e < sz; e += 64) q1[e] = (sW[e] * (c * c) + q1[e] * (2 * k * c)) + q3[e] * (k * k);
  WaveSync();
  for (int e = threadIdx.x; e < sz; e += 64) {  // MakeHermitian: (x + x^*) / 2, x^* = transpose, planes 1 .. 7 negated
    const int p = e / nn, rc = e - p * nn, r = rc % n, cc = rc / n;
    const double t = q1[p * nn + r * n + cc];
    g.W[(size_t)mem * sz + e] = (q1[e] + (p == 0 ? t : -t)) * .5;
  }
}

}  // namespace cxk
