// Lorentz cone with an inner-product matrix Q on its vector part, { (x0, x1) : x0 >= sqrt(x1' Q x1) }:
// the reference's QuadraticConstraint (quadratic_cone_constraint.h:11-86, .cc:14-297; Q absent = I,
// its two-argument constructor).  Jordan algebra of the spin factor with <x, y> = x0 y0 + x1' Q y1:
//   Q(x) y = 2 <x, y> x - det(x) R y,  det x = x0^2 - x1' Q x1,  R = diag(1, -I)   (:46-61)
//   exp / sqrt through the two eigenvalues x0 +- |x1|_Q                              (:63-79)
// Data per cone: A ((n + 1) x m, row 0 = A0, the rest A1), c = (C0, C1), Q (n x n or none), and
// A_gram = A1' Q A1 (m x m, made once by the host: QuadraticConstraintBase::Initialize :216-219).
// State: W = (W0, W1); between PrepareStep and TakeStep D = (d0, d1) and S = (w^{1/2}_1, |w^{1/2}_1|_Q^2),
// the scalar part of w^{1/2} sits in W0 itself -- the reference binds `wsqrt_q0` as a REFERENCE to
// *W0 (:181), so PrepareStep overwrites W0 and TakeStep reads it from there; reproduced.
// One 64-thread workgroup per cone; the maps run on one thread in the order the oracle restates
// (these cones are a handful of doubles each: an epigraph of a stage cost in solver_failures.cc).
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

struct QuadGroup {
  int n, m, count;
  const double* A;      // count x (n + 1) x m
  const double* c;      // count x (n + 1)
  const double* Q;      // count x n x n, nullptr: identity
  const double* Agram;  // count x m x m
  double* W;            // count x (n + 1)
  double* D;            // count x (n + 1)
  double* S;            // count x (n + 1): wsqrt_q1 (n), wsqrt_q1_norm_sqr
  const int* ids;
};

// out = Q x (x when Q is the identity); single thread
__device__ inline void QuadApplyQ(int n, const double* Q, const double* x, double* out) {
  if (!Q) {
    for (int i = 0; i < n; i++) out[i] = x[i];
    return;
  }
  for (int i = 0; i < n; i++) out[i] = 0;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) out[i] += Q[(size_t)j * n + i] * x[j];
}
__device__ inline double QuadDot(int n, const double* x, const double* y) {
  double s = 0;
  for (int i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}
__device__ inline double QuadIp(int n, const double* Q, const double* x, const double* y, double* tmp) {
  QuadApplyQ(n, Q, y, tmp);
  return QuadDot(n, x, tmp);
}
__device__ inline void QuadRep(int n, double x1_norm_sq, double ip, double x0, const double* x1, double y0,
                               const double* y1, double* z0, double* z1) {
  const double det_x = x0 * x0 - x1_norm_sq;
  const double scale = 2 * (x0 * y0 + ip);
  *z0 = scale * x0 - det_x * y0;
  for (int i = 0; i < n; i++) z1[i] = scale * x1[i] + det_x * y1[i];
}

// ConstructSchurComplementSystem(QuadraticConstraintBase*) :240-290 with SchurComplement :88-100
__global__ void __launch_bounds__(64) quad_schur(QuadGroup g, Arena ar) {
  extern __shared__ double lds[];
  const int n = g.n, m = g.m, len = n + 1, mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * len * m;
  const double* c = g.c + (size_t)mem * len;
  const double* Q = g.Q ? g.Q + (size_t)mem * n * n : nullptr;
  const double* Agram = g.Agram + (size_t)mem * m * m;
  const double* W = g.W + (size_t)mem * len;
  double* QW1 = lds;        // n
  double* QC1 = QW1 + n;    // n
  double* v = QC1 + n;      // m
  double* sc = v + m;       // det_w, scale, c_dot_x
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  const double W0 = W[0], C0 = c[0];
  if (threadIdx.x == 0) {
    QuadApplyQ(n, Q, W + 1, QW1);
    QuadApplyQ(n, Q, c + 1, QC1);
    sc[0] = W0 * W0 - QuadDot(n, W + 1, QW1);
    sc[1] = QuadDot(n, QW1, c + 1) + C0 * W0;
    sc[2] = QuadDot(n, c + 1, QW1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += blockDim.x)
    v[i] = QuadDot(n, A + (size_t)i * len + 1, QW1) + A[(size_t)i * len] * W0;
  __syncthreads();
  const double det_w = sc[0], scale = sc[1];
  for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
    const int i = idx % m, j = idx / m;
    double t = (A[(size_t)i * len] * A[(size_t)j * len] - Agram[idx]) * -det_w;
    t += v[i] * v[j];
    t += v[i] * v[j];
    G[idx] = t * 2;
  }
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    double q = det_w * (QuadDot(n, A + (size_t)i * len + 1, QC1) - A[(size_t)i * len] * C0);
    q += 2 * v[i] * scale;
    AW[i] = v[i] * 2;
    AQc[i] = q * 2;
  }
  if (threadIdx.x == 0) {
    double cq = det_w * (QuadDot(n, c + 1, QC1) - C0 * C0);
    cq += 2 * (sc[2] + C0 * W0) * scale;
    ar.sc[2 * id] = scale * 2;
    ar.sc[2 * id + 1] = cq * 2;
  }
}

// MODE 0: PrepareStep :176-214; MODE 1: GetWeightedSlackEigenvalues :142-174
template <int MODE>
__global__ void __launch_bounds__(64) quad_prepare(QuadGroup g, StepArgs sa) {
  sa.c_weight = CWeightOf(sa);  // (the barrier parameter may live on the device: cxk_select_mu_async)
  extern __shared__ double lds[];
  const int n = g.n, m = g.m, len = n + 1, mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * len * m;
  const double* c = g.c + (size_t)mem * len;
  const double* Q = g.Q ? g.Q + (size_t)mem * n * n : nullptr;
  double* W = g.W + (size_t)mem * len;
  double* D = g.D + (size_t)mem * len;
  double* S = g.S + (size_t)mem * len;
  double* sy = lds;         // m
  double* ms = sy + m;      // len: minus_s
  double* wsq = ms + len;   // n
  double* tmp = wsq + n;    // n
  double* d1 = tmp + n;     // n
  for (int q = threadIdx.x; q < m; q += blockDim.x) sy[q] = sa.y[sa.cl_perm[sa.cl_ptr[id] + q]];
  __syncthreads();
  for (int k = threadIdx.x; k < len; k += blockDim.x) {  // ComputeNegativeSlack :131-139
    double s = 0;
    for (int j = 0; j < m; j++) s += A[k + (size_t)j * len] * sy[j];
    ms[k] = s - c[k] * sa.c_weight;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double w0 = W[0];
  for (int i = 0; i < n; i++) wsq[i] = W[1 + i];
  {  // Sqrt :72-79 at k = |w1|_Q
    const double k = sqrt(fabs(QuadIp(n, Q, wsq, wsq, tmp)));
    if (k > 0) {
      const double f = .5 * (sqrt(fabs(w0 + k)) - sqrt(fabs(w0 - k))) / k;
      for (int i = 0; i < n; i++) wsq[i] *= f;
    }
    w0 = .5 * (sqrt(fabs(w0 + k)) + sqrt(fabs(w0 - k)));
  }
  const double nsq = QuadIp(n, Q, wsq, wsq, tmp);
  const double ip = QuadIp(n, Q, wsq, ms + 1, tmp);
  double d0;
  QuadRep(n, nsq, ip, w0, wsq, ms[0], ms + 1, &d0, d1);
  if (MODE == 0) {
    // (`wsqrt_q0` is *W0 itself; not behind a failed factorization: the reference returns before
    // PrepareStep then, cone_program.cc:360-371)
    if (!StepSkipped(sa)) W[0] = w0;
    d0 += 1;
    D[0] = d0;
    for (int i = 0; i < n; i++) {
      D[1 + i] = d1[i];
      S[i] = wsq[i];
    }
    S[n] = nsq;
    const double nd = sqrt(fabs(QuadIp(n, Q, d1, d1, tmp)));
    const double e0 = d0 + nd, e1 = d0 - nd;
    sa.info[2 * id] = e0 * e0 + e1 * e1;
    sa.info[2 * id + 1] = fabs(e0) < fabs(e1) ? fabs(e1) : fabs(e0);
  } else {
    const double nq = sqrt(fabs(QuadIp(n, Q, d1, d1, tmp)));
    const double e0 = d0 + nq, e1 = d0 - nq;
    const double lmax = -fmin(e0, e1), lmin = -fmax(e0, e1);
    sa.info[4 * id] = lmin;
    sa.info[4 * id + 1] = lmax;
    sa.info[4 * id + 2] = lmax * lmax + lmin * lmin;
    sa.info[4 * id + 3] = lmax + lmin;
  }
}

// TakeStep :221-243
__global__ void __launch_bounds__(64) quad_take_step(QuadGroup g, StepArgs sa) {
  if (StepSkipped(sa)) return;  // (enqueued before the host saw the factorization fail: leave W alone)
  extern __shared__ double lds[];
  const int n = g.n, len = n + 1, mem = blockIdx.x;
  const double* Q = g.Q ? g.Q + (size_t)mem * n * n : nullptr;
  double* W = g.W + (size_t)mem * len;
  double* D = g.D + (size_t)mem * len;
  const double* S = g.S + (size_t)mem * len;
  double* d1 = lds;       // n
  double* tmp = d1 + n;   // n
  double* w1 = tmp + n;   // n
  if (threadIdx.x != 0) return;
  double d0 = D[0];
  for (int i = 0; i < n; i++) d1[i] = D[1 + i];
  const double step = StepSizeOf(sa);
  if (step != 1.0) {
    d0 = step * d0;
    for (int i = 0; i < n; i++) d1[i] = step * d1[i];
  }
  {  // Exp :63-70 at k = |d1|_Q
    const double k = sqrt(fabs(QuadIp(n, Q, d1, d1, tmp)));
    if (k > 0) {
      const double f = .5 * (exp(d0 + k) - exp(d0 - k)) / k;
      for (int i = 0; i < n; i++) d1[i] *= f;
    }
    d0 = .5 * (exp(d0 + k) + exp(d0 - k));
  }
  D[0] = d0;  // (the reference exponentiates its d in place)
  for (int i = 0; i < n; i++) D[1 + i] = d1[i];
  const double ip = QuadIp(n, Q, S, d1, tmp);
  double w0;
  QuadRep(n, S[n], ip, W[0], S, d0, d1, &w0, w1);
  W[0] = w0;
  for (int i = 0; i < n; i++) W[1 + i] = w1[i];
}

}  // namespace cxk
