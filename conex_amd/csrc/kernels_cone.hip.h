// Linear (nonnegative orthant) and second-order (spin factor) cone kernels, plus the
// constant Schur block.  One workgroup per constraint.
//
// Reference semantics:
//   ConstructSchurComplementSystem(LinearConstraint*)  linear_constraint.cc:177-205
//   PrepareStep/TakeStep/GetWeightedSlackEigenvalues   linear_constraint.cc:108-175
//   ConstructSchurComplementSystem(SOCConstraint*)     soc_constraint.cc:272-303
//   Sqrt/Exp/QuadraticRepresentation/NormInf           soc_constraint.cc:14-191
//   PrepareStep/TakeStep/GetWeightedSlackEigenvalues   soc_constraint.cc:200-270
//   QuadraticFunction / SupernodalAssemblerStatic      quadratic_cost.cc:46-57
#pragma once
#include "device_utils.h"
#include "kernels_lmi.hip.h"
#include "mu_rule.h"

namespace cxk {

struct VecGroup {  // linear: len = rows ; SOC: len = n + 1
  int len;
  int m;
  int count;
  const double* A;  // count x (len x m), col-major
  const double* c;  // count x len
  double* W;        // count x len   (SOC: W[0] = W0, W[1:] = W1)
  double* T1;       // count x len   linear temp_1 / SOC d (d0, d1)
  double* T2;       // count x len   linear temp_2 (d)
  const int* ids;
};

// ------------------------------------------------------------------ linear
__global__ void __launch_bounds__(256) linear_schur(VecGroup g, Arena ar) {
  __shared__ double red[8];
  const int r = g.len, m = g.m;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * r * m;
  const double* c = g.c + (size_t)mem * r;
  const double* w = g.W + (size_t)mem * r;
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
    const int i = idx % m, j = idx / m;
    double s = 0;
    for (int k = 0; k < r; k++) {
      const double wk = w[k];
      s = fma(wk * A[k + (size_t)i * r], wk * A[k + (size_t)j * r], s);
    }
    G[idx] = s;
  }
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    double a = 0, q = 0;
    for (int k = 0; k < r; k++) {
      const double wk = w[k];
      a = fma(A[k + (size_t)i * r], wk, a);
      q = fma(wk * A[k + (size_t)i * r], wk * c[k], q);
    }
    AW[i] = a;
    AQc[i] = q;
  }
  double s1 = 0, s2 = 0;
  for (int k = threadIdx.x; k < r; k += blockDim.x) {
    const double wc = w[k] * c[k];
    s1 += wc;
    s2 = fma(wc, wc, s2);
  }
  s1 = BlockSum(s1, red);
  s2 = BlockSum(s2, red);
  if (threadIdx.x == 0) {
    ar.sc[2 * id] = s1;
    ar.sc[2 * id + 1] = s2;
  }
}

// PerformLineSearch(LinearConstraint*) + FindMinimumMu (linear_constraint.cc:48-103):
//   d0 = e + w o (A y0 - c k0),  delta = (e + w o (A y1 - c k1)) - d0,
//   per row the interval of t with |d0 + t delta| <= dinfmax; out[2 id] = max lower end,
//   out[2 id + 1] = min upper end (the caller treats lower > upper as failure).
struct LineSearchArgs {
  const double* y0;     // permuted solve of -2 AW
  const double* y1;     // permuted solve of AQc c_s + b b_s - 2 AW
  const int* cl_ptr;
  const int* cl_perm;
  double c0_weight, c1_weight, dinfmax;
  double* out;          // [2 K]
};

__global__ void __launch_bounds__(256) linear_line_search(VecGroup g, LineSearchArgs a) {
  extern __shared__ double lds[];
  __shared__ double red[8];
  const int r = g.len, m = g.m;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * r * m;
  const double* c = g.c + (size_t)mem * r;
  const double* w = g.W + (size_t)mem * r;
  double* s0 = lds;
  double* s1 = lds + m;
  for (int q = threadIdx.x; q < m; q += blockDim.x) {
    const int v = a.cl_perm[a.cl_ptr[id] + q];
    s0[q] = a.y0[v];
    s1[q] = a.y1[v];
  }
  __syncthreads();
  double ub = 1.7976931348623157e308, lb = -1.7976931348623157e308;
  for (int k = threadIdx.x; k < r; k += blockDim.x) {
    double t0 = 0, t1 = 0;
    for (int j = 0; j < m; j++) {
      t0 = fma(A[k + (size_t)j * r], s0[j], t0);
      t1 = fma(A[k + (size_t)j * r], s1[j], t1);
    }
    t0 -= c[k] * a.c0_weight;
    t1 -= c[k] * a.c1_weight;
    const double d0 = t0 * w[k] + 1, d1 = t1 * w[k] + 1;
    const double delta = d1 - d0;
    double ubi = (a.dinfmax - d0) / delta, lbi = (-a.dinfmax - d0) / delta;
    if (lbi > ubi) {
      const double t = ubi;
      ubi = lbi;
      lbi = t;
    }
    ub = fmin(ub, ubi);
    lb = fmax(lb, lbi);
  }
  lb = WaveMax(lb);
  ub = -WaveMax(-ub);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave] = lb;
    red[4 + wave] = ub;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    for (int q = 1; q < nw; q++) {
      lb = fmax(lb, red[q]);
      ub = fmin(ub, red[4 + q]);
    }
    a.out[2 * id] = lb;
    a.out[2 * id + 1] = ub;
  }
}

// mode 0 PrepareStep, mode 1 GetWeightedSlackEigenvalues
template <int MODE>
__global__ void __launch_bounds__(256) linear_prepare(VecGroup g, StepArgs sa) {
  sa.c_weight = CWeightOf(sa);  // (the barrier parameter may live on the device: cxk_select_mu_async)
  extern __shared__ double lds[];
  __shared__ double red[8];
  const int r = g.len, m = g.m;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * r * m;
  const double* c = g.c + (size_t)mem * r;
  double* w = g.W + (size_t)mem * r;
  double* T1 = g.T1 + (size_t)mem * r;
  double* T2 = g.T2 + (size_t)mem * r;
  double* sy = lds;
  for (int q = threadIdx.x; q < m; q += blockDim.x) sy[q] = sa.y[sa.cl_perm[sa.cl_ptr[id] + q]];
  __syncthreads();
  const double kc = (MODE == 0 && sa.affine) ? 0.0 : sa.c_weight;
  double mx = 0, mn = 0, s2 = 0, s1 = 0;
  bool first = true;
  for (int k = threadIdx.x; k < r; k += blockDim.x) {
    double s = 0;
    for (int j = 0; j < m; j++) s = fma(A[k + (size_t)j * r], sy[j], s);
    s -= c[k] * kc;
    if (MODE == 0) {
      if (sa.affine) {  // AffineUpdate: SW = minus_s .* W ; W += W .* SW
        const double sw = s * w[k];
        T1[k] = sw;
        w[k] += w[k] * sw;
      } else {
        const double d = s * w[k] + sa.e_weight;
        T2[k] = d;
        mx = fmax(mx, fabs(d));
        s2 = fma(d, d, s2);
      }
    } else {
      const double ws = w[k] * s;
      mx = first ? ws : fmax(mx, ws);
      mn = first ? ws : fmin(mn, ws);
      first = false;
      s2 = fma(ws, ws, s2);
      s1 += ws;
    }
  }
  if (MODE == 0 && sa.affine) return;
  if (MODE == 1) {  // lanes without rows must not disturb min/max
    if (first) {
      mx = -1.7976931348623157e308;
      mn = 1.7976931348623157e308;
    }
    mn = -WaveMax(-mn);
  }
  mx = WaveMax(mx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    red[wave] = mx;
    red[4 + wave] = mn;
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  double gmx = red[0], gmn = red[4];
  for (int q = 1; q < nw; q++) {
    gmx = fmax(gmx, red[q]);
    gmn = fmin(gmn, red[4 + q]);
  }
  __syncthreads();
  s2 = BlockSum(s2, red);
  s1 = BlockSum(s1, red);
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      sa.info[2 * id] = s2;
      sa.info[2 * id + 1] = gmx;
    } else {
      sa.info[4 * id] = -gmx;
      sa.info[4 * id + 1] = -gmn;
      sa.info[4 * id + 2] = s2;
      sa.info[4 * id + 3] = -s1;
    }
  }
}

__global__ void linear_take_step(VecGroup g, StepArgs sa) {
  if (StepSkipped(sa)) return;  // (enqueued before the host saw the factorization fail: leave W alone)
  const size_t total = (size_t)g.count * g.len;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < total;
       q += (size_t)gridDim.x * blockDim.x) {
    double d = g.T2[q];
    const double step = StepSizeOf(sa);
    if (step != 1) d *= step;
    g.T2[q] = d;
    g.W[q] *= exp(d);
  }
}

__global__ void vec_set_identity(VecGroup g, int soc) {
  const size_t total = (size_t)g.count * g.len;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < total;
       q += (size_t)gridDim.x * blockDim.x)
    g.W[q] = soc ? ((q % g.len) == 0 ? 1.0 : 0.0) : 1.0;
}

// -------------------------------------------------------------------- SOC
// z = f(x) through the spin-factor spectral decomposition; op 0 sqrt, 1 exp. Single thread.
__device__ inline void SocSpectral(int n, double x0, const double* x1, int op, double* z) {
  double nq = 0;
  for (int i = 0; i < n; i++) nq = fma(x1[i], x1[i], nq);
  nq = sqrt(nq);
  const double e0 = x0 + nq, e1 = x0 - nq;
  const double f0 = op == 0 ? sqrt(e0) : exp(e0);
  const double f1 = op == 0 ? sqrt(e1) : exp(e1);
  z[0] = f0 * .5 + f1 * .5;
  for (int i = 0; i < n; i++) {
    const double q = nq > 0 ? x1[i] / nq : 0.0;
    z[1 + i] = nq > 0 ? f0 * (.5 * q) + f1 * (-.5 * q) : 0.0;
  }
}

// out = Q(x) y = 2 (x.y) x - det(x) R y
__device__ inline void SocQuadRep(int len, const double* x, const double* y, double* out) {
  double t2 = 0, xy = 0;
  for (int i = 1; i < len; i++) t2 = fma(x[i], x[i], t2);
  for (int i = 0; i < len; i++) xy = fma(x[i], y[i], xy);
  const double det = x[0] * x[0] - t2;
  for (int i = 0; i < len; i++) out[i] = (2 * xy) * x[i] + (i == 0 ? -det * y[i] : det * y[i]);
}

// The two maps above dealt to the lanes of one wavefront: the reductions stay one lane's fma chain
// (their order is the single-thread order: same bits), the per-element work -- above all the n
// divisions of the spectral map -- runs one element per lane.  A SIMD issues a vector instruction
// in the same 4-8 cycles whether one lane or all are active, and with five single-lane cones
// resident per SIMD the serial forms cost 16 us per launch of 5000 cones.  x / y may be global or
// LDS; z / out LDS (read back by other lanes: the caller synchronises the wavefront afterwards).
__device__ __forceinline__ void SocSpectralWave(int n, double x0, const double* x1, int op, double* z, int lane) {
  double nq = 0;
  for (int i = 0; i < n; i++) nq = fma(x1[i], x1[i], nq);  // every lane the same chain: no broadcast needed
  nq = sqrt(nq);
  const double e0 = x0 + nq, e1 = x0 - nq;
  const double f0 = op == 0 ? sqrt(e0) : exp(e0);
  const double f1 = op == 0 ? sqrt(e1) : exp(e1);
  if (lane == 0) z[0] = f0 * .5 + f1 * .5;
  for (int i = lane; i < n; i += 64) {
    const double q = nq > 0 ? x1[i] / nq : 0.0;
    z[1 + i] = nq > 0 ? f0 * (.5 * q) + f1 * (-.5 * q) : 0.0;
  }
}
__device__ __forceinline__ void SocQuadRepWave(int len, const double* x, const double* y, double* out, int lane) {
  double t2 = 0, xy = 0;
  for (int i = 1; i < len; i++) t2 = fma(x[i], x[i], t2);
  for (int i = 0; i < len; i++) xy = fma(x[i], y[i], xy);
  const double det = x[0] * x[0] - t2;
  for (int i = lane; i < len; i += 64) out[i] = (2 * xy) * x[i] + (i == 0 ? -det * y[i] : det * y[i]);
}

// One wavefront per cone, four cones per workgroup.
// STAGED: the cone's data (A, c, W) is copied to LDS in one round trip first; large cones whose
// staged image would not fit read their operands where they are.
template <bool STAGED>
__global__ void __launch_bounds__(256) soc_schur(VecGroup g, Arena ar) {
  extern __shared__ double lds_all[];
  const int len = g.len, n = len - 1, m = g.m;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int mem = blockIdx.x * nw + wave;
  if (mem >= g.count) return;  // wave-uniform; only wavefront-level synchronisation below
  double* lds = lds_all + (size_t)wave * (len * (STAGED ? 2 * m + 4 : m + 2));
  const int id = g.ids[mem];
  const double* A = g.A + (size_t)mem * len * m;
  const double* c = g.c + (size_t)mem * len;
  const double* W = g.W + (size_t)mem * len;
  double* wsqrt = lds;            // len
  double* wc = wsqrt + len;       // len
  double* WA = wc + len;          // len x m
  const double *sA = A, *sW = W, *sC = c;
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  if constexpr (STAGED) {  // (otherwise the phases below fetch their operands themselves: four dependent round trips)
    double* tA = WA + len * m;  // len x m
    double* tW = tA + len * m;  // len
    double* tC = tW + len;      // len
    for (int q = lane; q < len * m; q += 64) tA[q] = A[q];
    for (int q = lane; q < len; q += 64) {
      tW[q] = W[q];
      tC[q] = c[q];
    }
    sA = tA;
    sW = tW;
    sC = tC;
    WaveSync();
  }
  SocSpectralWave(n, sW[0], sW + 1, 0, wsqrt, lane);
  WaveSync();
  SocQuadRepWave(len, wsqrt, sC, wc, lane);
  WaveSync();
  for (int i = lane; i < m; i += 64) SocQuadRep(len, wsqrt, sA + (size_t)i * len, WA + i * len);
  WaveSync();
  for (int idx = lane; idx < m * m; idx += 64) {
    const int i = idx % m, j = idx / m;
    double s = 0;
    for (int k = 0; k < len; k++) s = fma(WA[k + i * len], WA[k + j * len], s);
    G[idx] = 2 * s;
  }
  for (int i = lane; i < m; i += 64) {
    double a = 0, q = 0;
    for (int k = 0; k < len; k++) {
      a = fma(sA[k + (size_t)i * len], sW[k], a);
      q = fma(WA[k + i * len], wc[k], q);
    }
    AW[i] = 2 * a;
    AQc[i] = 2 * q;
  }
  if (lane == 0) {
    double s = 0;
    for (int k = 0; k < len; k++) s = fma(wc[k], wc[k], s);
    ar.sc[2 * id] = 2 * wc[0];
    ar.sc[2 * id + 1] = 2 * s;
  }
}

template <int MODE>
__global__ void __launch_bounds__(64) soc_prepare(VecGroup g, StepArgs sa) {
  sa.c_weight = CWeightOf(sa);  // (the barrier parameter may live on the device: cxk_select_mu_async)
  extern __shared__ double lds[];
  const int len = g.len, n = len - 1, m = g.m;
  const int mem = blockIdx.x, id = g.ids[mem];
  const double* A = g.A + (size_t)mem * len * m;
  const double* c = g.c + (size_t)mem * len;
  double* W = g.W + (size_t)mem * len;
  double* D = g.T1 + (size_t)mem * len;
  double* sy = lds;          // m
  double* ms = sy + m;       // len
  double* wsqrt = ms + len;  // len
  double* d = wsqrt + len;   // len
  for (int q = threadIdx.x; q < m; q += blockDim.x) sy[q] = sa.y[sa.cl_perm[sa.cl_ptr[id] + q]];
  __syncthreads();
  for (int k = threadIdx.x; k < len; k += blockDim.x) {
    double s = 0;
    for (int j = 0; j < m; j++) s = fma(A[k + (size_t)j * len], sy[j], s);
    ms[k] = s - c[k] * sa.c_weight;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    SocSpectral(n, W[0], W + 1, 0, wsqrt);
    SocQuadRep(len, wsqrt, ms, d);
    double nq = 0;
    if (MODE == 0) {
      // PrepareStep leaves w^{1/2} in W (soc_constraint.cc:259-261) -- unless it was enqueued behind a
      // factorization that turns out to have failed: the reference returns before PrepareStep then,
      // with W untouched (cone_program.cc:360-371)
      if (!StepSkipped(sa))
        for (int k = 0; k < len; k++) W[k] = wsqrt[k];
      d[0] += 1;
      double s = 0;
      for (int k = 0; k < len; k++) {
        D[k] = d[k];
        s = fma(d[k], d[k], s);
      }
      for (int k = 1; k < len; k++) nq = fma(d[k], d[k], nq);
      nq = sqrt(nq);
      const double e0 = fabs(d[0] + nq), e1 = fabs(d[0] - nq);
      sa.info[2 * id] = 2 * s;
      sa.info[2 * id + 1] = e0 > e1 ? e0 : e1;
    } else {
      for (int k = 1; k < len; k++) nq = fma(d[k], d[k], nq);
      nq = sqrt(nq);
      const double e0 = d[0] + nq, e1 = d[0] - nq;
      const double lmax = -fmin(e0, e1), lmin = -fmax(e0, e1);
      sa.info[4 * id] = lmin;
      sa.info[4 * id + 1] = lmax;
      sa.info[4 * id + 2] = lmax * lmax + lmin * lmin;
      sa.info[4 * id + 3] = lmax + lmin;
    }
  }
}

__global__ void __launch_bounds__(64) soc_take_step(VecGroup g, StepArgs sa) {
  if (StepSkipped(sa)) return;  // (enqueued before the host saw the factorization fail: leave W alone)
  extern __shared__ double lds[];
  const int len = g.len, n = len - 1;
  const int mem = blockIdx.x;
  double* W = g.W + (size_t)mem * len;
  double* D = g.T1 + (size_t)mem * len;
  double* d = lds;
  double* ex = d + len;
  double* wn = ex + len;
  double* w = wn + len;
  if (threadIdx.x == 0) {
    for (int k = 0; k < len; k++) {
      d[k] = D[k];
      w[k] = W[k];
    }
    const double step = StepSizeOf(sa);
    if (step != 1.0) {
      for (int k = 0; k < len; k++) d[k] *= step;
      for (int k = 1; k < len; k++) D[k] = d[k];  // the reference scales temp1_1 in place
    }
    SocSpectral(n, d[0], d + 1, 1, ex);
    SocQuadRep(len, w, ex, wn);
    for (int k = 0; k < len; k++) W[k] = wn[k];
  }
}

// ----------------------------------------------------------- constant block
struct StaticGroup {
  int m;
  int count;
  const double* Gc;    // count x (m x m)
  const double* AQc0;  // count x m constant AQc (equality constraints: [0; b], equality_constraint.cc:14-30)
  const int* ids;
};

__global__ void static_schur(StaticGroup g, Arena ar) {
  const int mem = blockIdx.x, id = g.ids[mem], mm = g.m * g.m;
  const double* src = g.Gc + (size_t)mem * mm;
  double* G = ar.G + ar.g_off[id];
  for (int q = threadIdx.x; q < mm; q += blockDim.x) G[q] = src[q];
  for (int q = threadIdx.x; q < g.m; q += blockDim.x) {
    ar.AWc[ar.r_off[id] + q] = 0;
    ar.AQcc[ar.r_off[id] + q] = g.AQc0[(size_t)mem * g.m + q];
  }
  if (threadIdx.x == 0) {
    ar.sc[2 * id] = 0;
    ar.sc[2 * id + 1] = 0;
  }
}

// What the host reads after a round trip, written into its pinned mailbox by the first wavefront
// of a workgroup: the four reduction outputs, the six step scalars, the factorization flag, the
// sequence number the host spins on and a checksum (MailboxWrite).
struct MailboxArgs {
  const double* red;   // [4]
  const double* scal;  // [6]
  const int* fail;     // [2]: flag, tag of a failed first-level pivot of a fused launch
  int tag;
  double seq;
  double* mb;          // pinned host memory, 16 doubles; nullptr: no mailbox write
  const double* mu;    // [1] inv_sqrt_mu as the device last selected it (MuRuleArgs), slot 13; may be null
};
// v: this thread's slot value (threads 0 .. 10 and 13; anything elsewhere).  The twelve values, the sequence
// number the host spins on (slot 11) and a checksum (slot 12) leave in ONE store instruction, no
// fence in between: the host does not rely on the order in which the bytes arrive (they cross PCIe
// as posted writes; with a system-scope fence between data and sequence number -- 2 us per round
// trip -- the sequence number was still seen ahead of the data about once in a thousand round trips
// on a cold box).  It accepts a mailbox only when slot 12 equals the XOR of the sequence number's
// bit pattern and the twelve values', each rotated by its own amount (kMailboxRot: stale slots
// cannot cancel each other the way equal changes of two slots would under a plain XOR).
__host__ __device__ constexpr int MailboxRot(int slot) { return (7 * slot + 1) & 63; }
__device__ __forceinline__ void MailboxWrite(const MailboxArgs& m, double v) {
  const int t = threadIdx.x;
  if (t >= 64) return;
  unsigned long long x = 0ull;
  if (t <= 10 || t == 13) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const int r = MailboxRot(t);
    x = r ? (bits << r) | (bits >> (64 - r)) : bits;
  }
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) x ^= (unsigned long long)__shfl_xor((long long)x, d, 64);
  double out = v;
  if (t == 11) out = m.seq;
  if (t == 12) out = __longlong_as_double((long long)(x ^ (unsigned long long)__double_as_longlong(m.seq)));
  if (t <= 13) __hip_atomic_store(m.mb + t, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// fail[1] == tag: a pivot failed in the first factor level of the latest fused launch
__device__ __forceinline__ double MailboxFailValue(const MailboxArgs& m) {
  return (m.fail[0] != 0 || (m.tag != 0 && m.fail[1] == m.tag)) ? 1.0 : 0.0;
}
__device__ __forceinline__ void MailboxPack(const MailboxArgs& m) {
  const int t = threadIdx.x;
  double v = 0.0;
  if (t < 4) v = m.red[t];
  if (t >= 4 && t < 10) v = m.scal[t - 4];
  if (t == 10) v = MailboxFailValue(m);
  if (t == 13 && m.mu) v = *m.mu;
  MailboxWrite(m, v);
}

// Sharded contexts: the ranks' partial step results meet in ONE sum all-reduce.  Rank r writes its
// (up to four) values into slot r of a world x 4 buffer and zeros the other slots (step_slots_fill);
// after the sum every rank holds every rank's values and combines them in RANK ORDER
// (step_slots_reduce): min / max / sum as the mode asks, the sums in one fixed order on every rank.
// mode 0: {sum normsqrd, max norminfd}; mode 1: {min lambda_min, max lambda_max, sum frob, sum trace}.
__global__ void step_slots_fill(int rank, int world, const double* __restrict__ red, double* __restrict__ slots) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < 4 * world; q += gridDim.x * blockDim.x)
    slots[q] = (q >> 2) == rank ? red[q & 3] : 0.0;
}
__global__ void step_slots_reduce(int mode, int world, const double* __restrict__ slots, double* __restrict__ red) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double v0 = slots[0], v1 = slots[1], v2 = slots[2], v3 = slots[3];
  for (int r = 1; r < world; r++) {
    const double* s = slots + 4 * r;
    if (mode == 0) {
      v0 += s[0];
      v1 = fmax(v1, s[1]);
    } else {
      v0 = fmin(v0, s[0]);
      v1 = fmax(v1, s[1]);
      v2 += s[2];
      v3 += s[3];
    }
  }
  red[0] = v0;
  red[1] = v1;
  if (mode != 0) {
    red[2] = v2;
    red[3] = v3;
  }
}

// The end of the fixed-order reduction below: the 256 threads' partial results -> out[], and with a
// mailbox the results travel to the host from here (no separate mailbox_pack).
// LOCAL (the tail workgroup): the mailbox takes the reduced values from LDS, the step scalars from
// `scal_lds` (this workgroup formed them) or, like the failure flag, from `pre_value` (threads 4 .. 10:
// their slot's value, read before the wait) -- no memory round trip between the last result and the
// host write.
// The barrier-parameter selection evaluated where its inputs appear (mode 1, the tail workgroup):
// out[0] <- NextInvSqrtMu(u, the reduced eigenvalue bounds), mu_rule.h.
struct MuRuleArgs {
  int on;
  cxk_mu::Update u;
  double* out;
};
template <bool LOCAL = false>
__device__ __forceinline__ void ReduceStepFinish(int mode, double a, double b, double c, double d,
                                                 double* __restrict__ out, const MailboxArgs& mbx,
                                                 const double* scal_lds = nullptr, double pre_value = 0.0,
                                                 const MuRuleArgs* rule = nullptr) {
  __shared__ double red[8], red2[8], red3[8], red4[8], res[5];
  // the two BlockSums (wave sum, wave totals added in wave order) and the max / min reductions
  // behind ONE barrier
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  a = WaveSum(a);
  d = WaveSum(d);
  const double bm = (mode == 0) ? WaveMax(b) : -WaveMax(-b);
  const double cm = WaveMax(c);
  if (lane == 0) {
    red[wave] = bm;
    red2[wave] = cm;
    red3[wave] = a;
    red4[wave] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double B = red[0], C = red2[0], A = 0, D = 0;
    for (int w = 1; w < nw; w++) {
      B = (mode == 0) ? fmax(B, red[w]) : fmin(B, red[w]);
      C = fmax(C, red2[w]);
    }
    for (int w = 0; w < nw; w++) {
      A += red3[w];
      D += red4[w];
    }
    if (mode == 0) {
      out[0] = A;
      out[1] = B;
      if (LOCAL) res[0] = A, res[1] = B;
    } else {
      out[0] = B;
      out[1] = C;
      out[2] = A;
      out[3] = D;
      if (LOCAL) res[0] = B, res[1] = C, res[2] = A, res[3] = D;
      if (rule && rule->on) {  // (not LOCAL: the mailbox reads it back from rule->out behind the fence below)
        cxk_mu::Wse e;
        e.lmin = B;
        e.lmax = C;
        e.frob = A;
        e.trace = D;
        const double v = cxk_mu::NextInvSqrtMu(rule->u, e);
        rule->out[0] = v;
        if (LOCAL) res[4] = v;
      }
    }
  }
  if (!mbx.mb) return;
  if (LOCAL) {
    __syncthreads();
    const int t = threadIdx.x;
    double v = 0.0;
    // (the separate launches send out[2], out[3] as the last mode-1 call left them: nobody reads them in mode 0)
    if (t < (mode == 0 ? 2 : 4)) v = res[t];
    if (t >= 4 && t < 10) v = scal_lds ? scal_lds[t - 4] : pre_value;
    if (t == 10) v = pre_value;
    if (t == 13) v = (mode != 0 && rule && rule->on) ? res[4] : pre_value;
    MailboxWrite(mbx, v);
  } else {
    __threadfence();  // out[] is read back below by other threads of this workgroup
    __syncthreads();
    MailboxPack(mbx);
  }
}

// Fixed-order reduction of the per-constraint step outputs on one workgroup.
// mode 0: info2 -> {sum normsqrd, max norminfd (init -1)}
// mode 1: info4 -> {min lambda_min (init 30000), max lambda_max (init -30000), sum frob, sum trace}
__global__ void __launch_bounds__(256)
reduce_step_info(int K, int mode, const double* __restrict__ info, const unsigned char* __restrict__ mask,
                 double* __restrict__ out, MailboxArgs mbx, MuRuleArgs rule) {
  double a = 0, b = (mode == 0) ? -1.0 : 30000.0, c = -30000.0, d = 0;
  // eight strided constraints per trip, loads issued together (a plain loop pays two dependent
  // round trips -- mask, values -- per element); accumulation order unchanged
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < K; i0 += U * blockDim.x) {
    bool on[U];
    double v[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * blockDim.x;
      const int ic = i < K ? i : 0;
      on[u] = mask[ic] != 0 && i < K;
      if (mode == 0) {
        v[u][0] = info[2 * ic];
        v[u][1] = info[2 * ic + 1];
        v[u][2] = v[u][3] = 0.0;
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) v[u][q] = info[4 * ic + q];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!on[u]) continue;
      if (mode == 0) {
        a += v[u][0];
        b = fmax(b, v[u][1]);
      } else {
        b = fmin(b, v[u][0]);
        c = fmax(c, v[u][1]);
        a += v[u][2];
        d += v[u][3];
      }
    }
  }
  ReduceStepFinish(mode, a, b, c, d, out, mbx, nullptr, 0.0, &rule);
}

// ---------------------------------------------------------------------------------------------
// The tail workgroup of a PrepareStep / eigenvalue-query launch (lmi_prepare_rows: one workgroup
// more than the constraints need): what reduce_step_info and step_scalars do in launches of their
// own behind it.  The constraints' wavefronts hand their results over through `slots` (four doubles
// per constraint, data-as-flag: armed with kTailSentinel, written with write-through stores, polled
// here with sc1 loads -- MI355X_MICROARCH.md's hand-off rules, as in tree_fused); this workgroup
// first forms the step scalars (they need y only), then takes the results as they arrive, reduces
// them in reduce_step_info's order (the same bits) and writes the mailbox.
constexpr unsigned long long kTailSentinel = 0x7FF4C0DEC0DE7A11ull;
constexpr int kTailSpin = 1 << 20;  // polls before the workgroup gives up (NaNs go out: the solve fails loudly)
struct StepTail {
  double* slots;  // nullptr: no tail workgroup in this launch
  double* rearm;  // the other slot set: consumed by the launch before, armed again by this one while it waits
  int K, mode;
  const unsigned char* mask;
  double* red_out;
  int scal, N;    // scal != 0: the step scalars as well
  const double *b, *AQc, *y, *sys_sc;
  double* scal_out;
  // ny > 0: the Newton direction is still the three solutions of the triple factorization (StepArgs::y3).  The
  // constraints' wavefronts combine the entries they need where they read them; ny further workgroups behind this
  // one write all of y out (DirectionBlock), count themselves in y_done, and this workgroup forms its scalars when
  // the count has reached y_target -- long before the constraints' results arrive.
  int ny;
  double* y_out;
  unsigned long long* y_done;
  unsigned long long y_target;
  MailboxArgs mbx;
  MuRuleArgs rule;  // mode 1: the selection of inv_sqrt_mu on the device
};
typedef unsigned int TailU4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void AgentStore(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// step_scalars (kernels_kkt.hip.h, 1024 threads) on 256: thread r plays threads r, r + 256, r + 512,
// r + 768 -- the same fma chains, the same wave sums (wave (r >> 6) + 4 q of the 1024), the sixteen
// wave totals added in the same order: the same bits.
__device__ inline void StepScalarsOn256(int N, const double* __restrict__ b, const double* __restrict__ AQc,
                                        const double* __restrict__ y, const double* __restrict__ sys_sc,
                                        double* __restrict__ out, double* lds_out, bool y_fresh = false) {
  __shared__ double red[4][16];
  double s[4][4];
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 4; k++) s[q][k] = 0.0;
  constexpr int U = 4;
  for (int base = 0; base < N; base += 1024 * U) {
    double vb[4][U], vq[4][U], vy[4][U];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int p = base + 1024 * u + 256 * q + (int)threadIdx.x;
        const bool on = p < N;
        vb[q][u] = on ? b[p] : 0.0;
        vq[q][u] = on ? AQc[p] : 0.0;
        // (y_fresh: written by other workgroups of this launch -- past this XCD's L2)
        vy[q][u] = on ? (y_fresh ? __hip_atomic_load(y + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : y[p]) : 0.0;
      }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        s[q][0] = fma(vb[q][u], vy[q][u], s[q][0]);
        s[q][1] = fma(vq[q][u], vy[q][u], s[q][1]);
        s[q][2] = fma(vb[q][u], vb[q][u], s[q][2]);
        s[q][3] = fma(vq[q][u], vq[q][u], s[q][3]);
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double t = WaveSum(s[q][k]);
      if (lane == 0) red[k][wave + 4 * q] = t;
    }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int w = 0; w < 16; w++) t += red[threadIdx.x][w];
    out[threadIdx.x] = lds_out[threadIdx.x] = t;
  }
  if (threadIdx.x == 4) out[4] = lds_out[4] = sys_sc[0];
  if (threadIdx.x == 5) out[5] = lds_out[5] = sys_sc[1];
}

// One of the ny workgroups that write the Newton direction out (newton_from_three riding in the launch that reads
// it): write-through stores, then -- behind their completion -- one count per workgroup.
__device__ inline void DirectionBlock(const StepTail& T, const StepArgs& sa, int b) {
  const int i = b * 256 + (int)threadIdx.x;
  if (i < T.N) AgentStore(T.y_out + i, YFromThree(sa.y3, sa.y3_stride, sa.y3_k[0], i));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(T.y_done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void PrepareTailBlock(const StepTail& T) {
  __shared__ double s_scal[6];
  bool y_ok = true;
  if (T.ny > 0) {  // the direction's workgroups first (a few microseconds; nothing else to do before)
    y_ok = false;
    for (int spin = 0; spin < kTailSpin && !y_ok; spin++) {
      y_ok = __hip_atomic_load(T.y_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= T.y_target;
      if (!y_ok) __builtin_amdgcn_s_sleep(8);
    }
    y_ok = __syncthreads_and(y_ok) != 0;
  }
  if (T.scal) {
    StepScalarsOn256(T.N, T.b, T.AQc, T.y, T.sys_sc, T.scal_out, s_scal, T.ny > 0);
    if (!y_ok) {  // (the wait ran out: NaNs go out, the solve fails loudly)
      __syncthreads();
      if (threadIdx.x < 2) T.scal_out[threadIdx.x] = s_scal[threadIdx.x] = __longlong_as_double(0x7FF8000000000000ll);
    }
  }
  const int K = T.K, mode = T.mode, t = threadIdx.x;
  // what the mailbox carries besides this launch's results: final before the launch
  double pre = 0.0;
  if (T.mbx.mb) {
    if (t >= 4 && t < 10 && !T.scal) pre = T.mbx.scal[t - 4];
    if (t == 10) pre = MailboxFailValue(T.mbx);
    if (t == 13 && T.mbx.mu) pre = *T.mbx.mu;
  }
  // (two slot sets, used in turn: the stores that arm a set again are long complete when the launch
  // that uses it next begins, and none of them sits between this launch's last result and its end)
  const double armed = __longlong_as_double((long long)kTailSentinel);
  for (int i = t; i < 4 * K; i += 256) AgentStore(T.rearm + i, armed);
  double a = 0, b = (mode == 0) ? -1.0 : 30000.0, c = -30000.0, d = 0;
  // A polling round is one or two 16-byte sc1 loads per constraint (8-byte loads of the four values
  // one by one made a round of 1000 constraints cost several microseconds of address processing on
  // this one CU -- more than the launch it replaces); offsets past the buffer return zeros without
  // a memory request.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(T.slots, 0, K * 32, 0x00020000);
  constexpr int U = 4;
  for (int i0 = t; i0 < K; i0 += U * 256) {
    double v[U][4];
    bool on[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * 256;
      on[u] = i < K && T.mask[i < K ? i : 0] != 0;
      off[u] = i < K ? 32 * i : 0x7ffffff0;
    }
    for (int spin = 0; spin < kTailSpin; spin++) {
      bool pending = false;
#pragma unroll
      for (int u = 0; u < U; u++) {
        TailU4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
        lo = __builtin_amdgcn_raw_buffer_load_b128(rs, off[u], 0, 16);  // (aux 16: sc1)
        if (mode != 0) hi = __builtin_amdgcn_raw_buffer_load_b128(rs, off[u] + 16, 0, 16);
        const unsigned long long w0 = ((unsigned long long)lo.y << 32) | lo.x, w1 = ((unsigned long long)lo.w << 32) | lo.z;
        const unsigned long long w2 = ((unsigned long long)hi.y << 32) | hi.x, w3 = ((unsigned long long)hi.w << 32) | hi.z;
        pending = pending || w0 == kTailSentinel || w1 == kTailSentinel || w2 == kTailSentinel || w3 == kTailSentinel;
        v[u][0] = __longlong_as_double((long long)w0);
        v[u][1] = __longlong_as_double((long long)w1);
        v[u][2] = __longlong_as_double((long long)w2);
        v[u][3] = __longlong_as_double((long long)w3);
      }
      if (!pending) break;
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!on[u]) continue;
      if (mode == 0) {
        a += v[u][0];
        b = fmax(b, v[u][1]);
      } else {
        b = fmin(b, v[u][0]);
        c = fmax(c, v[u][1]);
        a += v[u][2];
        d += v[u][3];
      }
    }
  }
  ReduceStepFinish<true>(mode, a, b, c, d, T.red_out, T.mbx, T.scal ? s_scal : nullptr, pre, &T.rule);
}

}  // namespace cxk
