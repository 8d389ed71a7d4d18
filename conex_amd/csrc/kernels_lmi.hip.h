// Dense-LMI (real PSD cone) kernels, LDS-resident formulation for small orders.
// One 256-thread workgroup per constraint; W, one A_i and two n x n temporaries live in
// LDS.  These are the shape-generic kernels (any n with 4 n^2 doubles <= LDS budget);
// lmi_fused_mfma.hip holds the fp64-MFMA kernel of the register-resident orders (the benchmark
// shape among them), kernels_lmi_large.hip.h the HBM-resident path for orders beyond LDS.
//
// Reference semantics reproduced here:
//   ConstructSchurComplementSystem(DenseLMIConstraint*)  dense_lmi_constraint.cc:72-103
//   PrepareStep(PsdConstraint*)                          psd_constraint.cc:45-84
//   TakeStep / GeodesicUpdate                            psd_constraint.cc:13-28, 86-90
//   AffineUpdate                                         psd_constraint.cc:33-43
//   GetWeightedSlackEigenvalues(PsdConstraint*)          psd_constraint.cc:97-128
//   AsymmetricLanczos                                    approximate_eigenvalues.cc:178-239
//   ExponentialMapPadeApproximation                      exponential_map_pade.cc:10-32
#pragma once
#include "device_utils.h"
#include "lmi_types.h"

namespace cxk {


// Safeguard on the Lanczos estimates (not in the reference).  The unreorthogonalised two-sided
// Lanczos of approximate_eigenvalues.cc:178-239 turns into noise once the Krylov space is nearly
// invariant: beta^2 then hovers around the 1e-6 break threshold, the next vectors are rounding
// error divided by beta, and a Ritz value far outside the spectrum can come out (seen at iteration
// 8 of the C4 solve: -52 for a spectrum in [-1.67, -0.32]; which run gets hit depends on summation
// order, the CPU restatement has the same failure mode on other inputs).  The eigenvalues of W S
// are real (W S is similar to W^1/2 S W^1/2), so with t1 = tr(WS), t2 = tr(WS WS) Samuelson's
// inequality bounds every one of them by  t1/n +- sqrt((n-1) (t2/n - (t1/n)^2)).  Ritz values of
// a healthy run lie inside the spectrum and are untouched; only provably wrong ones are clamped.
__device__ __forceinline__ void ClampToSpectrumBound(int n, double t1, double t2, double* mn, double* mx) {
  const double mean = t1 / n;
  double var = t2 / n - mean * mean;
  if (!(var > 0.0)) var = 0.0;
  const double half = sqrt((n - 1) * var);
  const double lo = mean - half, hi = mean + half;
  if (!(*mn >= lo)) *mn = lo;  // also catches NaN
  if (!(*mn <= hi)) *mn = hi;
  if (!(*mx <= hi)) *mx = hi;
  if (!(*mx >= lo)) *mx = lo;
}

// Stateless generator shared with the oracle (oracle/cxo_hermitian.c cxo_hc_random): the
// reference's start vector is libc rand() and unpinned.
__device__ __forceinline__ double HcRandom(unsigned long long id, unsigned long long call,
                                           unsigned long long idx) {
  unsigned long long z = 0x243F6A8885A308D3ull + id * 0x9E3779B97F4A7C15ull +
                         call * 0xD1B54A32D192ED03ull + idx * 0x8CB92BA72F3D8DD7ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}


// out(n x n) = X(n x n) * Y(n x n), all in LDS, column-major. Caller syncs.
// From order 32 up each thread owns a 2 x 2 block of the product (rows r, r+1; columns c, c+1):
// four LDS reads feed four FMAs instead of eight reads -- the plain one-output-per-thread form is
// LDS-issue bound there.  Every output is still one fma chain over k = 0..n-1 (same rounding as before).
__device__ __forceinline__ void LdsGemm(int n, const double* X, const double* Y, double* out) {
  if (n * n < 4 * (int)blockDim.x) {  // small orders: one output per thread keeps every lane busy
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
      const int r = idx % n, c = idx / n;
      double s = 0;
      for (int k = 0; k < n; k++) s = fma(X[r + k * n], Y[k + c * n], s);
      out[idx] = s;
    }
    return;
  }
  const int hr = (n + 1) >> 1;  // row pairs (and column pairs)
  for (int idx = threadIdx.x; idx < hr * hr; idx += blockDim.x) {
    const int r = 2 * (idx % hr), c = 2 * (idx / hr);
    const bool r1 = r + 1 < n, c1 = c + 1 < n;
    const double* x = X + r;
    const double* y0 = Y + (size_t)c * n;
    const double* y1 = Y + (size_t)(c1 ? c + 1 : c) * n;
    double s00 = 0, s10 = 0, s01 = 0, s11 = 0;
    for (int k = 0; k < n; k++) {
      const double xa = x[k * n], xb = r1 ? x[k * n + 1] : 0.0;
      const double ya = y0[k], yb = y1[k];
      s00 = fma(xa, ya, s00);
      s10 = fma(xb, ya, s10);
      s01 = fma(xa, yb, s01);
      s11 = fma(xb, yb, s11);
    }
    out[r + (size_t)c * n] = s00;
    if (r1) out[r + 1 + (size_t)c * n] = s10;
    if (c1) {
      out[r + (size_t)(c + 1) * n] = s01;
      if (r1) out[r + 1 + (size_t)(c + 1) * n] = s11;
    }
  }
}

__global__ void __launch_bounds__(256) lmi_schur_generic(LmiGroup g, Arena ar) {
  extern __shared__ double lds[];
  const int n = g.n, m = g.m, nn = n * n;
  double* sW = lds;
  double* sA = sW + nn;
  double* sP = sA + nn;
  double* sX = sP + nn;
  const int mem = blockIdx.x;
  const int id = g.ids[mem];
  const double* A = g.A + (size_t)mem * g.a_stride;
  const double* Cm = g.C + (size_t)mem * nn;
  const double* Wg = g.W + (size_t)mem * nn;
  double* G = ar.G + ar.g_off[id];
  double* AW = ar.AWc + ar.r_off[id];
  double* AQc = ar.AQcc + ar.r_off[id];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const double osc = g.herm_d > 1 ? 1.0 / g.herm_d : 1.0;  // exact (power of two)

  for (int q = threadIdx.x; q < nn; q += blockDim.x) sW[q] = Wg[q];
  for (int i = 0; i <= m; i++) {
    const double* M = (i < m) ? A + (size_t)i * nn : Cm;
    __syncthreads();
    for (int q = threadIdx.x; q < nn; q += blockDim.x) sA[q] = M[q];
    __syncthreads();
    LdsGemm(n, sA, sW, sP);  // AW = A_i * W
    __syncthreads();
    LdsGemm(n, sW, sP, sX);  // WAW = W * AW
    __syncthreads();
    if (i < m) {
      for (int task = wave; task <= i + 2; task += nwaves) {
        double v = 0;
        if (task <= i) {
          const double* Aj = A + (size_t)task * nn;
          for (int q = lane; q < nn; q += 64) v = fma(sX[q], Aj[q], v);
        } else if (task == i + 1) {
          for (int r = lane; r < n; r += 64) v += sP[r + r * n];
        } else {
          for (int q = lane; q < nn; q += 64) v = fma(Cm[q], sX[q], v);
        }
        v = WaveSum(v);
        if (lane == 0) {
          v *= osc;
          if (task <= i)
            G[i + (size_t)task * m] = v;
          else if (task == i + 1)
            AW[i] = v;
          else
            AQc[i] = v;
        }
      }
    } else {
      if (wave == 0) {
        double v = 0;
        for (int q = lane; q < nn; q += 64) v = fma(Cm[q], sW[q], v);
        v = WaveSum(v);
        if (lane == 0) ar.sc[2 * id] = v * osc;
      } else if (wave == 1) {
        double v = 0;
        for (int q = lane; q < nn; q += 64) v = fma(Cm[q], sX[q], v);
        v = WaveSum(v);
        if (lane == 0) ar.sc[2 * id + 1] = v * osc;
      }
    }
  }
}

// Extreme eigenvalues of a symmetric tridiagonal (the Jacobi matrix of the Lanczos run) by
// 32-way multisection on Sturm counts, executed by ONE wavefront: lanes 0-31 bracket the smallest
// eigenvalue, lanes 32-63 the largest; every round each lane evaluates one shift, a ballot picks
// the sub-interval (x* = inf{x : #eigenvalues below x >= target}), 10 rounds shrink the Gershgorin
// interval by 33^10 > 2^50.  Replaces the sequential QL iteration (which cost ~100 us per launch
// on one lane); the reference only consumes min / max of
// SelfAdjointEigenSolver::computeFromTridiagonal (approximate_eigenvalues.cc:232-238), and both
// methods are accurate to a few ulps of the matrix norm.  d, e are read-only (LDS).
// MAXN > 0: n <= MAXN, the matrix is copied into registers once (a lone wavefront pays ~50 cycles per
// LDS read: two per step of every count, ten counts).
template <int MAXN = 0>
__device__ __forceinline__ void TridiagMinMaxWave(int n, const double* d, const double* e, double* mn, double* mx) {
  const int lane = threadIdx.x & 63;
  if (n == 1) {
    if (lane == 0) *mn = *mx = d[0];
    return;
  }
  double glo = 1.7976931348623157e308, ghi = -1.7976931348623157e308;
  for (int i = lane; i < n; i += 64) {
    const double r = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i < n - 1 ? fabs(e[i]) : 0.0);
    glo = fmin(glo, d[i] - r);
    ghi = fmax(ghi, d[i] + r);
  }
  glo = -WaveMax(-glo);
  ghi = WaveMax(ghi);
  // The matrix is mapped onto [0, 1] first -- T' = (T - glo) / width has the counts of T at the mapped shifts
  // -- so that |d' - x| <= 1 and e'^2 <= 1/4 bound every minor by 1.25^n: the register form (MAXN > 0:
  // n <= 17) then needs no rescaling test in its recurrence (a third of its instructions on this lone
  // wavefront: the ten rounds took 8.5 of PrepareStep's 25 us chain); the memory form (any n) keeps the
  // test, which multiplies by 1.0 at these sizes: the same bits.  The result is mapped back.
  const double width = ghi - glo;
  if (!(width > 1e-300)) {  // (wave-uniform: a multiple of the identity)
    if (lane == 0) *mn = *mx = d[0];
    return;
  }
  const double inv = 1.0 / width, org = glo;
  const double pad = 1e-12;  // count(0 - pad) = 0, count(1 + pad) = n
  double a = 0.0 - pad, b = 1.0 + pad;
  const int half = lane >> 5, sub = lane & 31;
  const int target = half == 0 ? 1 : n;
  double dd[MAXN > 0 ? MAXN : 1], e2[MAXN > 0 ? MAXN : 1];
  if constexpr (MAXN > 0) {
#pragma unroll
    for (int i = 0; i < MAXN; i++) {
      dd[i] = (d[i < n ? i : 0] - org) * inv;
      const double ev = e[(i >= 1 && i < n) ? i - 1 : 0] * inv;
      e2[i] = ev * ev;
    }
  }
  for (int round = 0; round < 10; round++) {
    const double w = b - a;
    const double x = a + w * ((sub + 1) * (1.0 / 33.0));
    // Sturm count as sign changes of the leading principal minors p_0 = 1, p_1 = d_0 - x,
    // p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1} (the pivot of the LDL^T recurrence is their ratio):
    // two dependent fmas per step where the pivot recurrence waits for a reciprocal and its Newton
    // step (~70 cycles a step on a lone wavefront; the ten rounds took 7 of PrepareStep's 24 us).
    // The memory form (any n) rescales when a minor leaves [1e-150, 1e150]; an exact zero keeps the sign
    // of its predecessor.
    double pm = 1.0, p = (MAXN > 0 ? dd[0] : (d[0] - org) * inv) - x;
    bool neg = p < 0.0;
    int c = neg;
    auto step = [&](double di, double ee) {
      double pn = fma(di - x, p, -(ee * pm));
      if constexpr (MAXN > 0) {
        pm = p;
        p = pn;
      } else {
        const double mag = fabs(pn);
        const double sc = mag > 1e150 ? 1e-150 : ((mag < 1e-150 && mag > 0.0) ? 1e150 : 1.0);
        pm = p * sc;
        p = pn * sc;
      }
      const bool ng = p < 0.0;
      c += (p != 0.0 && ng != neg);
      neg = p != 0.0 ? ng : neg;
    };
    if constexpr (MAXN > 0) {
#pragma unroll
      for (int i = 1; i < MAXN; i++)
        if (i < n) step(dd[i], e2[i]);  // wave-uniform
    } else {
#pragma unroll 4
      for (int i = 1; i < n; i++) {
        const double ev = e[i - 1] * inv;
        step((d[i] - org) * inv, ev * ev);
      }
    }
    const unsigned long long m = __ballot(c >= target);
    const unsigned int mh = half ? (unsigned int)(m >> 32) : (unsigned int)(m & 0xffffffffull);
    const int first = mh ? __ffs(mh) - 1 : 32;  // first probe at or beyond x*
    const double na = a + w * (first * (1.0 / 33.0));
    const double nb = first == 32 ? b : a + w * ((first + 1) * (1.0 / 33.0));
    a = na;
    b = nb;
  }
  const double val = fma(0.5 * (a + b), width, org);
  if (lane == 0) *mn = val;
  if (lane == 32) *mx = val;
}

// Two-sided Lanczos on WS with V = [W r, r] run by wave 0; result (min,max eigenvalue of the
// Jacobi matrix) is left in out[0], out[1] (LDS).  vec: 6n doubles of LDS; ab: 2*num_iter+2.
// herm = false: AsymmetricLanczos of approximate_eigenvalues.cc (absolute break 1e-6, divisions);
// herm = true: MatrixAlgebra<d>::ApproximateEigenvalues of jordan_matrix_algebra.cc:386-452
// (break relative to <U,U> of the first step, normalisations by reciprocal multiply).
__device__ __forceinline__ void LanczosWave0(int n, const double* sWS, const double* sW, const double* r,
                                    int num_iter, double* vec, double* ab, double* out,
                                    bool herm = false) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  if (n == 1) {
    if (lane == 0) out[0] = out[1] = sWS[0];
    return;
  }
  double* V0 = vec;
  double* V1 = vec + n;
  double* U0 = vec + 2 * n;
  double* U1 = vec + 3 * n;
  double* P0 = vec + 4 * n;
  double* P1 = vec + 5 * n;
  double* alpha = ab;
  double* beta = ab + num_iter + 1;
  // V.col(1) = r ; V.col(0) = W r
  for (int i = lane; i < n; i += 64) {
    V1[i] = r[i];
    double s = 0;
    for (int k = 0; k < n; k++) s = fma(sW[i + k * n], r[k], s);
    V0[i] = s;
  }
  double ip = 0;
  for (int i = lane; i < n; i += 64) ip = fma(V0[i], V1[i], ip);
  const double nrm = sqrt(WaveSum(ip));
  const double inrm = 1.0 / nrm;
  for (int i = lane; i < n; i += 64) {
    V0[i] = herm ? V0[i] * inrm : V0[i] / nrm;
    V1[i] = herm ? V1[i] * inrm : V1[i] / nrm;
    P0[i] = V0[i];
    P1[i] = V1[i];
  }
  int cnt = 0;
  double beta_prev = 0, scaling = 0;
  for (int j = 0; j < num_iter; j++) {
    if (j > 0) {
      double b2 = 0;
      for (int i = lane; i < n; i += 64) b2 = fma(U0[i], U1[i], b2);
      b2 = WaveSum(b2);
      if (herm ? (b2 < 1e-5 * scaling) : (b2 < 1e-6)) break;
      beta_prev = sqrt(b2);
      if (lane == 0) beta[j - 1] = beta_prev;
      const double ib = 1.0 / beta_prev;
      for (int i = lane; i < n; i += 64) {
        P0[i] = V0[i];
        P1[i] = V1[i];
        V0[i] = herm ? U0[i] * ib : U0[i] / beta_prev;
        V1[i] = herm ? U1[i] * ib : U1[i] / beta_prev;
      }
      cnt++;
    }
    // U.col(0) = WS V.col(0) ; U.col(1) = WS^T V.col(1)
    double a = 0;
    for (int i = lane; i < n; i += 64) {
      double s0 = 0, s1 = 0;
      for (int k = 0; k < n; k++) {
        s0 = fma(sWS[i + k * n], V0[k], s0);
        s1 = fma(sWS[k + i * n], V1[k], s1);
      }
      U0[i] = s0;
      U1[i] = s1;
    }
    if (j == 0 && herm) {
      double sc = 0;
      for (int i = lane; i < n; i += 64) sc = fma(U0[i], U1[i], sc);
      scaling = WaveSum(sc);
    }
    for (int i = lane; i < n; i += 64) a = fma(V0[i], U1[i], a);
    a = WaveSum(a);
    if (lane == 0) alpha[j] = a;
    for (int i = lane; i < n; i += 64) {
      double u0 = U0[i] - a * V0[i], u1 = U1[i] - a * V1[i];
      if (j > 0) {
        u0 -= beta_prev * P0[i];
        u1 -= beta_prev * P1[i];
      }
      U0[i] = u0;
      U1[i] = u1;
    }
  }
  WaveSync();  // alpha / beta were written by lane 0
  TridiagMinMaxWave(cnt + 1, alpha, beta, &out[0], &out[1]);
}


// mode 0: PrepareStep ; mode 1: GetWeightedSlackEigenvalues.  NF > 0 fixes the order at compile
// time (the helpers are force-inlined, so their loops unroll and the index arithmetic folds: a
// lone wavefront issues one dependent instruction per ~8 cycles, instruction count is latency).
template <int MODE, int NF = 0>
__global__ void __launch_bounds__(256) lmi_prepare_generic(LmiGroup g, StepArgs sa) {
  sa.c_weight = CWeightOf(sa);  // (the barrier parameter may live on the device: cxk_select_mu_async)
  extern __shared__ double lds[];
  const int n = NF > 0 ? NF : g.n, m = g.m, nn = n * n;
  double* sW = lds;
  double* sS = sW + nn;
  double* sWS = sS + nn;
  double* vec = sWS + nn;           // 6n
  double* ab = vec + 6 * n;         // 2*(n/2+1)+2
  double* sy = ab + 2 * (n / 2 + 2);  // m
  double* red = sy + m;             // 8
  const int mem = blockIdx.x;
  const int id = g.ids[mem];
  const double* A = g.A + (size_t)mem * g.a_stride;
  const double* Cm = g.C + (size_t)mem * nn;
  double* Wg = g.W + (size_t)mem * nn;
  double* T1 = g.T1 + (size_t)mem * nn;

  for (int q = threadIdx.x; q < m; q += blockDim.x) sy[q] = sa.y[sa.cl_perm[sa.cl_ptr[id] + q]];
  for (int q = threadIdx.x; q < nn; q += blockDim.x) sW[q] = Wg[q];
  __syncthreads();
  // minus_s = sum_i y_i A_i - k C   (dense_lmi_constraint.cc:8-27)
  if (g.sp_pptr) {
    // sparse group: one thread per position sums its nonzeros in the reference's order of i
    // (the skipped terms are exact zeros)
    const int* pp = g.sp_pptr + (size_t)mem * nn;
    for (int q = threadIdx.x; q < nn; q += blockDim.x) {
      double s = 0;
      for (int e = pp[q]; e < pp[q + 1]; e++) s += sy[g.sp_pvar[e]] * g.sp_pval[e];
      sS[q] = s - sa.c_weight * Cm[q];
    }
  } else if ((nn & 1) == 0) {
    // A is streamed once more here (m n^2 doubles per constraint): 16-byte loads, eight matrices
    // in flight per thread; the sum over i keeps the reference's order
    const int half = nn >> 1;
    for (int q2 = threadIdx.x; q2 < half; q2 += blockDim.x) {
      double s0 = 0, s1 = 0;
      const double2* base = reinterpret_cast<const double2*>(A) + q2;
      constexpr int kBatch = 20;  // one memory round trip for the C4 shape (m = 20)
      for (int i0 = 0; i0 < m; i0 += kBatch) {
        double2 v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++)
          v[u] = (i0 + u < m) ? base[(size_t)(i0 + u) * half] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < kBatch; u++)
          if (i0 + u < m) {
            s0 += sy[i0 + u] * v[u].x;
            s1 += sy[i0 + u] * v[u].y;
          }
      }
      sS[2 * q2] = s0 - sa.c_weight * Cm[2 * q2];
      sS[2 * q2 + 1] = s1 - sa.c_weight * Cm[2 * q2 + 1];
    }
  } else {
    for (int q = threadIdx.x; q < nn; q += blockDim.x) {
      double s = 0;
      for (int i = 0; i < m; i++) s += sy[i] * A[(size_t)i * nn + q];
      s -= sa.c_weight * Cm[q];
      sS[q] = s;
    }
  }
  __syncthreads();
  LdsGemm(n, sW, sS, sWS);  // WS = W * minus_s
  __syncthreads();
  if (MODE == 0) {
    for (int q = threadIdx.x; q < nn; q += blockDim.x) T1[q] = sWS[q];
    if (sa.affine) {  // AffineUpdate: W = W (1 + w_e) + (WS) W
      LdsGemm(n, sWS, sW, sS);
      __syncthreads();
      for (int q = threadIdx.x; q < nn; q += blockDim.x) {
        const double w = (sa.e_weight == 0) ? sW[q] : sW[q] * (1 + sa.e_weight);
        Wg[q] = w + sS[q];
      }
      return;
    }
  }
  // index of the first maximal diagonal entry of WS
  __shared__ int s_index;
  if (threadIdx.x == 0) {
    int idx = 0;
    for (int i = 1; i < n; i++)
      if (sWS[i + i * n] > sWS[idx + idx * n]) idx = i;
    s_index = idx;
  }
  __syncthreads();
  // PrepareStep aliases minus_s and WS (both temp_1), so its start vector is WS.col(index);
  // GetWeightedSlackEigenvalues keeps them apart and starts from minus_s.col(index).
  const double* r = (MODE == 0 ? sWS : sS) + s_index * n;
  int iters = n / 2;
  if (g.herm_d) {  // T::Random(n, 1), n/2 + 1 iterations on the hyper-complex order
    double* rnd = vec + 5 * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) rnd[i] = HcRandom(id, sa.call, i);
    __syncthreads();
    r = rnd;
    iters = (n / g.herm_d) / 2 + 1;
  }
  LanczosWave0(n, sWS, sW, r, iters, vec, ab, red + 4, g.herm_d != 0);
  // tr(WS*WS) and tr(WS)
  double t2 = 0, t1 = 0;
  for (int q = threadIdx.x; q < nn; q += blockDim.x) {
    const int a = q % n, b = q / n;
    t2 = fma(sWS[q], sWS[b + a * n], t2);
    if (a == b) t1 += sWS[q];
  }
  t2 = BlockSum(t2, red);
  t1 = BlockSum(t1, red);
  __syncthreads();
  if (threadIdx.x == 0) {
    double mn = red[4], mx = red[5];
    if (!sa.no_clamp) ClampToSpectrumBound(n, t1, t2, &mn, &mx);
    if (g.herm_d > 1) {  // traces over the real representation are d x the reference's
      t2 /= g.herm_d;
      t1 /= g.herm_d;
    }
    const int rank = g.herm_d ? n / g.herm_d : n;
    if (MODE == 0) {
      const double l1 = fabs(sa.e_weight + mn), l2 = fabs(sa.e_weight + mx);
      sa.info[2 * id] = t2 + 2 * t1 + rank;
      sa.info[2 * id + 1] = l1 < l2 ? l2 : l1;
    } else {
      sa.info[4 * id] = -mx;      // lambda_min
      sa.info[4 * id + 1] = -mn;  // lambda_max
      sa.info[4 * id + 2] = t2;
      sa.info[4 * id + 3] = -t1;
    }
  }
}

// W <- sym( pade33( (WS + e I) * alpha ) * W )
template <int NF = 0>
__global__ void __launch_bounds__(256) lmi_take_step_generic(LmiGroup g, StepArgs sa) {
  if (StepSkipped(sa)) return;  // (enqueued before the host saw the factorization fail: leave W alone)
  extern __shared__ double lds[];
  const int n = NF > 0 ? NF : g.n, nn = n * n;
  double* sW = lds;
  double* sX = sW + nn;
  double* sV = sX + nn;
  double* aug = sV + nn;  // n x 2n : [denom | numer]
  __shared__ int s_piv;
  const int mem = blockIdx.x;
  double* Wg = g.W + (size_t)mem * nn;
  const double* T1 = g.T1 + (size_t)mem * nn;
  for (int q = threadIdx.x; q < nn; q += blockDim.x) {
    sW[q] = Wg[q];
    double x = T1[q];
    if (q % n == q / n) x += sa.e_weight;
    const double step = StepSizeOf(sa);
    if (step != 1.0) x *= step;
    sX[q] = x;
  }
  __syncthreads();
  if (g.herm_d) {
    // DoExponentialMap (exponential_map.cc:15-43): E = ((I + X/4 + X^2/32)^2)^2, then
    // W <- sym(E W).  aug: Y, aug + nn: scratch.
    double* sY = aug;
    double* sT = aug + nn;
    for (int q = threadIdx.x; q < nn; q += blockDim.x) sV[q] = sX[q] * 1.0 / 4.0;
    __syncthreads();
    LdsGemm(n, sX, sV, sT);  // X * (X/4)
    __syncthreads();
    for (int q = threadIdx.x; q < nn; q += blockDim.x)
      sY[q] = (sV[q] + ((q % n == q / n) ? 1.0 : 0.0)) + sT[q] * 0.125;
    __syncthreads();
    LdsGemm(n, sY, sY, sT);
    __syncthreads();
    LdsGemm(n, sT, sT, sY);
    __syncthreads();
    LdsGemm(n, sY, sW, sV);  // E * W
    __syncthreads();
    for (int q = threadIdx.x; q < nn; q += blockDim.x) {
      const int a = q % n, b = q / n;
      Wg[q] = (sV[q] + sV[b + a * n]) * 0.5;
    }
    return;
  }
  LdsGemm(n, sX, sX, sV);  // A^2
  __syncthreads();
  // tmp = A^2 + 60 I (in aug[0:nn]) ; U = A * tmp (in aug[nn:2nn])
  for (int q = threadIdx.x; q < nn; q += blockDim.x) aug[q] = sV[q] + ((q % n == q / n) ? 60.0 : 0.0);
  __syncthreads();
  LdsGemm(n, sX, aug, aug + nn);
  __syncthreads();
  for (int q = threadIdx.x; q < nn; q += blockDim.x) {
    const double v = sV[q] * 12.0 + ((q % n == q / n) ? 120.0 : 0.0);
    const double u = aug[nn + q];
    aug[q] = -u + v;      // denom
    aug[nn + q] = u + v;  // numer
  }
  __syncthreads();
  // LU with partial pivoting on denom, carrying the n right-hand sides along.
  const int n2 = 2 * n;
  for (int k = 0; k < n; k++) {
    if (threadIdx.x == 0) {
      int piv = k;
      double best = fabs(aug[k + k * n]);
      for (int i = k + 1; i < n; i++) {
        const double v = fabs(aug[i + k * n]);
        if (v > best) {
          best = v;
          piv = i;
        }
      }
      s_piv = piv;
    }
    __syncthreads();
    const int piv = s_piv;
    if (piv != k) {
      for (int c = threadIdx.x; c < n2; c += blockDim.x) {
        const double t = aug[k + c * n];
        aug[k + c * n] = aug[piv + c * n];
        aug[piv + c * n] = t;
      }
      __syncthreads();
    }
    const double d = aug[k + k * n];
    for (int i = k + 1 + threadIdx.x; i < n; i += blockDim.x) aug[i + k * n] /= d;
    __syncthreads();
    const int rows = n - k - 1, cols = n2 - k - 1;
    for (int idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) {
      const int i = k + 1 + idx % rows, c = k + 1 + idx / rows;
      aug[i + c * n] -= aug[i + k * n] * aug[k + c * n];
    }
    __syncthreads();
  }
  // back substitution, one right-hand side per thread
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    double* b = aug + nn + c * n;
    for (int j = n - 1; j >= 0; j--) {
      b[j] /= aug[j + j * n];
      const double bj = b[j];
      for (int i = 0; i < j; i++) b[i] -= aug[i + j * n] * bj;
    }
  }
  __syncthreads();
  LdsGemm(n, aug + nn, sW, sV);  // expWS * W
  __syncthreads();
  for (int q = threadIdx.x; q < nn; q += blockDim.x) {
    const int a = q % n, b = q / n;
    Wg[q] = (sV[q] + sV[b + a * n]) * 0.5;
  }
}

__global__ void lmi_set_identity(LmiGroup g) {
  const int nn = g.n * g.n;
  const size_t total = (size_t)g.count * nn;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < total;
       q += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(q % nn);
    g.W[q] = (e % g.n == e / g.n) ? 1.0 : 0.0;
  }
}

}  // namespace cxk
