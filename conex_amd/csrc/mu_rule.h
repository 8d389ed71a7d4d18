// The barrier-parameter rule of conex::Solve, ONE source for the host loop (program.cc) and for the
// device (the tail workgroup of the eigenvalue query, kernels_cone.hip.h): compiled twice, the same
// IEEE operations in the same order (no contraction: -ffp-contract=off on the device, no fused
// multiply-add on the host's baseline ISA), so both sides get the same bits.
//   conex/divergence.cc:17-110        DivergenceUpperBoundInverse and helpers
//   conex/cone_program.cc:166-224     MinimizeNormInf, ComputeMuFromDivergence, ApplyLimits
//   conex/cone_program.cc:386-392     inv_sqrt_mu <- the selection, or half of itself; limits
#pragma once
#include <math.h>

#if defined(__HIP__)
#define CXK_HD __host__ __device__ inline
#else
#define CXK_HD inline
#endif

namespace cxk_mu {

struct Wse {  // WeightedSlackEigenvalues
  double frob = 0, trace = 0, lmin = 0, lmax = 0, rank = 0;
};

CXK_HD double SolveRational(double a, double b, double c, double d, double k) {
  const double dk = d * k;  // (pow(d k, 2) in the reference: the correctly rounded square)
  const double ur = b * b - 4 * a * c + 8 * a * k + 2 * b * d * k + dk * dk;
  return -(b + d * k - sqrt(ur)) / (2 * a);
}
CXK_HD bool InLimits(double x, double lo, double hi) { return x >= lo && x <= hi; }
CXK_HD double InverseLambdaMaxBranch(double bound, const Wse& p) {
  const double x = SolveRational(p.frob, -2 * p.trace, p.rank, p.lmax, bound);
  const double lower = 2.0 / (p.lmax + p.lmin);
  return x >= lower ? x : -1;
}
CXK_HD double InverseLambdaMinBranch(double bound, const Wse& p) {
  const double upper = 2.0 / (p.lmax + p.lmin);
  const double a = p.frob / p.lmin, b = 2 * p.trace / p.lmin, n = p.rank / p.lmin, c = bound;
  const double ur = b * b + 2 * b * c + c * c - 4 * a * n;
  const double f = (b + c + sqrt(ur)) / (2 * a), s = (b + c - sqrt(ur)) / (2 * a);
  double k = -1;
  if (!(ur < 0)) {
    if (InLimits(f, 0, upper)) k = f;
    if (InLimits(s, 0, upper) && s > k) k = s;
  }
  return k;
}
CXK_HD bool BoundIsFinite(double k, const Wse& p) {
  double ni = fabs(k * p.lmax - 1);
  if (ni < fabs(k * p.lmin - 1)) ni = fabs(k * p.lmin - 1);
  return ni < 1;
}
CXK_HD double DivergenceUpperBoundInverse(double bound, const Wse& p) {
  double k = -1;
  const double k1 = InverseLambdaMinBranch(bound, p);
  const double k2 = InverseLambdaMaxBranch(bound, p);
  if (BoundIsFinite(k1, p)) k = k1;
  if (k2 > k && BoundIsFinite(k2, p)) k = k2;
  return k;
}

// ComputeMuFromDivergence :173-214 behind its solve and eigenvalue query: the selected
// inv_sqrt_mu, or a negative value
CXK_HD double SelectFromDivergence(double divergence_upper_bound, int rankK, Wse mp) {
  mp.rank = rankK;
  const double bound = divergence_upper_bound * rankK;
  double inv = DivergenceUpperBoundInverse(bound, mp);
  if (inv == -1) {  // MinimizeNormInf :166-172
    inv = -1;
    if (mp.lmin > 0) inv = 2.0 / (mp.lmin + mp.lmax);
  }
  if (inv < 0 && mp.trace > 1e-12) {
    const double kstar = mp.trace / mp.frob;
    double nb = 1.5 * (mp.frob * kstar * kstar - 2 * mp.trace * kstar + rankK);
    if (nb > rankK * .7) nb = rankK * .7;
    const double a = mp.frob, b = -2 * mp.trace, c = rankK - nb;
    if (b * b - 4 * a * c < 0)
      inv = mp.trace / mp.frob;
    else
      inv = (-b + sqrt(b * b - 4 * a * c)) / (2 * a);
  }
  return inv;
}

CXK_HD void ApplyLimits(double* x, double lb, double ub) {
  if (*x > ub) *x = ub;
  if (*x < lb) *x = lb;
}

// What Solve does with the selection (:386-392): take it, or halve the current value; then the limits.
struct Update {
  double divergence_upper_bound;
  int rankK;
  double prev;    // inv_sqrt_mu before the selection
  double lb, ub;  // ApplyLimits
};
CXK_HD double NextInvSqrtMu(const Update& u, const Wse& eig) {
  const double temp = SelectFromDivergence(u.divergence_upper_bound, u.rankK, eig);
  double inv = u.prev;
  if (temp > 0)
    inv = temp;
  else
    inv *= .5;
  ApplyLimits(&inv, u.lb, u.ub);
  return inv;
}

}  // namespace cxk_mu
