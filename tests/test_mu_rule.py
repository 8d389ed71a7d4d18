"""csrc/mu_rule.h -- the barrier-parameter rule that the host loop (program.cc) and the tail workgroup
of the eigenvalue query (kernels_cone.hip.h) both compile -- against the oracle's restatement of
divergence.cc / cone_program.cc:166-224, :386-392, on the CPU: the header is built into a small
shared object with g++ and driven over random and hand-made inputs that reach every branch
(bound branch of either end, MinimizeNormInf, the quadratic fall-back, no selection -> half the
previous value, both limits).  Bit for bit: both sides are plain IEEE doubles in the same order.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "mu_rule.h"
extern "C" double mu_next(double dub, int rankK, double prev, double lb, double ub,
                          double lmin, double lmax, double frob, double trace) {
  cxk_mu::Update u;
  u.divergence_upper_bound = dub; u.rankK = rankK; u.prev = prev; u.lb = lb; u.ub = ub;
  cxk_mu::Wse e;
  e.lmin = lmin; e.lmax = lmax; e.frob = frob; e.trace = trace;
  return cxk_mu::NextInvSqrtMu(u, e);
}
extern "C" double mu_select(double dub, int rankK, double lmin, double lmax, double frob, double trace) {
  cxk_mu::Wse e;
  e.lmin = lmin; e.lmax = lmax; e.frob = frob; e.trace = trace;
  return cxk_mu::SelectFromDivergence(dub, rankK, e);
}
'''


@pytest.fixture(scope="module")
def rule(tmp_path_factory):
    d = tmp_path_factory.mktemp("mu_rule")
    src = d / "mu_rule_test.cc"
    src.write_text(SRC)
    so = d / "libmu_rule_test.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                           "-I", os.path.join(ROOT, "conex_amd", "csrc"), str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    lib.mu_next.restype = C.c_double
    lib.mu_next.argtypes = [C.c_double, C.c_int] + [C.c_double] * 7
    lib.mu_select.restype = C.c_double
    lib.mu_select.argtypes = [C.c_double, C.c_int] + [C.c_double] * 4
    return lib


def oracle_select(dub, rankK, lmin, lmax, frob, trace):
    p5 = np.array([frob, trace, lmin, lmax, float(rankK)])
    inv = ol.lib().cxo_divergence_upper_bound_inverse(dub * rankK, ol.dp(p5))
    if inv == -1:
        inv = -1.0
        if lmin > 0:
            inv = 2.0 / (lmin + lmax)
    if inv < 0 and trace > 1e-12:
        kstar = trace / frob
        nb = 1.5 * (frob * kstar * kstar - 2 * trace * kstar + rankK)
        if nb > rankK * .7:
            nb = rankK * .7
        a, b, c = frob, -2 * trace, rankK - nb
        if b * b - 4 * a * c < 0:
            inv = trace / frob
        else:
            inv = float((-b + np.sqrt(b * b - 4 * a * c)) / (2 * a))
    return float(inv)


def oracle_next(dub, rankK, prev, lb, ub, lmin, lmax, frob, trace):
    inv = oracle_select(dub, rankK, lmin, lmax, frob, trace)
    out = inv if inv > 0 else prev * .5
    out = min(out, ub)
    out = max(out, lb)
    return float(out)


def spectra(rng, count):
    """(rank, lmin, lmax, frob, trace) of random spectra: consistent inputs, as the query produces them."""
    for _ in range(count):
        r = int(rng.integers(2, 400))
        shift = rng.choice([-2.0, -0.5, 0.0, 0.5, 2.0, 50.0])
        lam = rng.normal(size=r) * rng.choice([1e-3, 0.1, 1.0, 30.0]) + shift
        yield r, float(lam.min()), float(lam.max()), float(np.sum(lam * lam)), float(np.sum(lam))


def test_selection_matches_the_oracle_on_random_spectra(rule):
    rng = np.random.default_rng(2024)
    seen_negative = seen_positive = 0
    for r, lmin, lmax, frob, trace in spectra(rng, 4000):
        dub = float(rng.choice([1e-6, 0.1, 1.0, 10.0]))
        got = rule.mu_select(dub, r, lmin, lmax, frob, trace)
        want = oracle_select(dub, r, lmin, lmax, frob, trace)
        assert got == want or (np.isnan(got) and np.isnan(want)), (r, lmin, lmax, frob, trace, dub, got, want)
        seen_negative += want < 0
        seen_positive += want > 0
    assert seen_negative > 50 and seen_positive > 50      # both outcomes were exercised


def test_update_takes_the_selection_or_half_of_the_previous_value_within_the_limits(rule):
    rng = np.random.default_rng(7)
    took_half = clamped_hi = clamped_lo = 0
    for r, lmin, lmax, frob, trace in spectra(rng, 3000):
        dub = float(rng.choice([1e-6, 1.0, 10.0]))
        prev = float(rng.choice([0.0, 1e-3, 0.3, 40.0]))
        lb = float(rng.choice([1e-8, 1e-2, 5.0]))
        ub = float(rng.choice([0.5, 1e3, 1e9]))
        got = rule.mu_next(dub, r, prev, lb, ub, lmin, lmax, frob, trace)
        want = oracle_next(dub, r, prev, lb, ub, lmin, lmax, frob, trace)
        assert got == want, (r, lmin, lmax, frob, trace, dub, prev, lb, ub, got, want)
        sel = oracle_select(dub, r, lmin, lmax, frob, trace)
        took_half += not (sel > 0)
        clamped_hi += got == ub
        clamped_lo += got == lb
    assert took_half > 20 and clamped_hi > 20 and clamped_lo > 20


def test_hand_made_cases(rule):
    # identity-like spectrum (the first iteration: W = I, minus_s = -c I): lmin = lmax
    for r, lam in [(20000, 1.0), (20000, 0.37), (40, 2.5)]:
        args = (1.0, r, lam, lam, r * lam * lam, r * lam)
        assert rule.mu_select(*args) == oracle_select(*args)
    # a spectrum with a non-positive end: MinimizeNormInf has no answer, the quadratic fall-back decides
    args = (1.0, 10, -0.5, 2.0, 9.0, 6.0)
    assert rule.mu_select(*args) == oracle_select(*args)
    # trace <= 1e-12: nothing is selected, the update halves the previous value
    assert rule.mu_select(1.0, 10, -2.0, -0.1, 9.0, -6.0) == oracle_select(1.0, 10, -2.0, -0.1, 9.0, -6.0)
    assert rule.mu_next(1.0, 10, 0.8, 1e-8, 1e9, -2.0, -0.1, 9.0, -6.0) == 0.4
