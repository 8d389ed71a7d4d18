"""The tail workgroup of the PrepareStep / eigenvalue-query launch (StepTail, kernels_cone.hip.h).

When every constraint of a program goes through lmi_prepare_rows on one GPU, the reduction of the
per-constraint step outputs (cone_program.cc:417-418 reads their sum and maximum), the by / cx
scalars (:439-446) and the mailbox write ride in that launch instead of two launches of their own
behind it.  Same operations in the same order: every number must come back BIT FOR BIT as from the
separate launches (CXK_NO_STEP_TAIL=1 at context creation), over repeated calls (the hand-over slots
are re-armed by the launch that consumed them) and with more constraints than one polling trip of
the tail covers.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _contexts(prob, monkeypatch):
    from conex_amd import KktContext
    with_tail = syn.build(KktContext, prob, "lmi", device=0)
    monkeypatch.setenv("CXK_NO_STEP_TAIL", "1")
    without = syn.build(KktContext, prob, "lmi", device=0)
    monkeypatch.delenv("CXK_NO_STEP_TAIL")
    return with_tail, without


def _round(k, b, inv_sqrt_mu, defer):
    """One iteration's worth of calls as conex::Solve issues them (program.cc)."""
    k.set_cost(b)
    k.kkt_solve_async(inv_sqrt_mu, 0.9, 0.8)
    assert k.sync() == 1
    e4 = k.weighted_slack_eigenvalues(None, 0.8)
    if defer:
        assert k.L.cxk_step_scalars_async(k.h) == 0  # normally rides in the PrepareStep below
    n2, ninf, took = k.prepare_take_step(None, inv_sqrt_mu * 0.8)
    sc = k.step_scalars()
    return np.r_[e4, n2, ninf, sc], took


@pytest.mark.parametrize("K,m", [(60, 20), (1000, 20), (2300, 3)])
def test_tail_workgroup_and_separate_launches_agree_bit_for_bit(K, m, monkeypatch):
    prob = syn.lmi_problem(K=K, n=20, m=m, branching=8, overlap=min(5, m - 1), seed=31 + K)
    a, b = _contexts(prob, monkeypatch)
    W = syn.scaling_points(K, 20, seed=5)
    for k in (a, b):
        for i in range(K):
            k.set_W(i, W[i])
    for it in range(3):
        ra, ta = _round(a, prob["b"], 0.7 + 0.1 * it, defer=True)
        rb, tb = _round(b, prob["b"], 0.7 + 0.1 * it, defer=True)
        assert ta == tb
        assert np.all(np.isfinite(ra))
        assert np.array_equal(ra, rb), (it, ra, rb)
        assert np.array_equal(a.get_y(), b.get_y())
    assert np.array_equal(a.get_W(K - 1), b.get_W(K - 1))


def test_deferred_step_scalars_go_out_before_any_other_call(monkeypatch):
    """cxk_step_scalars_async only notes the request; whatever runs next that is not the PrepareStep
    (here: a new right-hand side, then the blocking read) must see the scalars of the y that was
    current when they were asked for."""
    prob = syn.lmi_problem(K=40, n=20, m=20, branching=4, overlap=5, seed=77)
    a, b = _contexts(prob, monkeypatch)
    for k in (a, b):
        k.set_cost(prob["b"])
        k.kkt_solve_async(0.7, 0.9, 0.8)
        assert k.sync() == 1
    y0 = a.get_y()
    want = b.step_scalars()
    assert a.L.cxk_step_scalars_async(a.h) == 0
    a.set_y(2.0 * y0)              # flushes the deferred launch first
    got = a.step_scalars()
    assert np.array_equal(got, want)
    # and without the request the read computes them from the current y
    got2 = a.step_scalars()
    b.set_y(2.0 * y0)
    assert np.array_equal(got2, b.step_scalars())


def test_prepare_step_results_match_the_oracle_with_the_tail(monkeypatch):
    prob = syn.lmi_problem(K=30, n=20, m=20, branching=4, overlap=5, seed=3)
    from conex_amd import KktContext
    k, o = syn.build(KktContext, prob, "lmi", device=0), syn.build(ol.Program, prob, "lmi")
    W = syn.scaling_points(30, 20, seed=9)
    for p in (k, o):
        for i in range(30):
            p.set_W(i, W[i])
    oko, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    okk, yk = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert oko == okk == 1
    io, ik = o.prepare_step(yo, 0.8, 1.0), k.prepare_step(yo, 0.8, 1.0)
    assert np.allclose(ik, io, rtol=1e-9)
    eo, ek = o.weighted_slack_eigenvalues(yo, 0.8), k.weighted_slack_eigenvalues(yo, 0.8)
    assert np.allclose(ek, eo, rtol=1e-9)
