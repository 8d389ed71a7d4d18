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


# ------------------------------------------------------------------ the barrier parameter on the device
def _host_rule(dub, rankK, e4, prev, lb, ub):
    """ComputeMuFromDivergence's rule and Solve's update of inv_sqrt_mu (cone_program.cc:166-224,
    :386-392), restated over the oracle's DivergenceUpperBoundInverse -- plain IEEE doubles in the
    reference's order of operations."""
    import ctypes as C
    lmin, lmax, frob, trace = (float(v) for v in e4)
    p5 = np.array([frob, trace, lmin, lmax, float(rankK)])
    bound = dub * rankK
    inv = ol.lib().cxo_divergence_upper_bound_inverse(bound, ol.dp(p5))
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
            inv = (-b + np.sqrt(b * b - 4 * a * c)) / (2 * a)
    out = inv if inv > 0 else prev * .5
    out = min(out, ub)
    out = max(out, lb)
    return float(out)


@pytest.mark.parametrize("dub,prev,lb,ub", [(1.0, 0.0, 1e-8, 1e9), (1e-9, 0.3, 1e-8, 1e9), (50.0, 0.3, 1e-8, 1e9),
                                            (1.0, 0.3, 1e-8, 0.01), (1.0, 0.3, 5.0, 1e9)])
def test_mu_selected_on_the_device_is_the_host_rule_bit_for_bit(dub, prev, lb, ub, monkeypatch):
    import ctypes as C
    K = 48
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=4, overlap=5, seed=19)
    a, b = _contexts(prob, monkeypatch)   # a: tail + device mu, b: separate launches, driven through the host rule
    assert a.L.cxk_device_mu_supported(a.h) == 1
    W = syn.scaling_points(K, 20, seed=6)
    bs, cs, rankK = 0.9, 0.8, 20 * K
    for k in (a, b):
        for i in range(K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
        k.assemble()
        k.factor_solve_async(-bs, cs, 0.0)          # the mu-selection solve rides in the factorization
    # host path
    e4 = b.weighted_slack_eigenvalues(None, cs)
    inv_host = _host_rule(dub, rankK, e4, prev, lb, ub)
    b._check(b.L.cxk_newton_direction(b.h, inv_host, bs, cs), "cxk_newton_direction")
    assert b.L.cxk_step_scalars_async(b.h) == 0
    n2b, ninfb, tookb = b.prepare_take_step(None, inv_host * cs)
    scb = b.step_scalars()
    # device path: nothing waits until the last call
    a._check(a.L.cxk_select_mu_async(a.h, cs, dub, rankK, prev, lb, ub), "cxk_select_mu_async")
    a._check(a.L.cxk_newton_direction_device_mu(a.h, bs, cs), "cxk_newton_direction_device_mu")
    assert a.L.cxk_step_scalars_async(a.h) == 0
    info, took, inv_dev = np.zeros(2), C.c_int(0), C.c_double(0)
    a._check(a.L.cxk_prepare_take_step_device_mu(a.h, cs, 1.0, ol.dp(info), C.byref(took), C.byref(inv_dev)),
             "cxk_prepare_take_step_device_mu")
    sca = a.step_scalars()
    assert inv_dev.value == inv_host, (inv_dev.value, inv_host)
    assert took.value == 1 and tookb
    assert info[0] == n2b and info[1] == ninfb
    assert np.array_equal(sca, scb)
    assert np.array_equal(a.get_y(), b.get_y())
    for i in (0, K // 2, K - 1):
        assert np.array_equal(a.get_W(i), b.get_W(i))


def test_take_step_behind_a_failed_factorization_leaves_w_alone(monkeypatch):
    """With mu selected on the device TakeStep is enqueued before the host has seen the LLT flag: it
    must not touch W when the factorization failed (the reference returns before it, cone_program.cc:360-365)."""
    import ctypes as C
    K = 24
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=4, overlap=5, seed=23)
    from conex_amd import KktContext
    k = syn.build(KktContext, prob, "lmi", device=0)
    W = syn.scaling_points(K, 20, seed=7)
    for i in range(K):
        k.set_W(i, W[i])
    k.set_cost(prob["b"])
    k.assemble()
    k.set_slab(-k.slab())                                   # negative definite: the first pivot fails
    k._check(k.L.cxk_factor_async(k.h), "cxk_factor_async")
    k._check(k.L.cxk_select_mu_async(k.h, 0.8, 1.0, 20 * K, 0.3, 1e-8, 1e9), "cxk_select_mu_async")
    k._check(k.L.cxk_newton_direction_device_mu(k.h, 0.9, 0.8), "cxk_newton_direction_device_mu")
    info, took, inv = np.zeros(2), C.c_int(0), C.c_double(0)
    k._check(k.L.cxk_prepare_take_step_device_mu(k.h, 0.8, 1.0, ol.dp(info), C.byref(took), C.byref(inv)),
             "cxk_prepare_take_step_device_mu")
    ok = C.c_int(1)
    k._check(k.L.cxk_factor_status(k.h, C.byref(ok)), "cxk_factor_status")
    assert ok.value == 0 and took.value == 1
    for i in range(K):
        assert np.array_equal(np.asarray(k.get_W(i)).ravel(), W[i].ravel())   # (symmetric: either layout)


def _problem(kind):
    if kind == "soc":
        return syn.soc_problem(K=60, dim=6, m=5, overlap=2, tree=4), "soc", 120
    if kind == "mixed":
        return syn.mixed_problem(K=46, herm_every=(4, 7), branching=4, overlap=3), "mixed", 300
    if kind == "lmi12":
        return syn.lmi_problem(K=30, n=12, m=6, branching=3, overlap=2, seed=4), "lmi", 360
    if kind == "lp":
        return syn.lp_problem(rows=30, num_vars=8), "lp", 30
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["soc", "mixed", "lmi12", "lp"])
def test_device_mu_on_every_cone_type_is_the_host_path_bit_for_bit(kind, monkeypatch):
    """Programs whose constraints do not take the register LMI kernels have no tail workgroup, but the
    selection still rides in the eigenvalue query's reduction launch, the right-hand side and every
    PrepareStep kernel read inv_sqrt_mu from the device (CWeightOf) and TakeStep follows without the
    host: two rounds (the second from the W the first produced) against the host-side sequence."""
    import ctypes as C
    from conex_amd import KktContext
    prob, build_kind, rankK = _problem(kind)
    a = syn.build(KktContext, prob, build_kind, device=0)
    monkeypatch.setenv("CXK_NO_DEVICE_MU", "1")
    b = syn.build(KktContext, prob, build_kind, device=0)
    monkeypatch.delenv("CXK_NO_DEVICE_MU")
    assert a.L.cxk_device_mu_supported(a.h) == 1 and b.L.cxk_device_mu_supported(b.h) == 0
    cost = prob["b"]
    bs, cs, prev = 0.9, 0.8, 0.0
    for k in (a, b):
        k.set_cost(cost)
    for rnd in range(2):
        for k in (a, b):
            k.assemble()
            k.factor_solve_async(-bs, cs, 0.0)
        e4 = b.weighted_slack_eigenvalues(None, cs)
        inv_host = _host_rule(1.0, rankK, e4, prev, 1e-8, 1e9)
        b._check(b.L.cxk_newton_direction(b.h, inv_host, bs, cs), "cxk_newton_direction")
        assert b.L.cxk_step_scalars_async(b.h) == 0
        n2b, ninfb, tookb = b.prepare_take_step(None, inv_host * cs)
        scb = b.step_scalars()
        a._check(a.L.cxk_select_mu_async(a.h, cs, 1.0, rankK, prev, 1e-8, 1e9), "cxk_select_mu_async")
        a._check(a.L.cxk_newton_direction_device_mu(a.h, bs, cs), "cxk_newton_direction_device_mu")
        assert a.L.cxk_step_scalars_async(a.h) == 0
        info, took, inv_dev = np.zeros(2), C.c_int(0), C.c_double(0)
        a._check(a.L.cxk_prepare_take_step_device_mu(a.h, cs, 1.0, ol.dp(info), C.byref(took), C.byref(inv_dev)),
                 "cxk_prepare_take_step_device_mu")
        sca = a.step_scalars()
        assert inv_dev.value == inv_host, (rnd, inv_dev.value, inv_host)
        assert took.value == 1 and tookb
        assert info[0] == n2b and info[1] == ninfb
        assert np.array_equal(sca, scb)
        assert np.array_equal(a.get_y(), b.get_y())
        prev = inv_host
    for i in range(len(a.cons)):
        assert np.array_equal(np.asarray(a.get_W(i)), np.asarray(b.get_W(i)))


def test_device_mu_is_not_offered_where_the_host_is_needed():
    """An LMI beyond LDS takes its step length from the host (its exponential is a chain of GEMM
    launches), equality rows are solved through the host-paced LDLT: the loop keeps the host round trip."""
    from conex_amd import KktContext
    big = syn.lmi_problem(K=1, n=80, m=6, branching=2, overlap=1, seed=2)
    k = syn.build(KktContext, big, "lmi", device=0)
    assert k.L.cxk_device_mu_supported(k.h) == 0


def test_conex_maximize_is_the_same_solve_with_mu_on_the_device_and_on_the_host(monkeypatch):
    """CONEX_Maximize on a C4-shaped program three ways -- the one-round-trip iteration with the Newton
    direction solved for as the reference does (CXK_NO_TRIPLE=1: a second sweep over the tree), the host-side
    selection (CXK_NO_DEVICE_MU=1) and the separate reduction launches as well (CXK_NO_STEP_TAIL=1): the same
    iterate, the same mu at every iteration, bit for bit.  (The default forms the direction from the three
    solutions of ONE sweep -- equal to rounding, tests/test_gpu_triple.py.)"""
    import ctypes as C
    from conex_amd import capi as ca
    prob = syn.lmi_problem(K=120, n=20, m=20, branching=4, overlap=5, seed=41)
    L = ca.api()
    b = np.ascontiguousarray(prob["b"], dtype=np.float64)

    def solve():
        p = L.CONEX_CreateConeProgram()
        assert L.CONEX_SetNumberOfVariables(p, prob["num_vars"]) == 0
        for c, cl in enumerate(prob["cliques"]):
            a, cm = ca.colmajor(prob["A"][c]), ca.colmajor(prob["C"][c])
            v = np.ascontiguousarray(cl, dtype=np.int64)
            assert L.CONEX_AddSparseLMIConstraint(p, ca.dp(a), 20, 20, len(cl), ca.dp(cm), 20, 20,
                                                  v.ctypes.data_as(C.POINTER(C.c_long)), len(cl)) == c
        cfg = ca.default_config()
        y = np.zeros(len(b))
        ok = L.CONEX_Maximize(p, ca.dp(b), len(b), C.byref(cfg), ca.dp(y), len(b))
        st = ca.IterationStats()
        L.CONEX_GetIterationStats(p, C.byref(st), -1)
        n_it = st.iteration_number + 1
        mus = []
        for i in range(n_it):
            L.CONEX_GetIterationStats(p, C.byref(st), i)
            mus.append(st.mu)
        L.CONEX_DeleteConeProgram(p)
        return ok, y, np.array(mus)

    monkeypatch.setenv("CXK_NO_TRIPLE", "1")
    ok0, y0, mu0 = solve()
    monkeypatch.setenv("CXK_NO_DEVICE_MU", "1")
    ok1, y1, mu1 = solve()
    monkeypatch.setenv("CXK_NO_STEP_TAIL", "1")
    ok2, y2, mu2 = solve()
    assert ok0 == ok1 == ok2 == 1 and len(mu0) > 5
    assert np.array_equal(mu0, mu1) and np.array_equal(mu0, mu2)
    assert np.array_equal(y0, y1) and np.array_equal(y0, y2)


@pytest.mark.parametrize("n,m", [(3, 2), (4, 3), (5, 5), (8, 8), (9, 12), (12, 12), (15, 7), (16, 20), (18, 9), (19, 20),
                                 (20, 20)])
def test_register_prepare_kernels_at_orders_up_to_20_match_the_workgroup_kernels_bit_for_bit(n, m, monkeypatch):
    """lmi_prepare_rows<.., 20, EXACT = false> takes any order 3 <= n <= 20 at run time (matrices n apart,
    lanes and columns beyond n holding zeros): PrepareStep, the eigenvalue query and the W that
    TakeStep then produces must equal the workgroup kernels' (CXK_PREPARE_LDS=1 at context creation)
    bit for bit -- zero factors leave every sum as it is -- and the oracle's to rounding."""
    from conex_amd import KktContext
    K = 24
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=3, overlap=min(2, m - 1), seed=100 + n)
    a = syn.build(KktContext, prob, "lmi", device=0)
    monkeypatch.setenv("CXK_PREPARE_LDS", "1")
    b = syn.build(KktContext, prob, "lmi", device=0)
    monkeypatch.delenv("CXK_PREPARE_LDS")
    o = syn.build(ol.Program, prob, "lmi")
    W = syn.scaling_points(K, n, seed=11)
    for p in (a, b, o):
        for i in range(K):
            p.set_W(i, W[i])
    oko, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert oko == 1
    for it in range(2):
        ea, eb = a.weighted_slack_eigenvalues(yo, 0.8), b.weighted_slack_eigenvalues(yo, 0.8)
        assert np.array_equal(ea, eb), (it, ea, eb)
        assert np.allclose(ea, o.weighted_slack_eigenvalues(yo, 0.8), rtol=1e-8, atol=1e-10)
        ia, ib = a.prepare_step(yo, 0.8, 1.0), b.prepare_step(yo, 0.8, 1.0)
        assert np.array_equal(ia, ib), (it, ia, ib)
        assert np.allclose(ia, o.prepare_step(yo, 0.8, 1.0), rtol=1e-8)
        step = min(1.0, 2.0 / ia[1] ** 2)
        for p in (a, b, o):
            p.take_step(step, 1.0)
        for i in (0, K - 1):
            assert np.array_equal(np.asarray(a.get_W(i)), np.asarray(b.get_W(i)))
            assert np.allclose(np.asarray(a.get_W(i)).ravel(), np.asarray(o.get_W(i)).ravel(), rtol=1e-8, atol=1e-10)
