"""The factorization with three right-hand sides in its one whole-tree launch (tree_fused.h kFusedTriple,
cxk_factor_solve_triple_async): the solve of the mu selection and -- by linearity -- the Newton direction
for the mu selected afterwards come out of ONE sweep over the tree (cone_program.cc:181, :409-411), the
interior-point iteration is five launches instead of seven (the combination itself rides in PrepareStep's).

Twin contexts on the same program: one runs the reference's sequence call for call (factorization with the
mu selection's right-hand side, eigenvalue query, a second sweep with the Newton right-hand side,
PrepareStep, TakeStep), the other the triple sequence.  The direction agrees to rounding (a combination of
three solutions against one solve of the combined right-hand side: the tolerance is stated), everything
downstream to the tolerances of the parity tests.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import KktContext, synthetic as syn

pytestmark = pytest.mark.gpu


def _pair(prob):
    a = syn.build(KktContext, prob, "lmi", device=0)
    b = syn.build(KktContext, prob, "lmi", device=0)
    return a, b


@pytest.mark.parametrize("K,branching,dub,prev", [(48, 4, 1.0, 0.0), (200, 8, 1.0, 0.3), (48, 4, 50.0, 0.3), (9, 2, 1e-9, 0.3)])
def test_triple_sequence_matches_the_two_sweep_sequence(K, branching, dub, prev):
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=branching, overlap=5, seed=31 + K)
    a, b = _pair(prob)
    W = syn.scaling_points(K, 20, seed=8)
    bs, cs, rankK, lb, ub = 0.9, 0.8, 20 * K, 1e-8, 1e9
    for k in (a, b):
        for i in range(K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
    for rep in range(3):    # (three iterations: the two sets of the extra hand-off slots both get used and re-armed)
        for k in (a, b):
            k.assemble()
        assert a.L.cxk_triple_supported(a.h) == 1
        a._check(a.L.cxk_factor_solve_triple_async(a.h, bs, cs), "cxk_factor_solve_triple_async")
        b.factor_solve_async(-bs, cs, 0.0)
        ya, yb = a.get_y(), b.get_y()
        scale = np.linalg.norm(yb)
        assert np.linalg.norm(ya - yb) <= 1e-12 * scale, np.linalg.norm(ya - yb) / scale   # K^-1(-bs b) + K^-1(cs AQc)
        for k in (a, b):
            k._check(k.L.cxk_select_mu_async(k.h, cs, dub, rankK, prev, lb, ub), "cxk_select_mu_async")
            k._check(k.L.cxk_newton_direction_device_mu(k.h, bs, cs), "cxk_newton_direction_device_mu")
            assert k.L.cxk_step_scalars_async(k.h) == 0
        res = []
        for k in (a, b):
            info, took, inv = np.zeros(2), C.c_int(0), C.c_double(0)
            k._check(k.L.cxk_prepare_take_step_device_mu(k.h, cs, 1.0, ol.dp(info), C.byref(took), C.byref(inv)),
                     "cxk_prepare_take_step_device_mu")
            res.append((info.copy(), took.value, inv.value, k.step_scalars(), k.get_y()))
        (ia, ta, va, sa, ya), (ib, tb, vb, sb, yb) = res
        assert ta == tb == 1
        # the selected mu: the eigenvalue query saw directions that differ by rounding
        assert abs(va - vb) <= 1e-9 * abs(vb), (va, vb)
        scale = np.linalg.norm(yb)
        assert np.linalg.norm(ya - yb) <= 1e-9 * scale, np.linalg.norm(ya - yb) / scale
        assert np.allclose(ia, ib, rtol=1e-8, atol=1e-12), (ia, ib)
        assert np.allclose(sa, sb, rtol=1e-8, atol=1e-12), (sa, sb)
        for i in (0, K // 2, K - 1):
            Wa, Wb = np.asarray(a.get_W(i)), np.asarray(b.get_W(i))
            assert np.linalg.norm(Wa - Wb) <= 1e-8 * np.linalg.norm(Wb)


def test_direction_from_three_solutions_against_the_oracle():
    """The Newton direction the PrepareStep leaves in y behind the triple launch, against a solve of the
    combined right-hand side for the same mu (itself held against the CPU oracle by tests/test_gpu_parity.py;
    north_star: <= 1e-10)."""
    K = 64
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=8, overlap=5, seed=77)
    a = syn.build(KktContext, prob, "lmi", device=0)
    W = syn.scaling_points(K, 20, seed=9)
    for i in range(K):
        a.set_W(i, W[i])
    a.set_cost(prob["b"])
    a.assemble()
    bs, cs = 0.9, 0.8
    a._check(a.L.cxk_factor_solve_triple_async(a.h, bs, cs), "cxk_factor_solve_triple_async")
    a._check(a.L.cxk_select_mu_async(a.h, cs, 1.0, 20 * K, 0.3, 1e-8, 1e9), "cxk_select_mu_async")
    a._check(a.L.cxk_newton_direction_device_mu(a.h, bs, cs), "cxk_newton_direction_device_mu")
    assert a.L.cxk_step_scalars_async(a.h) == 0
    info, took, inv = np.zeros(2), C.c_int(0), C.c_double(0)
    a._check(a.L.cxk_prepare_take_step_device_mu(a.h, cs, 1.0, ol.dp(info), C.byref(took), C.byref(inv)),
             "cxk_prepare_take_step_device_mu")
    y = a.get_y()
    # the same direction from a context that solves the combined right-hand side, mu given
    b = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(K):
        b.set_W(i, W[i])
    b.set_cost(prob["b"])
    ok, yb = b.kkt_solve(prob["b"], inv.value, bs, cs)
    assert ok
    assert np.linalg.norm(y - yb) <= 1e-10 * np.linalg.norm(yb)


def test_direction_asked_for_before_prepare_step_is_materialised():
    """Behind the triple launch the direction is normally formed inside PrepareStep; whoever reads y first gets
    it from one elementwise launch instead (FlushDeferred) -- the same values."""
    K = 40
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=4, overlap=5, seed=5)
    a, b = _pair(prob)
    W = syn.scaling_points(K, 20, seed=10)
    bs, cs = 0.9, 0.8
    for k in (a, b):
        for i in range(K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
        k.assemble()
    a._check(a.L.cxk_factor_solve_triple_async(a.h, bs, cs), "cxk_factor_solve_triple_async")
    b.factor_solve_async(-bs, cs, 0.0)
    for k in (a, b):
        k._check(k.L.cxk_select_mu_async(k.h, cs, 1.0, 20 * K, 0.3, 1e-8, 1e9), "cxk_select_mu_async")
        k._check(k.L.cxk_newton_direction_device_mu(k.h, bs, cs), "cxk_newton_direction_device_mu")
    ya, yb = a.get_y(), b.get_y()          # (a: materialised here)
    assert np.linalg.norm(ya - yb) <= 1e-9 * np.linalg.norm(yb)
    # ... and PrepareStep then takes y as it finds it
    res = []
    for k in (a, b):
        assert k.L.cxk_step_scalars_async(k.h) == 0
        info, took, inv = np.zeros(2), C.c_int(0), C.c_double(0)
        k._check(k.L.cxk_prepare_take_step_device_mu(k.h, cs, 1.0, ol.dp(info), C.byref(took), C.byref(inv)),
                 "cxk_prepare_take_step_device_mu")
        res.append((info.copy(), k.step_scalars()))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-8, atol=1e-12)
    assert np.allclose(res[0][1], res[1][1], rtol=1e-8, atol=1e-12)


def test_direction_inside_prepare_step_equals_the_launch_of_its_own_bit_for_bit(monkeypatch):
    """Behind the triple launch the direction is combined INSIDE the PrepareStep launch (the constraints' wavefronts
    form the entries they read, extra workgroups write y out, the tail workgroup waits for them before it forms the
    step scalars).  CXK_NO_Y_DEFERRAL=1 keeps newton_from_three as a launch of its own: the same expression, so the
    direction, the step scalars, PrepareStep's norms and the W that TakeStep leaves are the same bits."""
    K = 96
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=4, overlap=5, seed=12)
    a = syn.build(KktContext, prob, "lmi", device=0)
    monkeypatch.setenv("CXK_NO_Y_DEFERRAL", "1")
    b = syn.build(KktContext, prob, "lmi", device=0)
    monkeypatch.delenv("CXK_NO_Y_DEFERRAL")
    W = syn.scaling_points(K, 20, seed=11)
    bs, cs = 0.9, 0.8
    for k in (a, b):
        for i in range(K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
    for rep in range(3):
        res = []
        for k in (a, b):
            k.assemble()
            assert k.L.cxk_triple_supported(k.h) == 1
            k._check(k.L.cxk_factor_solve_triple_async(k.h, bs, cs), "cxk_factor_solve_triple_async")
            k._check(k.L.cxk_select_mu_async(k.h, cs, 1.0, 20 * K, 0.3 if rep else 0.0, 1e-8, 1e9), "cxk_select_mu_async")
            k._check(k.L.cxk_newton_direction_device_mu(k.h, bs, cs), "cxk_newton_direction_device_mu")
            assert k.L.cxk_step_scalars_async(k.h) == 0
            info, took, inv = np.zeros(2), C.c_int(0), C.c_double(0)
            k._check(k.L.cxk_prepare_take_step_device_mu(k.h, cs, 1.0, ol.dp(info), C.byref(took), C.byref(inv)),
                     "cxk_prepare_take_step_device_mu")
            res.append((info.copy(), took.value, inv.value, np.asarray(k.step_scalars()), k.get_y(),
                        [np.asarray(k.get_W(i)) for i in (0, K // 2, K - 1)]))
        (ia, ta, va, sa, ya, Wa), (ib, tb, vb, sb, yb, Wb) = res
        assert ta == tb == 1 and va == vb
        assert np.array_equal(ya, yb) and np.array_equal(ia, ib) and np.array_equal(sa, sb)
        for x, y in zip(Wa, Wb):
            assert np.array_equal(x, y)


def test_not_offered_where_it_does_not_apply():
    prob, kind = syn.soc_problem(K=40, dim=6, m=5, overlap=2, tree=4), "soc"
    k = syn.build(KktContext, prob, kind, device=0)
    k.set_cost(prob["b"])
    k.assemble()
    assert k.L.cxk_triple_supported(k.h) == 0
    assert k.L.cxk_factor_solve_triple_async(k.h, 0.9, 0.8) != 0
