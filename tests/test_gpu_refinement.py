"""cxk_set_iterative_refinement: the refinement loop of SupernodalKKTSolver::SolveInPlace
(kkt_solver.cc:233-261) on the device -- K y from the supernodal blocks of the assembled matrix the
factor sweep keeps, instead of the reference's dense N x N copy -- against the oracle's restatement."""
import numpy as np
import pytest

import conex_api as capi
import oracle_lib as ol
from conex_amd import KktContext
from conex_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.mark.parametrize("K,n,m,iters", [(40, 6, 8, 1), (40, 6, 8, 3), (12, 20, 20, 2)])
def test_refined_newton_direction_matches_oracle(K, n, m, iters):
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=3, overlap=2, seed=21)
    W = syn.scaling_points(K, n, seed=22)
    o = syn.build(ol.Program, prob, "lmi")
    k = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(K):
        o.set_W(i, W[i])
        k.set_W(i, W[i])
    o.set_refinement(iters)
    k.set_refinement(iters)
    ok_o, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    ok_k, yk = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok_o == 1 and ok_k == 1
    assert rel(yk, yo) <= 1e-10
    # solve-only path on the stored factor (mu selection, line search)
    rhs = np.random.default_rng(5).uniform(-1, 1, k.N)
    assert rel(k.solve_inplace(rhs), o.solve_inplace(rhs)) <= 1e-10
    # switching it off restores the plain solve
    k.set_refinement(0)
    o.set_refinement(0)
    assert rel(k.solve_inplace(rhs), o.solve_inplace(rhs)) <= 1e-10


def clamped_program(cls, **kw):
    """test_oracle_equality.eq_lp_program(4, splits=...): two lone multipliers get clamped pivots."""
    rng = np.random.default_rng(4)
    p = cls(6, **kw)
    A = rng.uniform(-1, 1, (9, 6))
    p.add_linear(A, np.abs(rng.uniform(0.5, 1.5, 9)))
    for vars_ in ((0, 1, 2), (2, 3, 5)):
        p.add_equality(rng.uniform(-1, 1, (1, len(vars_))), rng.uniform(-1, 1, 1), list(vars_))
    p.add_equality(rng.uniform(-1, 1, (2, 6)), rng.uniform(-1, 1, 2))
    p.initialize()
    return p


def test_refinement_repairs_clamped_ldlt_pivot_on_the_device():
    o = clamped_program(ol.Program)
    k0 = clamped_program(KktContext, device=0)
    k2 = clamped_program(KktContext, device=0)
    o.assemble()
    Kmat = o.kkt_matrix()
    o.set_refinement(2)
    k2.set_refinement(2)
    assert o.factor() == 1
    for k in (k0, k2):
        k.assemble()
        assert k.factor() == 1
    rhs = np.ones(o.N)
    x0, x2, xo = k0.solve_inplace(rhs), k2.solve_inplace(rhs), o.solve_inplace(rhs)
    r0 = np.linalg.norm(Kmat @ x0 - rhs) / np.linalg.norm(rhs)
    r2 = np.linalg.norm(Kmat @ x2 - rhs) / np.linalg.norm(rhs)
    assert r0 > 1e-12 and r2 < 1e-3 * r0
    assert rel(x2, xo) <= 1e-9


def test_refinement_option_through_conex_h():
    """SolverConfiguration.iterative_refinement_iterations reaches the device solver
    (cone_program.cc:303-304): same optimum, solved status."""
    import ctypes as C
    L = capi.api()
    prob = syn.lp_problem(rows=20, num_vars=10)
    A, c, b = prob["A"], prob["c"], prob["b"]
    sols = []
    for iters in (0, 2):
        p = L.CONEX_CreateConeProgram()
        Af, cf = capi.colmajor(A), capi.colmajor(c)
        assert L.CONEX_AddDenseLinearConstraint(p, capi.dp(Af), A.shape[0], A.shape[1], capi.dp(cf), len(c)) >= 0
        cfg = capi.default_config()
        cfg.iterative_refinement_iterations = iters
        y = np.zeros(A.shape[1])
        bb = np.ascontiguousarray(b, dtype=np.float64)
        assert L.CONEX_Maximize(p, capi.dp(bb), len(bb), C.byref(cfg), capi.dp(y), len(y)) == 1
        sols.append(y)
        L.CONEX_DeleteConeProgram(p)
    assert rel(sols[1], sols[0]) <= 1e-8
