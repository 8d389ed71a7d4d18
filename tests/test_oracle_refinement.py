"""Iterative refinement of SupernodalKKTSolver::SolveInPlace (kkt_solver.cc:233-261): with
SetIterativeRefinementIterations(n), Factor keeps kkt_matrix_ = KKTMatrix() (:177-179) and every
solve runs  y <- y + K^-1 (b - kkt_matrix_ y)  n times.  The oracle's restatement against the same
loop written with numpy on the oracle's own primitives (dense KKT matrix, unrefined solve)."""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import synthetic as syn
from test_oracle_equality import eq_lp_program


def numpy_refined(p_plain, K, rhs, iters):
    y = p_plain.solve_inplace(rhs)
    for _ in range(iters):
        y = y + p_plain.solve_inplace(rhs - K @ y)
    return y


def lmi_tree(seed):
    prob = syn.lmi_problem(K=9, n=5, m=6, branching=3, overlap=2, seed=seed)
    W = syn.scaling_points(9, 5, seed=seed + 1)

    def make():
        p = syn.build(ol.Program, prob, "lmi")
        for i in range(9):
            p.set_W(i, W[i])
        return p
    return make


@pytest.mark.parametrize("iters", [1, 3])
def test_cholesky_refinement_is_the_reference_loop(iters):
    make = lmi_tree(5)
    plain, refined = make(), make()
    refined.set_refinement(iters)
    for p in (plain, refined):
        p.assemble()
    K = plain.kkt_matrix()
    assert plain.factor() == 1 and refined.factor() == 1
    rhs = np.random.default_rng(3).uniform(-1, 1, plain.N)
    want = numpy_refined(plain, K, rhs, iters)
    got = refined.solve_inplace(rhs)
    assert np.linalg.norm(got - want) <= 1e-14 * np.linalg.norm(want)
    assert np.linalg.norm(K @ got - rhs) <= 1e-13 * np.linalg.norm(rhs)


def test_refinement_repairs_a_clamped_ldlt_pivot():
    """The lone-multiplier program (test_oracle_equality): RLDLT clamps a zero pivot to 1e-9, the
    factorization is that of a perturbed matrix; refinement against the assembled matrix brings the
    residual down by orders of magnitude per step."""
    plain = eq_lp_program(4, splits=((0, 1, 2), (2, 3, 5)))
    refined = eq_lp_program(4, splits=((0, 1, 2), (2, 3, 5)))
    refined.set_refinement(2)
    for p in (plain, refined):
        p.assemble()
    K = plain.kkt_matrix()
    assert plain.factor() == 1 and refined.factor() == 1
    rhs = np.ones(plain.N)
    x0 = plain.solve_inplace(rhs)
    x2 = refined.solve_inplace(rhs)
    r0 = np.linalg.norm(K @ x0 - rhs) / np.linalg.norm(rhs)
    r2 = np.linalg.norm(K @ x2 - rhs) / np.linalg.norm(rhs)
    assert r0 > 1e-12 and r2 < 1e-3 * r0
    want = numpy_refined(plain, K, rhs, 2)
    assert np.linalg.norm(x2 - want) <= 1e-12 * np.linalg.norm(want)


def test_refinement_off_by_default_and_switchable():
    make = lmi_tree(7)
    a, b = make(), make()
    for p in (a, b):
        p.assemble()
    b.set_refinement(2)
    b.set_refinement(0)
    assert a.factor() == 1 and b.factor() == 1
    rhs = np.random.default_rng(1).uniform(-1, 1, a.N)
    assert np.array_equal(a.solve_inplace(rhs), b.solve_inplace(rhs))
