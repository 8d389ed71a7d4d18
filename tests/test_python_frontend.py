"""conex_amd.program.Conex -- the Python-3 front end with the surface of the reference's
interfaces/python/ConexProgram.py -- exercised the way interfaces/python/test/run_tests.py
exercises the SWIG wrapper: random mixed programs must come back solved with small optimality
errors (|Ax - b| and <x, s> < 1e-5, run_tests.py:40-43), sparse LMIs solve, mu never increases,
builder errors raise.  Data are seeded (the reference's are numpy's global RNG)."""
import numpy as np
import pytest


def randsym(rng, d):
    A = rng.standard_normal((d, d))
    return 0.5 * (A + A.T)


def test_surface_matches_reference_wrapper():
    """Every public method of ConexProgram.Conex (ConexProgram.py:58-277) exists."""
    from conex_amd.program import Conex, Errors, LMIOperator, Solution  # noqa: F401
    for name in ["GetIterationStats", "GetIterationNumberStats", "AddQuadraticCost", "AddLinearInequality",
                 "AddLinearInequalities", "DefaultConfiguration", "Solve", "Maximize", "GetDualVariables",
                 "NewLinearMatrixInequality", "NewLorentzConeConstraint", "NewLinearInequality",
                 "NewQuadraticCost", "UpdateQuadraticCostMatrix", "UpdateLinearOperator", "UpdateAffineTerm",
                 "AddDenseLinearMatrixInequality", "AddSparseLinearMatrixInequality", "ComputeErrors"]:
        assert callable(getattr(Conex, name))


def test_lmi_operator_and_adjoint():
    from conex_amd.program import LMIOperator
    rng = np.random.default_rng(0)
    A = np.stack([randsym(rng, 4) for _ in range(3)], axis=2)
    op = LMIOperator(A, 5, [4, 0, 2])
    y, X = rng.standard_normal(5), randsym(rng, 4)
    assert np.isclose(np.trace(op.apply(y) @ X), op.adjoint(X) @ y)     # <A y, X> = <y, A* X>


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_mixed_instance_is_solved_to_optimality(seed):
    """run_tests.py:62-89: two linear blocks and a dense LMI; the cost is chosen feasible."""
    from conex_amd.program import Conex
    rng = np.random.default_rng(seed)
    m, n = 2, 4
    prog = Conex(m)
    A1 = np.array([[1.0, 3.0], [4.0, 1.0], [1.0, 1.0]])
    c1 = np.ones(3)
    b = A1.T @ c1
    prog.AddLinearInequality(A1, c1)
    prog.AddLinearInequality(A1.copy(), c1.copy())
    Amat = np.stack([randsym(rng, n) for _ in range(m)], axis=2)
    Amat[:, :, m - 1] = 0
    Amat[0, 0, m - 1] = 1
    prog.AddDenseLinearMatrixInequality(Amat, np.eye(n))
    sol = prog.Maximize(b)
    assert sol.status == 1
    sol.x = prog.GetDualVariables()
    sol.s, sol.err = prog.ComputeErrors(sol.y, sol.x, b)
    assert sol.err.Ax_minus_b < 1e-5 and abs(sol.err.x_dot_s) < 1e-5
    assert min(sol.err.min_eig_S) > -1e-6 and min(sol.err.min_eig_X) > -1e-6


@pytest.mark.gpu
def test_sparse_lmis_over_overlapping_variables():
    """run_tests.py:91-112."""
    from conex_amd.program import Conex, ConexError
    rng = np.random.default_rng(7)
    prog = Conex(3)
    n = 4
    for variables in (np.arange(0, 2), np.arange(1, 3)):
        A = np.stack([randsym(rng, n) for _ in range(2)], axis=2)
        prog.AddSparseLinearMatrixInequality(A, np.eye(n), variables)
    with pytest.raises(ConexError):
        prog.AddSparseLinearMatrixInequality(A, np.eye(n), np.array([2, 3]))   # variable 3 of 3
    sol = prog.Maximize(np.ones(3))
    assert sol.status == 1
    x = prog.GetDualVariables()
    _, err = prog.ComputeErrors(sol.y, x, np.ones(3))
    assert err.Ax_minus_b < 1e-5 and abs(err.x_dot_s) < 1e-5


@pytest.mark.gpu
def test_mu_is_non_increasing_and_iterations_are_capped():
    """run_tests.py:249-281."""
    from conex_amd.program import Conex
    m = 2
    prog = Conex(m)
    A = np.vstack([np.eye(m), np.eye(m)])
    prog.AddLinearInequality(A, -np.ones(2 * m))
    cfg = prog.DefaultConfiguration()
    cfg.max_iterations = 6
    prog.Maximize(np.ones(m), cfg)
    assert prog.GetIterationNumberStats(-1).iteration_number + 1 <= cfg.max_iterations
    mus = [s.mu for s in prog.GetIterationStats()]
    assert len(mus) >= 1 and all(b <= a for a, b in zip(mus, mus[1:]))


@pytest.mark.gpu
def test_entrywise_builders_lmi_and_lorentz_cone():
    """New* / Update* (run_tests.py:6-34, 283-357): a diagonal LMI y_i <= 1 and the cone |y| <= 1."""
    from conex_amd.program import Conex
    n = 3
    prog = Conex(n)
    lmi = prog.NewLinearMatrixInequality(n, 1)
    for i in range(n):
        prog.UpdateLinearOperator(lmi, 1.0, i, i, i)
        prog.UpdateAffineTerm(lmi, 1.0, i, i)
    sol = prog.Maximize(np.ones(n))
    assert sol.status == 1 and np.allclose(sol.y, 1.0, atol=1e-3)

    prog = Conex(n)
    soc = prog.NewLorentzConeConstraint(n)
    prog.UpdateAffineTerm(soc, 1.0, 0)
    for i in range(n):
        prog.UpdateLinearOperator(soc, 1.0, i, i + 1)
    sol = prog.Maximize(np.ones(n))
    assert sol.status == 1 and np.allclose(sol.y, np.ones(n) / np.sqrt(n), atol=1e-3)


def test_cost_dimension_is_checked_before_the_library_is_called():
    from conex_amd.program import Conex, ConexError
    prog = Conex(3)
    with pytest.raises(ConexError):
        prog.Maximize(np.ones(2))
    with pytest.raises(ConexError):
        prog.AddQuadraticCost(np.eye(2))
