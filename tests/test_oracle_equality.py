"""Pins the oracle's equality-constraint / LDLT path (8f item 1):

  RLDLT.h:298-431                        P A P^T = L D L^T on symmetric indefinite input, +-1e-9 clamp
  block_triangular_operations_test.cc    LDLT factor + solve reproduces the dense solve (<= 1e-12 there)
  equality_constraints_test.cc:11-52     Basic: A_eq y = b_eq and y = optimal_y to 1e-5
  equality_constraints_test.cc:54-134    Many / ManySeparate: feasibility 5e-7, objective bound
The reference draws its data with libc rand(); the instances here are seeded numpy.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol


def rldlt(A):
    L = ol.lib()
    n = A.shape[0]
    a = ol.colmajor(A).copy()
    tr = np.zeros(max(n, 1), dtype=np.int32)
    L.cxo_rldlt_inplace.restype = C.c_int
    L.cxo_rldlt_inplace.argtypes = [C.c_int, ol.c_double_p, C.c_int, ol.c_int_p]
    ok = L.cxo_rldlt_inplace(n, ol.dp(a), n, ol.ip(tr))
    return ok, a.reshape(n, n).T, tr


@pytest.mark.parametrize("n", [1, 2, 7, 20])
def test_rldlt_reconstructs_indefinite_matrix(n):
    rng = np.random.default_rng(n)
    A = rng.uniform(-1, 1, (n, n))
    A = A + A.T                                   # symmetric, indefinite
    ok, F, tr = rldlt(A)
    assert ok == 1
    Lm = np.tril(F, -1) + np.eye(n)
    D = np.diag(np.diag(F))
    P = np.eye(n)
    for k in range(n):                            # transpositions applied in order
        P[[k, tr[k]]] = P[[tr[k], k]]
    assert np.allclose(P @ A @ P.T, Lm @ D @ Lm.T, atol=1e-12)


def test_rldlt_clamps_zero_pivots():
    A = np.zeros((3, 3))
    A[0, 0] = 2.0
    ok, F, _ = rldlt(A)
    assert ok == 0                                # regularisation used
    assert F[0, 0] == 2.0 and F[1, 1] == 1e-9 and F[2, 2] == 1e-9
    ok, F, _ = rldlt(np.array([[-1e-12]]))        # size <= 1 branch clamps but reports success
    assert ok == 1 and F[0, 0] == -1e-9


def regularized(p):
    L = ol.lib()
    L.cxo_factor_regularized.restype = C.c_int
    L.cxo_factor_regularized.argtypes = [C.c_void_p]
    return L.cxo_factor_regularized(p.h)


def eq_lp_program(seed, num_vars=6, rows=9, splits=()):
    rng = np.random.default_rng(seed)
    p = ol.Program(num_vars)
    A = rng.uniform(-1, 1, (rows, num_vars))
    p.add_linear(A, np.abs(rng.uniform(0.5, 1.5, rows)))
    for vars_ in splits:
        p.add_equality(rng.uniform(-1, 1, (1, len(vars_))), rng.uniform(-1, 1, 1), list(vars_))
    p.add_equality(rng.uniform(-1, 1, (2, num_vars)), rng.uniform(-1, 1, 2))
    p.initialize()
    return p


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_block_ldlt_solve_matches_dense_solve(seed):
    p = eq_lp_program(seed)
    assert p.N == 6 + 2
    p.assemble()
    K = p.kkt_matrix()
    assert np.allclose(K, K.T)
    assert np.linalg.eigvalsh(K).min() < 0 < np.linalg.eigvalsh(K).max()   # saddle-point system
    assert p.factor() == 1
    rhs = np.random.default_rng(seed + 10).uniform(-1, 1, p.N)
    x = p.solve_inplace(rhs)
    assert regularized(p) == 0
    assert np.linalg.norm(K @ x - rhs) <= 1e-11 * np.linalg.norm(rhs)


def test_lone_multiplier_supernode_is_clamped():
    """An equality whose variables all sit in other cliques leaves a 1 x 1 zero diagonal block for
    its multiplier: RLDLT clamps it to 1e-9 and the solver records it (kkt_solver.cc:190-192)."""
    p = eq_lp_program(4, splits=((0, 1, 2), (2, 3, 5)))
    assert p.N == 6 + 1 + 1 + 2
    p.assemble()
    K = p.kkt_matrix()
    assert p.factor() == 1
    assert regularized(p) == 0        # the size <= 1 branch of RLDLT clamps silently (RLDLT.h:311-330)
    rhs = np.ones(p.N)
    x = p.solve_inplace(rhs)
    res = np.linalg.norm(K @ x - rhs) / np.linalg.norm(rhs)
    assert np.all(np.isfinite(x)) and 1e-12 < res < 1e-3      # solves the 1e-9-perturbed system


def test_equality_basic():
    rng = np.random.default_rng(3)
    nv, neq, nin = 3, 1, 4
    A = rng.uniform(-1, 1, (nin, nv))
    slack = np.array([0, 0, 1.0, 1.0])
    dual = np.array([1.0, 1.0, 0, 0])
    y_opt = rng.uniform(-1, 1, nv)
    Cc = slack + A @ y_opt
    Aeq = rng.uniform(-1, 1, (neq, nv))
    beq = Aeq @ y_opt
    p = ol.Program(nv)
    p.add_equality(Aeq, beq)
    p.add_linear(A, Cc)
    ok, y = p.solve(A.T @ dual)
    assert np.linalg.norm(Aeq @ y - beq) <= 1e-5
    assert np.linalg.norm(y - y_opt) <= 1e-5


@pytest.mark.parametrize("separate", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_equality_many(separate, seed):
    rng = np.random.default_rng(100 + seed)
    nv = 10
    nin, neq = nv + 10, nv - 2
    A = rng.uniform(-1, 1, (nin, nv))
    m = nin // 2
    slack = np.ones(nin)
    dual = np.ones(nin)
    slack[:m] = 1e-7
    dual[m:] = 1e-7
    y_opt = rng.uniform(-1, 1, nv)
    Cc = slack + A @ y_opt
    p = ol.Program(nv)
    p.add_linear(A, Cc)
    eq = np.zeros((neq, nv))
    Bi = np.array([[1.0, 2.0, 3.0]])
    for i in range(neq):
        vars_ = [0, i + 1, nv - 1]
        eq[i, vars_] = Bi[0]
        if separate:
            p.add_equality(Bi, eq[i] @ y_opt, vars_)
    if not separate:
        p.add_equality(eq, eq @ y_opt)
    cfg = ol.default_config()
    cfg.final_centering_steps = 10
    cfg.initial_centering_steps_coldstart = 0
    cfg.max_iterations = 40
    cfg.divergence_upper_bound = .5
    cost = A.T @ dual
    ok, y = p.solve(cost, cfg)
    assert np.linalg.norm(eq @ y - eq @ y_opt) <= 5e-7
    assert cost @ y + 1e-4 >= cost @ y_opt
