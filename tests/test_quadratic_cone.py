"""QuadraticConstraint (quadratic_cone_constraint.{h,cc}): the Lorentz cone x0 >= sqrt(x1' Q x1).

CPU: the oracle's restatement (oracle/cxo_program.c, quad_*) pinned by the properties the reference's
own test pins (conex/test/test_socp.cc:15-93, data from rand()): the same problem posed as a
second-order cone, as an LMI, as a quadratic cone with Q = Wsqrt' Wsqrt and as a quadratic cone with
the square root in the constraint matrix has the same solution (8e-6 there).
GPU: the HIP kernels (kernels_quad.hip.h) against the oracle stage by stage, alone and mixed with
matrix cones in a clique tree, and the test_socp.cc property through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import conex_api as ca
import oracle_lib as ol
from conex_amd import KktContext
from conex_amd import synthetic as syn

TOL_SCHUR = 1e-13


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    n = np.linalg.norm(b)
    return np.linalg.norm(a - b) / n if n > 0 else np.linalg.norm(a - b)


def socp_formulations(rng, n=3):
    """test_socp.cc:15-54"""
    Wsqrt = rng.uniform(-1, 1, (n, n))
    As = np.zeros((n + 1, n))
    As[1:, :] = Wsqrt
    Cs = np.zeros(n + 1)
    Cs[0] = 1
    Aq = np.zeros((n + 1, n))
    Aq[1:, :] = np.eye(n)
    A_lmi = np.zeros((n, n + 1, n + 1))
    for i in range(n):
        A_lmi[i, 1:, 0] = Wsqrt[:, i]
        A_lmi[i, 0, 1:] = Wsqrt[:, i]
    return Wsqrt, As, Cs, Wsqrt.T @ Wsqrt, Aq, A_lmi


def oracle_cfg():
    cfg = ol.default_config()
    cfg.inv_sqrt_mu_max = 10000
    return cfg


@pytest.mark.parametrize("seed", range(5))
def test_oracle_soc_lmi_and_both_quadratic_forms_agree(seed):
    rng = np.random.default_rng(seed)
    n = 3
    Wsqrt, As, Cs, Q, Aq, A_lmi = socp_formulations(rng, n)
    for i in range(-2, 2):
        b = np.full(n, float(i)) + rng.uniform(-1, 1, n) * .02
        ys = []
        for build in (lambda p: p.add_soc(As, Cs), lambda p: p.add_lmi(A_lmi, np.eye(n + 1)),
                      lambda p: p.add_quadratic(Q, Aq, Cs), lambda p: p.add_quadratic(None, As, Cs)):
            p = ol.Program(n)
            assert build(p) == 0
            p.initialize()
            ok, y = p.solve(b, oracle_cfg())
            ys.append(y)
        assert np.linalg.norm(ys[0] - ys[1]) <= 1e-4        # test_socp.cc:70
        assert np.linalg.norm(ys[0] - ys[2]) <= 8e-6        # :76
        assert np.linalg.norm(ys[0] - ys[3]) <= 8e-6        # :82


def test_oracle_schur_block_is_the_quadratic_representation():
    """A' Q(w) A with Q(w) = 2 w w' - det(w) R in the Q-inner product, <x, y> = 2 x'y (the comment of
    quadratic_cone_constraint.cc:238-239), against dense numpy."""
    rng = np.random.default_rng(3)
    n, m = 5, 4
    R = rng.uniform(-1, 1, (n, n))
    Q = R @ R.T + n * np.eye(n)
    A = rng.uniform(-1, 1, (n + 1, m))
    c = rng.uniform(-1, 1, n + 1)
    w1 = rng.uniform(-.2, .2, n)
    w = np.r_[np.sqrt(w1 @ Q @ w1) + 1.0, w1]
    p = ol.Program(m)
    assert p.add_quadratic(Q, A, c) == 0
    p.initialize()
    p.set_W(0, w)
    p.assemble()
    G, AW, AQc, sc = p.constraint_schur(0)
    J = np.zeros((n + 1, n + 1))                      # the form <x, y>_Q = x0 y0 + x1' Q y1
    J[0, 0] = 1
    J[1:, 1:] = Q
    Rm = np.zeros((n + 1, n + 1))
    Rm[0, 0] = 1
    Rm[1:, 1:] = -Q
    det_w = w[0] ** 2 - w1 @ Q @ w1
    Qw = 2 * np.outer(J @ w, J @ w) - det_w * Rm      # quadratic representation as a bilinear form
    assert rel(np.tril(G), np.tril(2 * A.T @ Qw @ A)) <= 1e-13
    assert rel(AW, 2 * A.T @ J @ w) <= 1e-13
    assert rel(AQc, 2 * A.T @ Qw @ c) <= 1e-13
    assert rel(sc, [2 * (c @ J @ w), 2 * (c @ Qw @ c)]) <= 1e-13


# ------------------------------------------------------------------------------------------- GPU
def quad_tree_problem(K, n, m, seed, with_q=True, lmi_every=0):
    """K cones in a clique tree; every lmi_every-th constraint is a dense LMI instead."""
    rng = np.random.default_rng(seed)
    cliques, num_vars = syn.tree_cliques(K, 3, m, 2)
    cons = []
    b = np.zeros(num_vars)
    for k in range(K):
        if lmi_every and k % lmi_every == 0:
            A = rng.uniform(-1, 1, (m, 6, 6))
            A = A + np.transpose(A, (0, 2, 1))
            cons.append(("lmi", A, np.eye(6)))
            b[cliques[k]] += .5 * np.trace(A, axis1=1, axis2=2)
        else:
            Rq = rng.uniform(-1, 1, (n, n))
            Q = Rq @ Rq.T / n + np.eye(n) if with_q else None
            A = rng.uniform(-1, 1, (n + 1, m))
            c = np.zeros(n + 1)
            c[0] = 1
            cons.append(("quad", Q, A, c))
            b[cliques[k]] += A[0]
    return cons, cliques, num_vars, b


def build_both(cons, cliques, num_vars):
    out = []
    for cls, kw in ((ol.Program, {}), (KktContext, {"device": 0})):
        p = cls(num_vars, **kw)
        for k, cn in enumerate(cons):
            if cn[0] == "lmi":
                assert p.add_lmi(cn[1], cn[2], cliques[k]) == k
            else:
                assert p.add_quadratic(cn[1], cn[2], cn[3], cliques[k]) == k
        p.initialize()
        out.append(p)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("K,n,m,with_q,lmi_every", [(1, 3, 3, True, 0), (12, 5, 4, True, 0), (12, 5, 4, False, 0),
                                                    (40, 9, 6, True, 3), (25, 8, 7, True, 4)])
def test_quadratic_cone_newton_step_against_the_oracle(K, n, m, with_q, lmi_every):
    from test_gpu_parity import check_newton_step
    cons, cliques, num_vars, b = quad_tree_problem(K, n, m, 40 + K, with_q, lmi_every)
    o, k = build_both(cons, cliques, num_vars)
    rng = np.random.default_rng(K)
    for i, cn in enumerate(cons):
        if cn[0] == "lmi":
            R = rng.uniform(-.2, .2, (6, 6))
            W = np.eye(6) + R + R.T
        else:
            w1 = rng.uniform(-.3, .3, n)
            Q = cn[1] if cn[1] is not None else np.eye(n)
            W = np.r_[np.sqrt(w1 @ Q @ w1) + rng.uniform(.5, 1.5), w1]
        o.set_W(i, W)
        k.set_W(i, W)
    check_newton_step(o, k, b)
    check_newton_step(o, k, b, inv_sqrt_mu=0.9)       # a second iteration from the updated scaling points


def _maximize(L, p, b, cfg):
    y = np.zeros(len(b))
    bb = np.ascontiguousarray(b, dtype=np.float64)
    ok = L.CONEX_Maximize(p, ca.dp(bb), len(bb), C.byref(cfg), ca.dp(y), len(y))
    return ok, y


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_socp_property_through_the_c_abi(seed):
    """test_socp.cc on the device: SOC = LMI = quadratic cone (Q) = quadratic cone (square root)."""
    rng = np.random.default_rng(100 + seed)
    n = 3
    Wsqrt, As, Cs, Q, Aq, A_lmi = socp_formulations(rng, n)
    L = ca.api()
    cfg = ca.default_config()
    cfg.inv_sqrt_mu_max = 10000
    for i in range(-2, 2):
        b = np.full(n, float(i)) + rng.uniform(-1, 1, n) * .02
        ys = []
        for form in range(4):
            p = L.CONEX_CreateConeProgram()
            assert L.CONEX_SetNumberOfVariables(p, n) == 0
            if form == 0:      # the second-order cone as a quadratic cone with Q = I: covered below; here the LMI
                a = ca.colmajor(A_lmi)
                cm = ca.colmajor(np.eye(n + 1))
                assert L.CONEX_AddDenseLMIConstraint(p, ca.dp(a), n + 1, n + 1, n, ca.dp(cm), n + 1, n + 1) == 0
            elif form == 1:
                a, cc, q = ca.colmajor(Aq), np.ascontiguousarray(Cs), ca.colmajor(Q)
                assert L.CONEX_HIP_AddQuadraticConstraint(p, ca.dp(q), n, ca.dp(a), n + 1, n, ca.dp(cc), n + 1, None, 0) == 0
            elif form == 2:
                a, cc = ca.colmajor(As), np.ascontiguousarray(Cs)
                assert L.CONEX_HIP_AddQuadraticConstraint(p, None, n, ca.dp(a), n + 1, n, ca.dp(cc), n + 1, None, 0) == 0
            else:              # the oracle's second-order cone, as the reference's prog1
                L.CONEX_DeleteConeProgram(p)
                o = ol.Program(n)
                o.add_soc(As, Cs)
                o.initialize()
                ys.append(o.solve(b, oracle_cfg())[1])
                continue
            ok, y = _maximize(L, p, b, cfg)
            ys.append(y)
            L.CONEX_DeleteConeProgram(p)
        assert np.linalg.norm(ys[3] - ys[0]) <= 1e-4
        assert np.linalg.norm(ys[3] - ys[1]) <= 8e-6
        assert np.linalg.norm(ys[3] - ys[2]) <= 8e-6


@pytest.mark.gpu
def test_quadratic_cost_epigraph():
    """AddQuadraticCostEpigraph (quadratic_cone_constraint.h:88-117): t >= 1/2 z' Qi z.  Maximising
    -t + g'z leaves t = 1/2 z' Qi z at z = Qi^{-1} g."""
    rng = np.random.default_rng(8)
    nz = 4
    R = rng.uniform(-1, 1, (nz, nz))
    Qi = R @ R.T + nz * np.eye(nz)
    g = rng.uniform(-1, 1, nz)
    L = ca.api()
    p = L.CONEX_CreateConeProgram()
    assert L.CONEX_SetNumberOfVariables(p, nz + 1) == 0
    z = np.arange(nz, dtype=np.int64)
    q = ca.colmajor(Qi)
    assert L.CONEX_HIP_AddQuadraticCostEpigraph(p, ca.dp(q), nz, z.ctypes.data_as(C.POINTER(C.c_long)), nz) == 0
    cfg = ca.default_config()
    cfg.inv_sqrt_mu_max = 1e4
    cfg.final_centering_steps = 10
    cfg.max_iterations = 50
    ok, y = _maximize(L, p, np.r_[g, -1.0], cfg)
    L.CONEX_DeleteConeProgram(p)
    assert ok == 1
    zs = np.linalg.solve(Qi, g)
    assert np.linalg.norm(y[:nz] - zs) <= 1e-4 * (1 + np.linalg.norm(zs))
    assert abs(y[nz] - .5 * zs @ Qi @ zs) <= 1e-4
