"""Hermitian matrices over the octonions (order <= 3): the reference's HermitianPsdConstraint<Octonions>.

CPU part -- pins the oracle's restatement (oracle/cxo_hermitian.c tables + the octonion branches of
cxo_program.c) with what the reference's own tests pin for T = Octonions (all properties: its data
come from libc rand()):
  jordan_matrix_algebra_test.cc:25-29    X e = X
  jordan_matrix_algebra_test.cc:30-51    Jordan identity  (A^2 o (A o B)) = A o (B o A^2), Hermitian results
  jordan_matrix_algebra_test.cc:53-71    1/2 ((B A) B + B (A B)) is NOT Q(B) A: the algebra is not associative
  jordan_matrix_algebra_test.cc:73-94    eigenvalues (roots of the minimal polynomial) of a square: positive,
                                         sum of squares = <Q, Q>, sum = <e, Q>
  exponential_map_test.cc:115-134        GeodesicUpdateScaled(e, s) has the eigenvalues of e exp(s) (to 1e-2)
plus the composition-algebra law |x y| = |x| |y| of the octonions themselves.

GPU part -- the HIP cone (kernels_oct.hip.h) against that oracle, stage by stage, alone and next to
matrix cones in a clique tree; the solves of hermitian_psd_test.cc:69-107 are in test_gpu_solver.py.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import synthetic as syn

D = 8


def planes(x):
    return ol.planes_colmajor(np.asarray(x, dtype=np.float64))


def unplanes(z, r, c):
    return np.transpose(np.asarray(z).reshape(D, c, r), (0, 2, 1))


def mul(X, Y):
    r, k, c = X.shape[1], X.shape[2], Y.shape[2]
    z = np.zeros(D * r * c)
    ol.lib().cxo_hc_multiply(D, r, k, c, ol.dp(planes(X)), ol.dp(planes(Y)), ol.dp(z))
    return unplanes(z, r, c)


def jordan(X, Y):
    return 0.5 * (mul(X, Y) + mul(Y, X))


def quadrep(X, Y):
    n = X.shape[1]
    z = np.zeros(D * n * n)
    ol.lib().cxo_hc_quadratic_representation(D, n, ol.dp(planes(X)), ol.dp(planes(Y)), ol.dp(z))
    return unplanes(z, n, n)


def ip(X, Y):
    return ol.lib().cxo_hc_trace_inner_product(D, X.shape[1], ol.dp(planes(X)), ol.dp(planes(Y)))


def conj_t(X):
    Z = np.transpose(X, (0, 2, 1)).copy()
    Z[1:] *= -1
    return Z


def identity(n):
    e = np.zeros((D, n, n))
    e[0] = np.eye(n)
    return e


def is_hermitian(X):
    return np.linalg.norm(X - conj_t(X)) <= 1e-12


def eigenvalues(X):
    """MatrixAlgebra::Eigenvalues (jordan_matrix_algebra.cc:33-56, 212-216): roots of the minimal
    polynomial found from the Jordan powers e, x, x o x, .."""
    n = X.shape[1]
    cols, p = [], identity(n)
    for _ in range(n):
        cols.append(p.ravel())
        p = jordan(p, X)
    coef = np.linalg.lstsq(np.array(cols).T, -p.ravel(), rcond=None)[0]
    return np.sort(np.roots(np.r_[1.0, coef[::-1]]).real)


def test_the_table_is_the_octonions():
    rng = np.random.default_rng(1)
    x, y, z = (rng.uniform(-1, 1, (D, 1, 1)) for _ in range(3))
    assert abs(np.linalg.norm(mul(x, y)) - np.linalg.norm(x) * np.linalg.norm(y)) < 1e-14   # composition algebra
    assert np.linalg.norm(mul(mul(x, y), z) - mul(x, mul(y, z))) > 1e-3                      # not associative
    assert np.allclose(mul(mul(x, x), y), mul(x, mul(x, y)), atol=1e-14)                     # but alternative
    X = rng.uniform(-1, 1, (D, 3, 3))
    assert np.allclose(mul(X, identity(3)), X, atol=1e-15)                                   # :25-29
    assert np.allclose(mul(X, identity(3)), syn.hc_multiply(X, identity(3)))


def test_jordan_identity_and_non_associativity():
    rng = np.random.default_rng(2)
    for _ in range(3):
        A, B = syn.random_hermitian(rng, D, 3), syn.random_hermitian(rng, D, 3)
        W = jordan(A, B)
        assert is_hermitian(A) and is_hermitian(B) and is_hermitian(W)
        A2 = jordan(A, A)
        assert np.allclose(jordan(A2, jordan(A, B)), jordan(A, jordan(B, A2)), atol=1e-10)  # :41-50
        Yref = quadrep(B, A)
        Y1 = 0.5 * mul(mul(B, A), B) + 0.5 * mul(B, mul(A, B))
        assert np.linalg.norm(Y1 - Yref) > 1e-8                                              # :66-67 EXPECT_FALSE
        assert np.allclose(quadrep(B, A), 2 * jordan(B, jordan(B, A)) - jordan(jordan(B, B), A), atol=1e-12)


def test_eigenvalues_of_a_square():
    rng = np.random.default_rng(3)
    for _ in range(3):
        Q = syn.random_hermitian(rng, D, 3)
        Q = jordan(Q, Q)
        lam = eigenvalues(Q)
        assert lam.min() > 1e-9                                                              # :88
        assert abs(np.sum(lam ** 2) - ip(Q, Q)) < 1e-7 * ip(Q, Q)                            # :89
        assert abs(np.sum(lam) - ip(identity(3), Q)) < 1e-8 * ip(identity(3), Q)             # :90


def test_geodesic_update_scaled_is_e_times_the_exponential():
    """exponential_map_test.cc:115-134 (GeodesicUpdateRescaling): w = e, s = -(e + 0.05 sym):
    eig(GeodesicUpdateScaled(w, s)) >= 0 and within 1e-2 of e * exp(eig(s))."""
    rng = np.random.default_rng(4)
    n = 3
    for _ in range(3):
        R = rng.uniform(-1, 1, (D, n, n))
        s = -(identity(n) + 0.05 * 0.5 * (R + conj_t(R)))
        out = np.zeros(D * n * n)
        ol.lib().cxo_hc_geodesic_update_scaled(D, n, ol.dp(planes(identity(n))), ol.dp(planes(s)), ol.dp(out))
        y = unplanes(out, n, n)
        assert is_hermitian(y)
        calc, ref = eigenvalues(y), np.sort(np.e * np.exp(eigenvalues(s)))
        assert calc.min() >= 0 and np.allclose(calc, ref, atol=1e-2)


def test_octonion_cone_of_order_above_three_is_refused_by_the_oracle():
    o = ol.Program(2)
    rng = np.random.default_rng(5)
    A = np.array([syn.random_hermitian(rng, D, 4) for _ in range(2)])
    assert o.add_hermitian(A, identity(4)) == -1


# ------------------------------------------------------------------------------------------ GPU
def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(b))


def octonion_scaling_point(rng, n, scale=0.2):
    H = syn.random_hermitian(rng, D, n) * (scale / 2) + identity(n)
    return jordan(H, H)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(3, 5), (2, 3), (1, 1), (3, 8)])
def test_octonion_newton_step_against_the_oracle(n, m):
    from conex_amd import KktContext
    rng = np.random.default_rng(10 * n + m)
    A = np.array([syn.random_hermitian(rng, D, n) for _ in range(m)])
    Cm = identity(n)
    b = 0.5 * np.trace(A[:, 0], axis1=1, axis2=2)
    o, k = ol.Program(m), KktContext(m, device=0)
    for p in (o, k):
        assert p.add_hermitian(A, Cm) == 0
        p.initialize()
    W = octonion_scaling_point(rng, n)
    o.set_W(0, W)
    k.set_W(0, W)
    o.assemble()
    k.assemble()
    Go, AWo, AQo, sco = o.constraint_schur(0)
    Gk, AWk, AQk, sck = k.constraint_schur(0)
    assert rel(np.tril(Gk), np.tril(Go)) <= 1e-13 and rel(AWk, AWo) <= 1e-13 and rel(AQk, AQo) <= 1e-13
    assert rel(sck, sco) <= 1e-13
    oko, yo = o.kkt_solve(b, 0.7, 0.9, 0.8)
    okk, yk = k.kkt_solve(b, 0.7, 0.9, 0.8)
    assert oko == okk == 1 and rel(yk, yo) <= 1e-10
    # the reference's heuristic step quantities, then the geodesic update, three times over
    for it in range(3):
        eo = o.weighted_slack_eigenvalues(yo, 0.8)
        ek = k.weighted_slack_eigenvalues(yo, 0.8)
        assert rel(ek, eo) <= 1e-11
        io = o.prepare_step(yo, 0.8, 0.3)
        ik = k.prepare_step(yo, 0.8, 0.3)
        assert rel(ik, io) <= 1e-11
        o.take_step(0.4 + 0.2 * it, 0.3)
        k.take_step(0.4 + 0.2 * it, 0.3)
        assert rel(k.get_W(0), o.get_W(0)) <= 1e-11
    # the affine flag changes nothing for this cone (hermitian_psd.cc:129-145 never reads it)
    Wb = k.get_W(0).copy()
    o.prepare_step(yo, 0.0, 0.0, affine=1)
    k.prepare_step(yo, 0.0, 0.0, affine=1)
    assert np.array_equal(k.get_W(0), Wb) and rel(k.get_W(0), o.get_W(0)) <= 1e-11


@pytest.mark.gpu
def test_octonion_cones_next_to_matrix_cones_in_a_clique_tree():
    from conex_amd import KktContext
    rng = np.random.default_rng(77)
    prob = syn.lmi_problem(K=7, n=6, m=6, branching=2, overlap=2, seed=7)
    nv = prob["num_vars"]
    octs = [(np.array([syn.random_hermitian(rng, D, 3) for _ in range(4)]), prob["cliques"][c][:4]) for c in (0, 3, 6)]
    b = prob["b"].copy()
    for A, cl in octs:
        b[cl] += 0.5 * np.trace(A[:, 0], axis1=1, axis2=2)
    Wl = syn.scaling_points(7, 6, seed=8)

    def build(cls, **kw):
        p = cls(nv, **kw)
        for ci, cl in enumerate(prob["cliques"]):
            p.add_lmi(prob["A"][ci], prob["C"][ci], cl)
        for A, cl in octs:
            assert p.add_hermitian(A, identity(3), cl) >= 0
        p.initialize()
        for i in range(7):
            p.set_W(i, Wl[i])
        for q in range(3):
            p.set_W(7 + q, octonion_scaling_point(np.random.default_rng(90 + q), 3))
        return p

    o, k = build(ol.Program), build(KktContext, device=0)
    oko, yo = o.kkt_solve(b, 0.7, 0.9, 0.8)
    okk, yk = k.kkt_solve(b, 0.7, 0.9, 0.8)
    assert oko == okk == 1 and rel(yk, yo) <= 1e-10
    io, ik = o.prepare_step(yo, 0.8, 0.3), k.prepare_step(yo, 0.8, 0.3)
    assert rel(ik, io) <= 1e-9
    o.take_step(0.7, 0.3)
    k.take_step(0.7, 0.3)
    for i in range(10):
        assert rel(k.get_W(i), o.get_W(i)) <= 1e-10
