"""Pins the oracle's restatement of the Hermitian PSD cone over R / C / H (oracle/cxo_hermitian.c)
with what the reference's own tests pin (all properties -- its data come from libc rand()):

  jordan_matrix_algebra_test.cc  complex / quaternion product is associative, Q(x) y identity
  exponential_map_test.cc:31-49  DoExponentialMap vs the true exponential on a small argument
  hermitian_psd_test.cc:25-66    Real Hermitian == DenseLMI solution (y and X to 1e-11 there)
  hermitian_psd_test.cc:69-116   random instances over R, C, H solve
"""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg

import oracle_lib as ol
from conex_amd import synthetic as syn


def hc_mul(d, X, Y):
    L = ol.lib()
    r, k, c = X.shape[1], X.shape[2], Y.shape[2]
    x, y = ol.planes_colmajor(X), ol.planes_colmajor(Y)
    z = np.zeros(d * r * c)
    L.cxo_hc_multiply(d, r, k, c, ol.dp(x), ol.dp(y), ol.dp(z))
    return np.transpose(z.reshape(d, c, r), (0, 2, 1))


def test_complex_product_is_numpy_complex_product():
    rng = np.random.default_rng(1)
    X, Y = rng.uniform(-1, 1, (2, 5, 4)), rng.uniform(-1, 1, (2, 4, 3))
    Z = hc_mul(2, X, Y)
    Zc = (X[0] + 1j * X[1]) @ (Y[0] + 1j * Y[1])
    assert np.allclose(Z[0], Zc.real, atol=1e-14) and np.allclose(Z[1], Zc.imag, atol=1e-14)
    assert np.allclose(Z, syn.hc_multiply(X, Y), atol=1e-14)


@pytest.mark.parametrize("d", [1, 2, 4])
def test_product_is_associative_and_norm_multiplicative(d):
    rng = np.random.default_rng(2 + d)
    X, Y, Zm = (rng.uniform(-1, 1, (d, 4, 4)) for _ in range(3))
    lhs = hc_mul(d, hc_mul(d, X, Y), Zm)
    rhs = hc_mul(d, X, hc_mul(d, Y, Zm))
    assert np.allclose(lhs, rhs, atol=1e-13)
    # 1 x 1 "matrices" are the scalars of the algebra: |xy| = |x||y|
    x, y = rng.uniform(-1, 1, (d, 1, 1)), rng.uniform(-1, 1, (d, 1, 1))
    assert abs(np.linalg.norm(hc_mul(d, x, y)) - np.linalg.norm(x) * np.linalg.norm(y)) < 1e-14


@pytest.mark.parametrize("d", [1, 2])
def test_exponential_map_matches_expm_on_small_argument(d):
    rng = np.random.default_rng(7)
    n = 6
    H = syn.random_hermitian(rng, d, n) * 1e-3
    y = np.zeros(d * n * n)
    ol.lib().cxo_hc_exponential_map(d, n, ol.dp(ol.planes_colmajor(H)), ol.dp(y))
    Y = np.transpose(y.reshape(d, n, n), (0, 2, 1))
    Hc = H[0] + (1j * H[1] if d == 2 else 0)
    E = scipy.linalg.expm(Hc)
    got = Y[0] + (1j * Y[1] if d == 2 else 0)
    assert np.max(np.abs(got - E)) < 1e-8          # reference test: 1e-8 on A * 1e-3


@pytest.mark.parametrize("d", [1, 2, 4])
def test_ritz_values_lie_in_the_spectrum(d):
    rng = np.random.default_rng(11 + d)
    n = 8
    W = syn.hermitian_scaling_points(1, n, d, seed=5)[0]
    S = syn.random_hermitian(rng, d, n)
    WS = syn.hc_multiply(W, S)
    r = rng.uniform(-1, 1, (d, n, 1))
    eigs = np.zeros(n + 2)
    ne = ol.lib().cxo_hc_approximate_eigenvalues(
        d, n, ol.dp(ol.planes_colmajor(WS)), ol.dp(ol.planes_colmajor(W)),
        ol.dp(ol.planes_colmajor(r)), n // 2 + 1, ol.dp(eigs))
    assert 1 <= ne <= n // 2 + 1
    # spectrum of W S through the real representation (symmetric-definite pencil)
    def embed(X):
        dd = X.shape[0]
        return np.block([[syn.HC_SIGN[k ^ j, j] * X[k ^ j] for j in range(dd)] for k in range(dd)])
    lam = scipy.linalg.eigh(embed(S), np.linalg.inv(embed(W)), eigvals_only=True)
    assert eigs[:ne].min() >= lam.min() - 1e-9 and eigs[:ne].max() <= lam.max() + 1e-9


def test_random_generator_is_uniform_and_stateless():
    L = ol.lib()
    v = np.array([L.cxo_hc_random(3, 7, i) for i in range(4000)])
    assert v.min() >= -1 and v.max() < 1 and abs(v.mean()) < 0.05 and abs(v.std() - 3 ** -0.5) < 0.02
    assert L.cxo_hc_random(3, 7, 5) == v[5] and L.cxo_hc_random(3, 8, 5) != v[5]


def solve(kind, prob, cfg):
    p = syn.build(ol.Program, prob, kind)
    ok, y = p.solve(prob["b"], cfg)
    return p, ok, y


def test_real_hermitian_agrees_with_dense_lmi():
    """hermitian_psd_test.cc:25-66 (rank 8, 4 variables, 4 instances)."""
    for inst in range(4):
        hp = syn.hermitian_problem(K=1, n=8, d=1, m=4, seed=40 + inst)
        lp = dict(hp, A=hp["A"][:, :, 0], C=hp["C"][:, 0])
        cfg = ol.default_config()
        cfg.inv_sqrt_mu_max = np.sqrt(1.0 / 1e-4)
        cfg.final_centering_tolerance = 1e-8
        cfg.prepare_dual_variables = 1
        ph, ok1, y1 = solve("herm", hp, cfg)
        pl, ok2, y2 = solve("lmi", lp, cfg)
        assert ok1 == 1 and ok2 == 1
        assert np.linalg.norm(y1 - y2) < 1e-9
        X1 = ph.dual_variable(0)[:64]
        X2 = pl.dual_variable(0)
        assert np.linalg.norm(X1 - X2) < 1e-8


@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("rank,m", [(3, 2), (13, 5)])
def test_random_instances_solve(d, rank, m):
    """hermitian_psd_test.cc:69-116 (sizes reduced to keep the CPU suite short)."""
    prob = syn.hermitian_problem(K=1, n=rank, d=d, m=m, seed=70 + rank + d)
    cfg = ol.default_config()
    cfg.inv_sqrt_mu_max = 1000
    cfg.final_centering_steps = 4
    cfg.max_iterations = 100
    p, ok, y = solve("herm", prob, cfg)
    assert ok == 1
    # dual feasibility of the returned point: C - sum y_i A_i is positive semidefinite
    S = prob["C"][0] - np.einsum("i,idab->dab", y, prob["A"][0])
    dd = d
    E = np.block([[syn.HC_SIGN[k ^ j, j] * S[k ^ j] for j in range(dd)] for k in range(dd)])
    assert np.linalg.eigvalsh(E).min() > -1e-7
