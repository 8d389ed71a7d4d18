"""Seeded sweeps over constraint shapes: every LMI order from 2 to 40 (each lands on some Schur
kernel: the MFMA instances, zero-padded orders, the batched GEMM, the generic kernel) and Hermitian
cones over R / C / H of every order whose real representation has order <= 26, each with a random
number of variables, clique-tree shape and constraint count, against the oracle (one Newton step:
Schur blocks, slab, factor, direction, eigenvalue bounds, scaling-point update).  Reference:
dense_lmi_constraint.cc:72-103, hermitian_psd.cc:171-230, psd_constraint.cc:45-84."""
import numpy as np
import pytest

from conex_amd import KktContext
from conex_amd import synthetic as syn
from test_gpu_parity import check_newton_step, make_pair, rel

pytestmark = pytest.mark.gpu


def test_every_lmi_order_from_2_to_40():
    rng = np.random.default_rng(0)
    cases = 0
    for n in range(2, 41):
        for rep in range(2):
            mmax = min(31, n * (n + 1) // 2 - 1)
            if mmax < 3:
                continue
            m = int(rng.integers(3, mmax + 1))
            K = int(rng.integers(3, 40))
            ov = int(rng.integers(1, min(m - 2, 6) + 1))
            prob = syn.lmi_problem(K=K, n=n, m=m, branching=int(rng.integers(2, 9)), overlap=ov,
                                   seed=1000 + 7 * n + rep)
            W = syn.scaling_points(K, n, seed=2000 + n + rep)
            o, k = make_pair(prob, "lmi", W)
            check_newton_step(o, k, prob["b"])
            cases += 1
    assert cases >= 70


def test_every_hermitian_order_over_r_c_h():
    rng = np.random.default_rng(1)
    cases = 0
    for d in (1, 2, 4):
        for n in range(2, 24 // d + 3):
            dim = {1: n * (n + 1) // 2, 2: n * n, 4: n * (2 * n - 1)}[d]
            mmax = min(31, dim - 1)
            if mmax < 3:
                continue
            m = int(rng.integers(3, mmax + 1))
            K = int(rng.integers(2, 12))
            ov = int(rng.integers(1, min(m - 2, 5) + 1))
            prob = syn.hermitian_problem(K=K, n=n, d=d, m=m, branching=int(rng.integers(2, 5)), overlap=ov,
                                         seed=3000 + 11 * n + d)
            W = syn.hermitian_scaling_points(K, n, d, seed=4000 + n)
            o, k = make_pair(prob, "herm", W)
            check_newton_step(o, k, prob["b"])
            cases += 1
    assert cases >= 35


@pytest.mark.parametrize("d,n,m", [(4, 7, 9), (4, 10, 12), (2, 13, 10), (2, 20, 14), (4, 6, 40), (2, 8, 45)])
def test_hermitian_cones_on_the_gemm_assembly_fold_onto_their_top_rows(d, n, m, monkeypatch):
    """Complex / quaternion cones whose real representation (order d n >= 25, or more than 31 variables)
    assembles through the batched GEMM: only the top n rows of every product are formed and the
    contraction runs over n x d n entries with the block signs (+, -, .., -) -- 1 / d of the multiply-adds
    (kernels_lmi_large.hip.h, LmiLargeSchurFolded).  Against the oracle and against the unfolded form
    (CXK_NO_HERM_FOLD=1)."""
    prob = syn.hermitian_problem(K=3, n=n, d=d, m=m, branching=2, overlap=2, seed=9000 + 10 * n + d)
    W = syn.hermitian_scaling_points(3, n, d, seed=9100 + n)
    o, k = make_pair(prob, "herm", W)
    check_newton_step(o, k, prob["b"])
    monkeypatch.setenv("CXK_NO_HERM_FOLD", "1")
    k2 = syn.build(KktContext, prob, "herm", device=0)
    monkeypatch.delenv("CXK_NO_HERM_FOLD")
    k1 = syn.build(KktContext, prob, "herm", device=0)
    for kk in (k1, k2):
        for i in range(3):
            kk.set_W(i, W[i])
        kk.assemble()
    for i in range(3):
        G1, AW1, AQ1, sc1 = k1.constraint_schur(i)
        G2, AW2, AQ2, sc2 = k2.constraint_schur(i)
        assert rel(np.tril(G1), np.tril(G2)) <= 1e-13 and rel(AW1, AW2) <= 1e-13
        assert rel(AQ1, AQ2) <= 1e-13 and rel(sc1, sc2) <= 1e-13
