"""Parity of the HIP Newton-step path (through the cxk_* C-ABI) with the CPU oracle.

Bars: integer/index structures bit-exact (tests/test_symbolic_parity.py); floating point --
per-constraint Schur blocks and the assembled slab <= 1e-13 relative, the Newton direction
<= 1e-10 relative norm (BASELINE.json north_star), scaling-point updates <= 1e-11.
All tests need a real MI355X.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import KktContext
from conex_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL_SCHUR = 1e-13
TOL_DIRECTION = 1e-10
TOL_UPDATE = 1e-11


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def blocks(p, slab):
    """Meaningful slab entries: lower triangle of each diagonal block + the off-diagonal block
    (the reference never reads the strict upper triangle: LLT and ToDense use Lower only)."""
    sizes = p.supernode_sizes()
    dof, oof = p.block_offsets()
    out = []
    for e in range(p.K):
        ns = int(sizes[e])
        nsep = len(p.get_list(0, e)) - ns
        D = slab[dof[e]:dof[e] + ns * ns].reshape(ns, ns).T
        out.append(np.tril(D).ravel())
        out.append(slab[oof[e]:oof[e] + ns * nsep])
    return np.concatenate(out) if out else np.zeros(0)


def make_pair(prob, kind, W=None):
    o = syn.build(ol.Program, prob, kind)
    k = syn.build(KktContext, prob, kind, device=0)
    if W is not None:
        for i in range(o.K):
            o.set_W(i, W[i])
            k.set_W(i, W[i])
    return o, k


def check_newton_step(o, k, b, inv_sqrt_mu=0.7, bs=0.9, cs=0.8, check_update=True, lanczos_tol=None):
    """assemble -> factor -> solve -> prepare -> take step, compared stage by stage."""
    o.assemble()
    k.assemble()
    for i in range(0, o.K, max(1, o.K // 7)):
        Go, AWo, AQo, sco = o.constraint_schur(i)
        Gk, AWk, AQk, sck = k.constraint_schur(i)
        assert rel(np.tril(Gk), np.tril(Go)) <= TOL_SCHUR
        assert rel(AWk, AWo) <= TOL_SCHUR and rel(AQk, AQo) <= TOL_SCHUR
        assert rel(sck, sco) <= TOL_SCHUR
    # (a chain-shaped tree factored in its segment-parallel order keeps the factor in a layout of its
    # own: the stage-by-stage slab comparison needs CXK_CHAIN_SEGMENTS=0, the reference's order)
    segmented = k.chain_segments() != 0
    if not segmented:
        assert rel(blocks(k, k.slab()), blocks(o, o.slab())) <= TOL_SCHUR
    AWo, AQo, sco = o.residuals()
    AWk, AQk, sck = k.residuals()
    assert rel(AWk, AWo) <= TOL_SCHUR and rel(AQk, AQo) <= TOL_SCHUR and rel(sck, sco) <= 1e-12

    assert o.factor() == 1
    assert k.factor() == 1
    if not segmented:
        assert rel(blocks(k, k.slab()), blocks(o, o.slab())) <= 1e-11

    N = o.N
    bb = np.zeros(N)
    bb[:len(b)] = b
    rhs = inv_sqrt_mu * (bb * bs + AQo * cs) - 2 * AWo
    yo = o.solve_inplace(rhs)
    k.set_cost(b)
    k.newton_direction(inv_sqrt_mu, bs, cs)
    yk = k.get_y()
    assert rel(yk, yo) <= TOL_DIRECTION
    # solve_inplace on an arbitrary host vector
    assert rel(k.solve_inplace(rhs), yo) <= TOL_DIRECTION

    if not check_update:
        return yo
    c_weight = inv_sqrt_mu * cs
    eo = o.weighted_slack_eigenvalues(yo, c_weight)
    ek = k.weighted_slack_eigenvalues(yo, c_weight)
    io = o.prepare_step(yo, c_weight, 1.0)
    ik = k.prepare_step(yo, c_weight, 1.0)
    if lanczos_tol is None:
        assert rel(ek, eo) <= 1e-9
        assert rel(ik, io) <= 1e-9
    else:
        # Dozens of unreorthogonalised Lanczos steps amplify summation-order differences in the
        # extreme Ritz values (they are estimates in the reference too): compare them relative to
        # the spectral width; traces / norms stay tight.
        width = abs(eo[1] - eo[0])
        assert np.max(np.abs(ek[:2] - eo[:2])) <= lanczos_tol * width
        assert rel(ek[2:], eo[2:]) <= 1e-11
        assert abs(ik[0] - io[0]) <= 1e-11 * abs(io[0])
        assert abs(ik[1] - io[1]) <= lanczos_tol * max(width, abs(io[1]))
    step = min(1.0, 2.0 / (io[1] * io[1]))
    o.take_step(step)
    k.take_step(step)
    for i in range(0, o.K, max(1, o.K // 11)):
        assert rel(k.get_W(i), o.get_W(i)) <= TOL_UPDATE
    return yo


# --------------------------------------------------------------------- LMI
@pytest.mark.parametrize("K,n,m,b_,ov", [(1, 4, 3, 8, 1), (9, 6, 6, 8, 2), (30, 20, 20, 8, 5),
                                          (40, 7, 9, 3, 4), (12, 33, 5, 2, 2), (11, 24, 24, 3, 6),
                                          # shapes on every Schur kernel: MFMA producer/consumer (12, 9), literal LDS
                                          # (16 with 31 matrices; 28), batched GEMM pipeline (32)
                                          (20, 12, 9, 3, 3), (14, 16, 30, 3, 8), (9, 28, 28, 2, 6), (8, 32, 20, 2, 5)])
def test_lmi_newton_step(K, n, m, b_, ov):
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=b_, overlap=ov, seed=100 + K)
    W = syn.scaling_points(K, n, seed=7 + K)
    o, k = make_pair(prob, "lmi", W)
    check_newton_step(o, k, prob["b"])


@pytest.mark.parametrize("K,n,m,b_,ov", [(1, 20, 1, 2, 1), (7, 20, 9, 3, 2), (300, 20, 15, 4, 3), (5, 20, 16, 2, 4),
                                          (70, 20, 20, 8, 5), (3, 24, 10, 2, 2), (9, 24, 14, 2, 5),
                                          # orders up to 16: part of one 16-column stage-1 tile, no 4x4x4 blocks there
                                          (40, 8, 5, 3, 2), (300, 8, 23, 4, 3), (9, 12, 12, 2, 4), (60, 12, 20, 4, 5),
                                          (1, 16, 1, 2, 1), (33, 16, 16, 3, 6), (270, 16, 21, 8, 5),
                                          # 25 .. 32 matrices: three contraction tiles
                                          (14, 16, 30, 3, 8), (300, 12, 31, 4, 6), (20, 8, 24, 3, 5), (5, 16, 27, 2, 9),
                                          # orders without an instance run zero-padded on the next one up
                                          (40, 7, 9, 3, 4), (300, 22, 14, 8, 5), (33, 10, 18, 3, 6), (9, 13, 28, 2, 7),
                                          (270, 17, 6, 4, 3), (12, 23, 14, 2, 5), (25, 5, 4, 3, 2), (7, 3, 2, 2, 1),
                                          (300, 19, 20, 4, 6), (16, 9, 31, 3, 8),
                                          # more matrices than fit LDS twice: ONE P image (stages 1 and 2 take turns)
                                          (300, 20, 23, 4, 6), (270, 24, 20, 8, 5), (40, 22, 17, 3, 4), (9, 18, 22, 2, 7),
                                          (600, 24, 15, 4, 3)])
def test_lmi_mfma_kernel_shapes(K, n, m, b_, ov):
    """The persistent MFMA Schur kernel (lmi_fused_mfma.hip) takes the number of variables at run
    time: one 16 x 16 contraction tile up to 16 matrices (m + 1), a tile plus two corner triangles
    from 17 to 24, three tiles from 25 to 32 (orders whose P images fit LDS twice);
    K = 300 gives the 256 workgroups two constraints each (both P images in use), K < 256 one."""
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=b_, overlap=ov, seed=700 + K + m)
    W = syn.scaling_points(K, n, seed=17 + K)
    o, k = make_pair(prob, "lmi", W)
    assert k.count_lmi_kernel(2) == K
    check_newton_step(o, k, prob["b"])
    # bit-reproducible: all sums run in a fixed order
    k.assemble()
    G1 = [k.constraint_schur(i)[0].copy() for i in range(0, K, max(1, K // 5))]
    k.assemble()
    G2 = [k.constraint_schur(i)[0] for i in range(0, K, max(1, K // 5))]
    assert all(np.array_equal(np.tril(a), np.tril(b)) for a, b in zip(G1, G2))


def test_lmi_mfma_kernel_many_constraints_per_workgroup():
    """More constraints than 4 x CUs: every workgroup walks a long list (P images alternate)."""
    K = 1500
    prob = syn.lmi_problem(K=K, n=20, m=4, branching=8, overlap=1, seed=3)
    W = syn.scaling_points(K, 20, seed=4)
    o, k = make_pair(prob, "lmi", W)
    assert k.count_lmi_kernel(2) == K
    o.assemble()
    k.assemble()
    for i in list(range(0, K, 97)) + [K - 1, K - 2, 255, 256, 257, 511, 512, 1023, 1024, 1279, 1280]:
        Go, AWo, AQo, sco = o.constraint_schur(i)
        Gk, AWk, AQk, sck = k.constraint_schur(i)
        assert rel(np.tril(Gk), np.tril(Go)) <= TOL_SCHUR
        assert rel(AWk, AWo) <= TOL_SCHUR and rel(AQk, AQo) <= TOL_SCHUR and rel(sck, sco) <= TOL_SCHUR
    assert rel(blocks(k, k.slab()), blocks(o, o.slab())) <= TOL_SCHUR


@pytest.mark.parametrize("n,m", [(20, 20), (7, 5), (24, 24), (40, 6)])
def test_lmi_non_symmetric_data_matches_the_reference_formula(n, m):
    """The reference accepts non-symmetric A_i / C and evaluates G(i,j) = <W A_i W, A_j>,
    AW(i) = tr(A_i W), AQc(i) = <C, W A_i W> as written (dense_lmi_constraint.cc:72-88; the oracle
    restates exactly that).  The fast kernels use tr(W A_i W A_j) = tr(P_i P_j), P = A W, which
    holds for symmetric data only, so such a constraint is routed to the literal kernels."""
    rng = np.random.default_rng(n + m)
    prob = syn.lmi_problem(K=6, n=n, m=m, branching=2, overlap=min(2, m - 1), seed=55)
    prob["A"] = rng.uniform(-1, 1, prob["A"].shape)               # no symmetrisation
    prob["A"][0] = 0.5 * (prob["A"][0] + np.transpose(prob["A"][0], (0, 2, 1)))  # constraint 0 stays symmetric
    W = syn.scaling_points(6, n, seed=8)
    o, k = make_pair(prob, "lmi", W)
    assert k.count_lmi_kernel(0) >= 5
    o.assemble()
    k.assemble()
    for i in range(6):
        Go, AWo, AQo, sco = o.constraint_schur(i)
        Gk, AWk, AQk, sck = k.constraint_schur(i)
        assert rel(np.tril(Gk), np.tril(Go)) <= TOL_SCHUR
        assert rel(AWk, AWo) <= TOL_SCHUR and rel(AQk, AQo) <= TOL_SCHUR and rel(sck, sco) <= TOL_SCHUR
    assert rel(blocks(k, k.slab()), blocks(o, o.slab())) <= TOL_SCHUR


def test_lmi_non_symmetric_data_beyond_lds_orders_is_refused():
    prob = syn.lmi_problem(K=1, n=90, m=3, branching=2, overlap=1, seed=5)
    prob["A"] = np.random.default_rng(1).uniform(-1, 1, prob["A"].shape)
    k = KktContext(prob["num_vars"], device=0)
    k.add_lmi(prob["A"][0], prob["C"][0], prob["cliques"][0])
    with pytest.raises(RuntimeError, match="non-symmetric"):
        k.initialize()


@pytest.mark.parametrize("K,n,m,b_,ov", [(1, 70, 6, 2, 1), (3, 96, 9, 2, 3), (1, 200, 50, 2, 1)])
def test_lmi_large_order_newton_step(K, n, m, b_, ov):
    """Orders beyond the LDS-resident kernels: HBM-resident matrices, every product on the fp64
    MFMA GEMM (kernels_lmi_large.hip.h).  (1, 200, 50) is BASELINE config 2."""
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=b_, overlap=ov, seed=300 + n)
    W = syn.scaling_points(K, n, seed=11 + n)
    o, k = make_pair(prob, "lmi", W)
    # 100 unreorthogonalised Lanczos steps at n = 200: the small-end Ritz value is converged to
    # ~1e-3 of the spectral width only (in the reference as well); the large end agrees to 1e-15
    check_newton_step(o, k, prob["b"], lanczos_tol=1e-5 if n < 200 else 2e-3)


def test_lmi_large_order_affine_update():
    prob = syn.lmi_problem(K=2, n=80, m=4, branching=2, overlap=2, seed=19)
    W = syn.scaling_points(2, 80, seed=4)
    o, k = make_pair(prob, "lmi", W)
    y = np.random.default_rng(0).uniform(-0.05, 0.05, o.N)
    o.prepare_step(y, 0.0, 0.0, affine=1)
    k.prepare_step(y, 0.0, 0.0, affine=1)
    for i in range(o.K):
        assert rel(k.get_W(i), o.get_W(i)) <= 1e-13


@pytest.mark.parametrize("K,n,m,b_,ov", [(1, 52, 40, 2, 1), (5, 10, 60, 2, 20), (3, 45, 100, 2, 30)])
def test_mid_size_supernodes_and_lds_resident_orders(K, n, m, b_, ov):
    """Supernodes beyond the wave-per-supernode kernels (ns > 32 or separators > 16) take the
    workgroup-per-supernode kernel; n = 45..52 exercises the LDS-resident LMI kernels above 64 KB
    of dynamic LDS."""
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=b_, overlap=ov, seed=400 + m)
    W = syn.scaling_points(K, n, seed=13 + n)
    o, k = make_pair(prob, "lmi", W)
    check_newton_step(o, k, prob["b"], lanczos_tol=1e-7)


def test_huge_supernodes_take_the_blocked_hbm_path():
    """Panels beyond LDS (> ~140 columns): blocked right-looking Cholesky in HBM, 32-column panels,
    SYRK / GEMM / TRSM updates on the fp64 MFMA GEMM (kernels_kkt_big.hip.h)."""
    # (a) dense LP with 300 variables: one 300 x 300 supernode
    prob = syn.lp_problem(rows=400, num_vars=300, seed=8)
    o, k = make_pair(prob, "lp")
    check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.4)
    # (b) LMI tree with 200-variable cliques: leaves 140 + 60 separators, root 200
    prob = syn.lmi_problem(K=3, n=24, m=200, branching=2, overlap=60, seed=9)   # 24*25/2 >= 200
    W = syn.scaling_points(3, 24, seed=5)
    o, k = make_pair(prob, "lmi", W)
    sizes = o.supernode_sizes()
    assert max(sizes) == 200 and min(sizes) == 140
    check_newton_step(o, k, prob["b"], check_update=False)


@pytest.mark.parametrize("num_vars,rows", [(141, 200), (193, 260), (257, 300), (321, 400)])
def test_big_supernode_odd_sizes(num_vars, rows):
    """One dense supernode whose size is not a multiple of the 32-column panel: the last panel is
    ragged, the work items of big_panel do not fill their wavefronts, the streamed solves end on a
    short block."""
    prob = syn.lp_problem(rows=rows, num_vars=num_vars, seed=num_vars)
    o, k = make_pair(prob, "lp")
    check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.4)


@pytest.mark.parametrize("num_vars,rows", [(500, 640), (896, 1000), (926, 1030), (960, 1100)])
def test_big_supernode_one_launch_and_host_driven_loop_agree(num_vars, rows, monkeypatch):
    """big_chol_dataflow (one launch: a workgroup per 32-column block column, block columns handed
    on through flags) against the host-driven panel loop it replaces up to 927 rows
    (CXK_NO_BIG_DATAFLOW=1; 926 variables + the right-hand side are the last size the one launch takes,
    960 variables take the host-driven loop either way) and the oracle."""
    prob = syn.lp_problem(rows=rows, num_vars=num_vars, seed=num_vars)
    o, k = make_pair(prob, "lp")
    check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.4)
    k1 = syn.build(KktContext, prob, "lp", device=0)
    monkeypatch.setenv("CXK_NO_BIG_DATAFLOW", "1")
    k2 = syn.build(KktContext, prob, "lp", device=0)
    monkeypatch.delenv("CXK_NO_BIG_DATAFLOW")
    ys = []
    for kk in (k1, k2):
        ok, y = kk.kkt_solve(prob["b"], 0.4, 0.9, 0.8)
        assert ok == 1
        ys.append(y)
        ok, y = kk.kkt_solve(prob["b"], 0.4, 0.9, 0.8)     # a second launch on the same flag words
        assert ok == 1 and np.array_equal(y, ys[-1])
    assert rel(ys[0], ys[1]) <= 1e-11
    # a pivot that is not positive is reported by either path
    for kk in (k1, k2):
        kk.set_W(0, np.zeros(rows))                        # G = 0
        ok, _ = kk.kkt_solve(prob["b"], 0.4, 0.9, 0.8)
        assert ok == 0


@pytest.mark.parametrize("m,overlap", [(170, 21), (230, 45)])
def test_big_supernodes_with_separators(m, overlap):
    """Big leaves (m - overlap columns, `overlap` separator columns: the off block rides through
    big_panel, U = off^T off goes to the parent) under a big root."""
    n = 24 if m <= 200 else 28
    prob = syn.lmi_problem(K=3, n=n, m=m, branching=2, overlap=overlap, seed=m)
    W = syn.scaling_points(3, n, seed=5)
    o, k = make_pair(prob, "lmi", W)
    sizes = o.supernode_sizes()
    assert max(sizes) == m and min(sizes) == m - overlap
    check_newton_step(o, k, prob["b"], check_update=False)


def test_lmi_identity_start_and_iterations():
    """Three IPM iterations from W = I through both paths stay in lock-step."""
    prob = syn.lmi_problem(K=20, n=8, m=8, branching=3, overlap=3, seed=5)
    o, k = make_pair(prob, "lmi")
    for it in range(3):
        check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.5 + 0.2 * it)


def test_lmi_cholesky_failure_is_reported():
    prob = syn.lmi_problem(K=5, n=4, m=4, branching=2, overlap=2, seed=3)
    k = syn.build(KktContext, prob, "lmi", device=0)
    W = np.zeros((4, 4))  # singular scaling point => zero Schur complement
    for i in range(k.K):
        k.set_W(i, W)
    k.assemble()
    assert k.factor() == 0


def test_lmi_affine_update():
    prob = syn.lmi_problem(K=6, n=5, m=5, branching=2, overlap=2, seed=9)
    W = syn.scaling_points(6, 5, seed=2)
    o, k = make_pair(prob, "lmi", W)
    y = np.random.default_rng(0).uniform(-0.1, 0.1, o.N)
    o.prepare_step(y, 0.0, 0.0, affine=1)
    k.prepare_step(y, 0.0, 0.0, affine=1)
    for i in range(o.K):
        assert rel(k.get_W(i), o.get_W(i)) <= 1e-13


# --------------------------------------------------------------------- Hermitian PSD over R/C/H
@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("K,n,m,b_,ov", [(1, 3, 2, 2, 1), (7, 6, 5, 2, 2), (3, 12, 8, 2, 3), (5, 12, 24, 2, 4)])
def test_hermitian_newton_step(d, K, n, m, b_, ov):
    """B2 / C7: HermitianPsdConstraint<Real|Complex|Quaternions> (hermitian_psd.cc) against the
    plane-by-plane oracle.  The device runs the real representation of order d n on the LMI
    kernels with the Hermitian step rules; W comes back as d planes."""
    prob = syn.hermitian_problem(K=K, n=n, d=d, m=m, branching=b_, overlap=ov, seed=500 + 7 * n + d)
    W = syn.hermitian_scaling_points(K, n, d, seed=31 + n)
    o, k = make_pair(prob, "herm", W)
    check_newton_step(o, k, prob["b"])


def test_hermitian_large_order_and_iterations():
    """Quaternion order 20 -> real representation of order 80 (HBM-resident GEMM path), three
    Newton iterations from W = I in lock-step with the oracle."""
    prob = syn.hermitian_problem(K=2, n=20, d=4, m=6, branching=2, overlap=2, seed=77)
    o, k = make_pair(prob, "herm")
    for it in range(3):
        check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.4 + 0.2 * it, lanczos_tol=1e-6)


def test_hermitian_affine_update_and_identity():
    prob = syn.hermitian_problem(K=3, n=5, d=2, m=4, branching=2, overlap=2, seed=12)
    o, k = make_pair(prob, "herm")
    W0 = k.get_W(0)
    assert W0.shape == (2, 5, 5) and np.array_equal(W0[0], np.eye(5)) and not W0[1].any()
    W = syn.hermitian_scaling_points(3, 5, 2, seed=9)
    for i in range(3):
        o.set_W(i, W[i])
        k.set_W(i, W[i])
        assert np.array_equal(k.get_W(i), W[i])
    y = np.random.default_rng(0).uniform(-0.1, 0.1, o.N)
    o.prepare_step(y, 0.0, 0.3, affine=1)
    k.prepare_step(y, 0.0, 0.3, affine=1)
    for i in range(o.K):
        assert rel(k.get_W(i), o.get_W(i)) <= 1e-13


# --------------------------------------------------------------------- equalities / LDLT (B8, B10)
def eq_pair(seed, kind):
    """LP or chordal-LMI program plus equality blocks, built identically on both sides."""
    rng = np.random.default_rng(seed)
    if kind == "lp":
        nv = 8
        A = rng.uniform(-1, 1, (14, nv))
        c = np.abs(rng.uniform(0.5, 1.5, 14))
        eqs = [(rng.uniform(-1, 1, (3, nv)), rng.uniform(-1, 1, 3), None)]
        b = rng.uniform(-1, 1, nv)

        def build(cls, **kw):
            p = cls(nv, **kw)
            p.add_linear(A, c)
            for Ae, be, v in eqs:
                p.add_equality(Ae, be, v)
            p.initialize()
            return p
        W = None
    else:
        prob = syn.lmi_problem(K=12, n=6, m=6, branching=3, overlap=2, seed=seed)
        nv = prob["num_vars"]
        b = prob["b"]
        # one coupling equality on the root clique and two on leaf cliques
        eqs = [(rng.uniform(-1, 1, (2, 6)), rng.uniform(-0.1, 0.1, 2), prob["cliques"][0]),
               (rng.uniform(-1, 1, (1, 6)), rng.uniform(-0.1, 0.1, 1), prob["cliques"][7]),
               (rng.uniform(-1, 1, (1, 3)), rng.uniform(-0.1, 0.1, 1), prob["cliques"][11][:3])]

        def build(cls, **kw):
            p = cls(nv, **kw)
            for ci, cl in enumerate(prob["cliques"]):
                p.add_lmi(prob["A"][ci], prob["C"][ci], cl)
            for Ae, be, v in eqs:
                p.add_equality(Ae, be, v)
            p.initialize()
            return p
        W = syn.scaling_points(12, 6, seed=seed + 1)
    o, k = build(ol.Program), build(KktContext, device=0)
    if W is not None:
        for i in range(len(W)):
            o.set_W(i, W[i])
            k.set_W(i, W[i])
    return o, k, b


@pytest.mark.parametrize("kind,seed", [("lp", 1), ("lp", 2), ("lmi", 3), ("lmi", 4)])
def test_equality_constraints_ldlt_newton_step(kind, seed):
    """Multipliers make the KKT matrix indefinite: block LDLT with diagonal pivoting
    (BlockLDLTInPlace over Eigen::RLDLT) and its solves, against the oracle."""
    o, k, b = eq_pair(seed, kind)
    assert k.N == o.N > len(b)
    check_newton_step(o, k, b, check_update=(kind == "lmi"))
    assert k.factor_regularized() == 0


@pytest.mark.parametrize("kind", ["lp", "lmi"])
def test_equality_constraints_with_supernodes_beyond_lds(kind):
    """Equality blocks on supernodes whose panel exceeds LDS (> ~140 columns): the LDLT kernel with
    its panel image in HBM -- same pivot rule (the largest |diagonal| of the whole trailing part,
    RLDLT.h:298-431), same operations as the LDS-resident one -- against the oracle."""
    rng = np.random.default_rng(77)
    if kind == "lp":
        nv = 180
        A = rng.uniform(-1, 1, (260, nv))
        c = np.abs(rng.uniform(0.5, 1.5, 260))
        eqs = [(rng.uniform(-1, 1, (4, nv)), rng.uniform(-1, 1, 4), None)]
        b = rng.uniform(-1, 1, nv)

        def build(cls, **kw):
            p = cls(nv, **kw)
            p.add_linear(A, c)
            for Ae, be, v in eqs:
                p.add_equality(Ae, be, v)
            p.initialize()
            return p
        W = None
    else:
        prob = syn.lmi_problem(K=3, n=24, m=170, branching=2, overlap=21, seed=170)
        nv, b = prob["num_vars"], prob["b"]
        eqs = [(rng.uniform(-1, 1, (2, 170)), rng.uniform(-0.1, 0.1, 2), prob["cliques"][0]),
               (rng.uniform(-1, 1, (1, 30)), rng.uniform(-0.1, 0.1, 1), prob["cliques"][2][-30:])]

        def build(cls, **kw):
            p = cls(nv, **kw)
            for ci, cl in enumerate(prob["cliques"]):
                p.add_lmi(prob["A"][ci], prob["C"][ci], cl)
            for Ae, be, v in eqs:
                p.add_equality(Ae, be, v)
            p.initialize()
            return p
        W = syn.scaling_points(3, 24, seed=5)
    o, k = build(ol.Program), build(KktContext, device=0)
    if W is not None:
        for i in range(len(W)):
            o.set_W(i, W[i])
            k.set_W(i, W[i])
    assert max(o.supernode_sizes()) > 140 and k.N == o.N > len(b)
    check_newton_step(o, k, b, check_update=False, inv_sqrt_mu=0.4 if kind == "lp" else 1.0)
    assert k.factor_regularized() == 0


@pytest.mark.parametrize("N", [2, 40])
def test_lqr_literal_through_the_hip_ldlt_path(N):
    """The reference's LQR KKT literals (assembly_test.cc:67-106 BuildLQRProblem, :108-169
    LDLT.TestAssembly, :171-194 LDLT.Benchmark2) through the HIP block-LDLT path: assembled slab and
    residuals equal to the oracle's exactly, three chained solves within 1e-9 of a dense solve."""
    from test_oracle_kat import LQR_B, build_lqr_problem, lqr_kkt_literal
    o = build_lqr_problem(ol.Program, N)
    k = build_lqr_problem(KktContext, N, device=0)
    assert k.N == o.N == (N + 1) * 3 + 2 * (N + 1)
    o.assemble()
    k.assemble()
    assert np.array_equal(blocks(k, k.slab()), blocks(o, o.slab()))     # constant blocks: exact
    _, AQo, _ = o.residuals()
    _, AQk, _ = k.residuals()
    assert np.array_equal(AQk, AQo)
    T = o.kkt_matrix()
    T = np.tril(T) + np.tril(T, -1).T
    if N == 2:
        assert np.array_equal(T, lqr_kkt_literal()) and np.array_equal(AQk, LQR_B)
    assert k.factor() == 1 and o.factor() == 1
    assert k.factor_regularized() == 0
    b = LQR_B.copy() if N == 2 else np.ones(k.N)
    for _ in range(3):
        y = k.solve_inplace(b)
        assert np.linalg.norm(y - np.linalg.solve(T, b)) <= 1e-9
        assert rel(y, o.solve_inplace(b)) <= TOL_DIRECTION
        if N == 2:
            b = y


def test_equality_multipliers_are_latched_by_prepare_step():
    o, k, b = eq_pair(5, "lp")
    y = np.random.default_rng(1).uniform(-1, 1, o.N)
    o.prepare_step(y, 0.3, 1.0)
    k.prepare_step(y, 0.3, 1.0)
    assert np.array_equal(k.get_W(1), o.get_W(1)) and len(k.get_W(1)) == 3


# --------------------------------------------------------------------- mixed Hermitian + SOC (C5, reduced)
def test_c5_mixed_hermitian_soc_newton_step():
    prob = syn.mixed_problem(K=230, seed=31)
    assert prob["kinds"].count("herm") == 80
    W = syn.mixed_scaling_points(prob, seed=32)
    o, k = make_pair(prob, "mixed", W)
    check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.5)


# --------------------------------------------------------------------- LP (C1)
def test_c1_lp_newton_steps():
    prob = syn.lp_problem(rows=20, num_vars=10)
    o, k = make_pair(prob, "lp")
    for it in range(4):
        check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.3 + 0.3 * it)


# --------------------------------------------------------------------- SOC (C3, reduced)
def test_c3_soc_newton_step():
    K = 300
    prob = syn.soc_problem(K=K, dim=10, m=10, overlap=2, seed=17)
    W = syn.soc_scaling_points(K, 10)
    o, k = make_pair(prob, "soc", W)
    check_newton_step(o, k, prob["b"])
    check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.9)


# --------------------------------------------------------------------- BASELINE configs 3 and 5 at FULL size
@pytest.mark.parametrize("tree,segments", [(0, "auto"), (0, "0"), (8, "auto")])
def test_c3_full_size_5000_soc(monkeypatch, tree, segments):
    """BASELINE config 3 as named: 5000 second-order cones of dimension 10.  tree = 0 is the
    reference-style chain (clique k = {8k .. 8k+9}: 5000 elimination levels, N = 40002), tree = 8
    SURVEY 8d's clique-tree variant.  Schur blocks / slab <= 1e-13, direction <= 1e-10, updates
    <= 1e-11 against the oracle, stage by stage.  The chain runs in both modes: the library's default
    (the factorization in its segment-parallel order, symbolic.h: everything but the stored factor is
    compared) and CXK_CHAIN_SEGMENTS=0 (the reference's order, the slab compared block by block)."""
    if segments == "auto":
        monkeypatch.delenv("CXK_CHAIN_SEGMENTS", raising=False)
    else:
        monkeypatch.setenv("CXK_CHAIN_SEGMENTS", segments)
    K = 5000
    prob = syn.soc_problem(K=K, dim=10, m=10, overlap=2, tree=tree)
    assert prob["num_vars"] == 40002
    W = syn.soc_scaling_points(K, 10)
    o, k = make_pair(prob, "soc", W)
    assert k.N == 40002
    assert k.chain_segments() == (2500 if (tree == 0 and segments == "auto") else 0)
    check_newton_step(o, k, prob["b"])


def test_c5_full_size_mixed_hermitian_soc():
    """BASELINE config 5 at full size: 4600 constraints (1600 complex Hermitian cones of order 12
    over 24 variables + 3000 second-order cones of dimension 10), N = 50004."""
    prob = syn.mixed_problem()
    assert prob["num_vars"] == 50004 and prob["kinds"].count("herm") == 1600
    W = syn.mixed_scaling_points(prob)
    o, k = make_pair(prob, "mixed", W)
    assert k.N == 50004
    check_newton_step(o, k, prob["b"], inv_sqrt_mu=0.5)


# --------------------------------------------------------------------- mixed cones
def test_mixed_cones_with_fill_in():
    """LMI + SOC + linear + constant block on the 4-cycle clique pattern (needs fill-in;
    test_lp.cc:230-315 structure)."""
    rng = np.random.default_rng(4)
    cliques = [[0, 1], [1, 2], [0, 3], [2, 3]]

    def build(cls, **kw):
        p = cls(4, **kw)
        A = rng0.uniform(-1, 1, (2, 3, 3))
        A = 0.5 * (A + np.transpose(A, (0, 2, 1)))
        p.add_lmi(A, np.eye(3), cliques[0])
        p.add_soc(rng0.uniform(-1, 1, (4, 2)), np.array([1.0, 0, 0, 0]), cliques[1])
        p.add_linear(rng0.uniform(-1, 1, (5, 2)), np.abs(rng0.uniform(0.5, 1, 5)), cliques[2])
        p.add_static(np.array([[2.0, 0.3], [0.3, 1.0]]), cliques[3])
        p.initialize()
        return p

    rng0 = np.random.default_rng(4)
    o = build(ol.Program)
    rng0 = np.random.default_rng(4)
    k = build(KktContext, device=0)
    b = rng.uniform(-1, 1, 4)
    check_newton_step(o, k, b)
    check_newton_step(o, k, b, inv_sqrt_mu=0.4)


# --------------------------------------------------------------------- headline shape (C4)
def test_c4_headline_direction_and_properties():
    """BASELINE config 4 at full size: 1000 LMIs of order 20, N = 15005."""
    prob = syn.lmi_problem()  # K=1000, n=20, m=20
    W = syn.scaling_points(1000, 20)
    k = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(k.K):
        k.set_W(i, W[i])
    assert k.N == 15005
    ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1
    # size-independent property 1: linearity of the solve in the right-hand side
    r1 = np.random.default_rng(1).uniform(-1, 1, k.N)
    r2 = np.random.default_rng(2).uniform(-1, 1, k.N)
    s1, s2 = k.solve_inplace(r1), k.solve_inplace(r2)
    s12 = k.solve_inplace(2.0 * r1 - 3.0 * r2)
    assert rel(s12, 2.0 * s1 - 3.0 * s2) <= 1e-10
    # property 2: G y = rhs, with G applied clique by clique from the per-constraint blocks
    k.assemble()
    AW, AQc, _ = k.residuals()
    rhs = 0.7 * (prob["b"] * 0.9 + AQc * 0.8) - 2 * AW
    Gy = np.zeros(k.N)
    for c, cl in enumerate(prob["cliques"]):
        G, _, _, _ = k.constraint_schur(c)
        G = np.tril(G) + np.tril(G, -1).T
        Gy[cl] += G @ y[cl]
    assert rel(Gy, rhs) <= 1e-9
    # property 3: agreement with the oracle on the Newton direction (the bar: 1e-10)
    o = syn.build(ol.Program, prob, "lmi")
    for i in range(o.K):
        o.set_W(i, W[i])
    ok_o, yo = o.kkt_solve(np.concatenate([prob["b"]]), 0.7, 0.9, 0.8)
    assert ok_o == 1
    assert rel(y, yo) <= TOL_DIRECTION
    # repeatability: the pull formulation has no atomics, so two runs agree bit for bit
    _, y2 = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("K,n,m,b_,ov,cols", [(1, 52, 40, 2, 1, 40), (1, 60, 64, 2, 1, 64), (3, 45, 50, 2, 20, 50),
                                              (30, 20, 20, 8, 5, 0)])
def test_dense_top_range(K, n, m, b_, ov, cols):
    """Last levels with a 33..64-column supernode run as one dense register factorization
    (kernels_kkt_top.hip.h); tops made of small supernodes (C4-like, cols == 0) do not."""
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=b_, overlap=ov, seed=300 + m)
    W = syn.scaling_points(K, n, seed=3 + n)
    o, k = make_pair(prob, "lmi", W)
    assert k.dense_top_columns() == cols
    check_newton_step(o, k, prob["b"], lanczos_tol=1e-7 if n > 32 else None)
