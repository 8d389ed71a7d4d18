"""Chain-shaped elimination trees in a segment-parallel order (conex_amd/csrc/symbolic.h,
SegmentChain; BASELINE config 3 as the reference's tests arrange it).

CPU part: what the library REPORTS stays the reference's structure (bit-equal to the oracle's literal
restatement of clique_ordering.cc / supernodal_solver.cc), and the structure the factorization RUNS on
is a valid supernodal structure of the same matrix in another elimination order -- checked by running
the oracle's own block Cholesky and block solves (block_triangular_operations.cc:114-219 restated) on
it and comparing with a dense solve.
GPU part: the Newton direction, the residuals and the cone updates of the segmented factorization
against the oracle, which eliminates in the reference's order.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import KktContext
from conex_amd import synthetic as syn


def chain_context(prob, segments, device=-1):
    k = KktContext(prob["num_vars"], device=device)
    for c, cl in enumerate(prob["cliques"]):
        k.add_soc(prob["A"][c], prob["c"][c], cl)
    k.set_chain_segments(segments)
    k.initialize()
    return k


@pytest.mark.parametrize("K,segments", [(40, 4), (64, 8), (300, 16), (33, 5)])
def test_reported_structure_stays_the_reference(K, segments):
    prob = syn.soc_problem(K=K, dim=10, m=10, overlap=2)
    k = chain_context(prob, segments)
    assert k.chain_segments() == segments
    o = syn.build(ol.Program, prob, "soc")
    assert np.array_equal(k.order(), o.order())
    pk, qk = k.permutation()
    po, qo = o.permutation()
    assert np.array_equal(pk, po) and np.array_equal(qk, qo)
    assert np.array_equal(k.supernode_sizes(), o.supernode_sizes())
    for e in range(K):
        for which in range(5):
            assert np.array_equal(k.get_list(which, e), o.get_list(which, e)), (which, e)
    dk, fk = k.block_offsets()
    do, fo = o.block_offsets()
    assert np.array_equal(dk, do) and np.array_equal(fk, fo)


def internal_structure(k):
    K = k.K
    sizes = k.get_list(20, 0)
    paths = [list(k.get_list(10, e)) for e in range(K)]
    perm = k.get_list(21, 0)          # original variable -> eliminated position
    return sizes, paths, perm


@pytest.mark.parametrize("K,segments,overlap,m", [(40, 4, 2, 10), (64, 8, 2, 10), (30, 3, 3, 9), (120, 12, 1, 6)])
def test_segmented_structure_factors_the_same_matrix(K, segments, overlap, m):
    rng = np.random.default_rng(K + segments)
    prob = syn.soc_problem(K=K, dim=m, m=m, overlap=overlap)
    k = chain_context(prob, segments)
    assert k.chain_segments() == segments
    sizes, paths, perm = internal_structure(k)
    N = prob["num_vars"]
    assert sorted(perm) == list(range(N)) and int(np.sum(sizes)) == N
    # the pieces are independent subtrees of K / segments steps, the deferred sets merge pairwise above
    # them: K / segments + log2(segments) dependent levels instead of K
    owner = np.zeros(N, dtype=int)
    start = np.concatenate([[0], np.cumsum(sizes)])
    for e in range(K):
        owner[start[e]:start[e + 1]] = e
    level = np.zeros(K, dtype=int)
    for e in range(K):
        for v in paths[e][sizes[e]:]:
            level[owner[v]] = max(level[owner[v]], level[e] + 1)
    assert level.max() + 1 <= -(-K // segments) + int(np.ceil(np.log2(segments))) + 1
    # a random SPD matrix with the program's sparsity: sum of P_c' G_c P_c, G_c = R R' + I
    M = np.zeros((N, N))
    for cl in prob["cliques"]:
        R = rng.uniform(-1, 1, (len(cl), len(cl)))
        M[np.ix_(cl, cl)] += R @ R.T + np.eye(len(cl))
    inv = np.argsort(perm)             # eliminated position -> original variable
    Mi = M[np.ix_(inv, inv)]
    ws = ol.Workspace(paths, list(sizes))
    assert ws.N == N
    for e in range(K):
        ns = sizes[e]
        rows = paths[e][:ns]
        sep = paths[e][ns:]
        ws.diag(e)[:, :] = Mi[np.ix_(rows, rows)]
        if sep:
            ws.offd(e)[:, :] = Mi[np.ix_(rows, sep)]
    # every nonzero of the matrix has a place in the structure
    D = ws.to_dense()
    assert np.allclose(np.tril(D), np.tril(Mi), atol=0)
    assert ws.cholesky() == 1
    b = rng.uniform(-1, 1, N)
    y = ws.backward(ws.forward(b))
    assert np.linalg.norm(y - np.linalg.solve(Mi, b)) <= 1e-11 * np.linalg.norm(y)


def test_not_a_chain_is_left_alone():
    prob = syn.soc_problem(K=60, dim=10, m=10, overlap=2, tree=3)
    k = KktContext(prob["num_vars"], device=-1)
    for c, cl in enumerate(prob["cliques"]):
        k.add_soc(prob["A"][c], prob["c"][c], cl)
    k.set_chain_segments(4)
    k.initialize()
    assert k.chain_segments() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("K,segments", [(64, 4), (400, 16), (1000, 0)])
def test_segmented_chain_newton_step_matches_the_oracle(K, segments):
    """segments = 0 here means: leave the choice to the library (automatic from 256 steps)."""
    from test_gpu_parity import rel
    prob = syn.soc_problem(K=K, dim=10, m=10, overlap=2)
    W = syn.soc_scaling_points(K, 10)
    k = KktContext(prob["num_vars"], device=0)
    for c, cl in enumerate(prob["cliques"]):
        k.add_soc(prob["A"][c], prob["c"][c], cl)
    if segments:
        k.set_chain_segments(segments)
    k.initialize()
    assert k.chain_segments() == (segments if segments else K // 2)
    o = syn.build(ol.Program, prob, "soc")
    for i in range(K):
        k.set_W(i, W[i])
        o.set_W(i, W[i])
    ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    oko, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1 and oko == 1
    assert rel(y, yo) <= 1e-10
    AWk, AQk, sck = k.residuals()
    AWo, AQo, sco = o.residuals()
    assert rel(AWk, AWo) <= 1e-13 and rel(AQk, AQo) <= 1e-13 and rel(sck, sco) <= 1e-12
    # solve-only sweeps and the cone updates run on the same factor
    rhs = np.random.default_rng(3).uniform(-1, 1, k.N)
    assert rel(k.solve_inplace(rhs), o.solve_inplace(rhs)) <= 1e-10
    io = o.prepare_step(yo, 0.56, 1.0)
    ik = k.prepare_step(yo, 0.56, 1.0)
    assert rel(ik, io) <= 1e-9
    step = min(1.0, 2.0 / io[1] ** 2)
    o.take_step(step)
    k.take_step(step)
    for i in range(0, K, max(1, K // 9)):
        assert rel(k.get_W(i), o.get_W(i)) <= 1e-11
    with pytest.raises(Exception):
        k.slab()                       # the stored factor is not in the reference's layout


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,overlap", [(6, 8, 3), (20, 20, 5)])
def test_segmented_lmi_chain_matches_the_oracle(n, m, overlap):
    """A chain of matrix inequalities (branching 1: every clique shares `overlap` variables with the next):
    deferred sets of `overlap` variables, supernodes of other shapes than config 3's."""
    from test_gpu_parity import rel
    K = 300
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=1, overlap=overlap, seed=5)
    W = syn.scaling_points(K, n, seed=6)
    k = syn.build(KktContext, prob, "lmi", device=0)
    assert k.chain_segments() == K // 2
    o = syn.build(ol.Program, prob, "lmi")
    for i in range(K):
        k.set_W(i, W[i])
        o.set_W(i, W[i])
    ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    oko, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1 and oko == 1
    assert rel(y, yo) <= 1e-10
    k.newton_direction(0.9, 0.8, 0.7)             # a solve-only sweep on the stored factor
    rhs = 0.9 * (np.concatenate([prob["b"], np.zeros(o.N - len(prob["b"]))]) * 0.8 + o.residuals()[1] * 0.7) - 2 * o.residuals()[0]
    assert rel(k.get_y(), o.solve_inplace(rhs)) <= 1e-10


@pytest.mark.gpu
def test_conex_maximize_on_a_chain_in_both_orders(monkeypatch):
    """The whole interior-point solve of a chain-shaped program through conex.h (matrix inequalities
    added with their variable lists: CONEX_AddSparseLMIConstraint): the segment-parallel factorization and
    the reference's order reach the same optimum, which is the oracle's."""
    import ctypes as C
    import conex_api as ca
    K, n, m = 300, 6, 8
    prob = syn.lmi_problem(K=K, n=n, m=m, branching=1, overlap=3, seed=9)
    L = ca.api()

    def solve(segments):
        if segments is None:
            monkeypatch.delenv("CXK_CHAIN_SEGMENTS", raising=False)
        else:
            monkeypatch.setenv("CXK_CHAIN_SEGMENTS", str(segments))
        p = L.CONEX_CreateConeProgram()
        assert L.CONEX_SetNumberOfVariables(p, prob["num_vars"]) == 0
        for c, cl in enumerate(prob["cliques"]):
            a, cm = ca.colmajor(prob["A"][c]), ca.colmajor(prob["C"][c])
            v = np.ascontiguousarray(cl, dtype=np.int64)
            assert L.CONEX_AddSparseLMIConstraint(p, ca.dp(a), n, n, m, ca.dp(cm), n, n,
                                                  v.ctypes.data_as(C.POINTER(C.c_long)), m) == c
        cfg = ca.default_config()
        b = np.ascontiguousarray(prob["b"])
        y = np.zeros(len(b))
        ok = L.CONEX_Maximize(p, ca.dp(b), len(b), C.byref(cfg), ca.dp(y), len(b))
        L.CONEX_DeleteConeProgram(p)
        return ok, y

    ok_seg, y_seg = solve(None)
    ok_ref, y_ref = solve(0)
    o = syn.build(ol.Program, prob, "lmi")
    oko, yo = o.solve(prob["b"])
    assert ok_seg == 1 and ok_ref == 1 and oko == 1
    assert abs(prob["b"] @ y_seg - prob["b"] @ yo) <= 1e-6 * abs(prob["b"] @ yo)
    assert np.linalg.norm(y_seg - y_ref) <= 1e-3 * np.linalg.norm(y_ref)
    assert np.linalg.norm(y_seg - yo) <= 1e-3 * np.linalg.norm(yo)
