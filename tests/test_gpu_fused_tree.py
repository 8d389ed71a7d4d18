"""The whole elimination tree in one launch (tree_fused.hip): assembly gather,
supernodal Cholesky with the first right-hand side, back substitution -- one wavefront per
supernode, published values handed upward / downward through sentinel-armed slots.

It restates the level kernels' arithmetic in their order, so factor, AW / AQc and the two scalars
must equal the level-by-level path's (CXK_NO_FUSED_TREE=1, read when a context is initialized) BIT
FOR BIT, the direction to rounding (see assert_same); the solve-only sweep is bit-identical again.
Both paths are held against the oracle elsewhere (test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import KktContext
from conex_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def both_paths(make):
    os.environ.pop("CXK_NO_FUSED_TREE", None)
    fused = make()
    os.environ["CXK_NO_FUSED_TREE"] = "1"
    try:
        levels = make()
    finally:
        os.environ.pop("CXK_NO_FUSED_TREE", None)
    assert fused.fused_tree() and not levels.fused_tree()
    return fused, levels


def snapshot(k):
    AW, AQc, sc = k.residuals()
    return k.get_y().copy(), k.slab().copy(), AW, AQc, sc


def assert_same(a, b):
    """factor, AW / AQc and the two scalars bit for bit; the direction to rounding (the fused launch
    solves the back substitution's 1 + |separator| right-hand sides ahead of the separator's solution
    and combines them when it arrives: same solution, another order of operations)"""
    assert np.linalg.norm(a[0] - b[0]) <= 1e-13 * np.linalg.norm(b[0])
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("K,n,branching,overlap", [(100, 20, 8, 5), (73, 20, 3, 5), (30, 20, 1, 5),
                                                   (200, 8, 8, 4), (41, 12, 2, 7), (1, 20, 8, 5)])
def test_fused_tree_equals_level_kernels(K, n, branching, overlap):
    prob = syn.lmi_problem(K=K, n=n, m=20, branching=branching, overlap=overlap, seed=31)
    W = syn.scaling_points(K, n, seed=32)

    def make():
        k = syn.build(KktContext, prob, "lmi", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
        return k

    fused, levels = both_paths(make)
    # several runs: the hand-off slots alternate between their two sets
    for mu in (0.7, 0.4, 0.9, 0.55, 0.61):
        for k in (fused, levels):
            k.kkt_solve_async(mu, 0.9, 0.8)
            assert k.sync()
        assert_same(snapshot(fused), snapshot(levels))
    # the interior-point loop's order: assemble (gather deferred), then factor with the first solve
    for k in (fused, levels):
        k.assemble()
        k.factor_solve_async(-0.9, 0.8, 0.0)
        assert k.sync()
    assert_same(snapshot(fused), snapshot(levels))
    # a solve-only sweep on the factor the fused launch stored
    for k in (fused, levels):
        k.solve_rhs(0.3, -0.2, 1.5)
        assert k.sync()
    assert np.array_equal(fused.get_y(), levels.get_y())


@pytest.mark.parametrize("K,n,branching,overlap", [(100, 20, 8, 5), (41, 12, 2, 7)])
def test_fused_tree_as_two_launches_equals_level_kernels(K, n, branching, overlap):
    """A tree with more supernodes than the chip holds wavefronts takes the way up and the way down
    as two launches (CXK_FUSED_SPLIT=1 forces that on a small tree): same results."""
    prob = syn.lmi_problem(K=K, n=n, m=20, branching=branching, overlap=overlap, seed=41)
    W = syn.scaling_points(K, n, seed=42)

    def make():
        k = syn.build(KktContext, prob, "lmi", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
        return k

    os.environ["CXK_FUSED_SPLIT"] = "1"
    try:
        fused, levels = both_paths(make)
    finally:
        os.environ.pop("CXK_FUSED_SPLIT", None)
    for mu in (0.7, 0.4, 0.9):
        for k in (fused, levels):
            k.kkt_solve_async(mu, 0.9, 0.8)
            assert k.sync()
        a, b = snapshot(fused), snapshot(levels)
        for x, y in zip(a, b):     # (the two-launch form sweeps back down as the level kernels do: same bits)
            assert np.array_equal(x, y, equal_nan=True)
        for k in (fused, levels):
            k.solve_rhs(0.3, -0.2, 1.5)
            assert k.sync()
        assert np.array_equal(fused.get_y(), levels.get_y())


@pytest.mark.parametrize("K,branching", [(1000, 8), (300, 3), (64, 1)])
def test_dealing_the_supernodes_per_xcd_changes_no_bit(K, branching, monkeypatch):
    """Which workgroup takes which supernode is a matter of speed only (depth-first order, an eighth per XCD:
    kkt_plans.hip; CXK_FUSED_LEVEL_ORDER=1 when the plans are built keeps the level order): direction, factor,
    residuals and scalars are the same bits either way, over several launches (both slot sets in use)."""
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=branching, overlap=5, seed=7 + K)
    W = syn.scaling_points(K, 20, seed=33)

    def make():
        k = syn.build(KktContext, prob, "lmi", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
        return k

    dealt = make()
    monkeypatch.setenv("CXK_FUSED_LEVEL_ORDER", "1")
    level = make()
    monkeypatch.delenv("CXK_FUSED_LEVEL_ORDER")
    assert dealt.fused_tree() and level.fused_tree()
    for rep in range(3):
        for k in (dealt, level):
            k.kkt_solve_async(0.7 + 0.1 * rep, 0.9, 0.8)
            assert k.sync()
        a, b = snapshot(dealt), snapshot(level)
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)


def test_fused_tree_against_the_oracle_and_two_runs_agree():
    prob = syn.lmi_problem(K=150, n=20, m=20, branching=8, overlap=5, seed=5)
    W = syn.scaling_points(150, 20, seed=6)
    k = syn.build(KktContext, prob, "lmi", device=0)
    o = syn.build(ol.Program, prob, "lmi")
    for i in range(k.K):
        k.set_W(i, W[i])
        o.set_W(i, W[i])
    assert k.fused_tree()
    ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    oko, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1 and oko == 1
    assert np.linalg.norm(y - yo) <= 1e-10 * np.linalg.norm(yo)
    ok2, y2 = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok2 == 1 and np.array_equal(y, y2)


def test_fused_tree_reports_a_failed_pivot_and_recovers():
    prob = syn.lmi_problem(K=120, n=20, m=20, branching=8, overlap=5, seed=7)
    W = syn.scaling_points(120, 20, seed=8)
    k = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(k.K):
        k.set_W(i, W[i])
    assert k.fused_tree()
    ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1
    for bad in (k.K - 1, 0, 17):                 # a leaf, the root's constraint, one in between
        k.set_W(bad, np.zeros((20, 20)))         # G = 0: a pivot that is not positive
        okb, _ = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        assert okb == 0
        k.set_W(bad, W[bad])
        ok2, y2 = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        assert ok2 == 1 and np.array_equal(y, y2)


def test_fused_tree_on_the_mixed_program_shapes():
    """Second-order cones (shape <8,8>) next to matrix cones: two register shapes in one launch."""
    prob = syn.mixed_problem(K=230, seed=3)
    W = syn.mixed_scaling_points(prob, seed=32)

    def make():
        k = syn.build(KktContext, prob, "mixed", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        k.set_cost(prob["b"])
        return k

    os.environ.pop("CXK_NO_FUSED_TREE", None)
    fused = make()
    assert fused.fused_tree()
    os.environ["CXK_NO_FUSED_TREE"] = "1"
    try:
        levels = make()
    finally:
        os.environ.pop("CXK_NO_FUSED_TREE", None)
    for mu in (0.7, 0.4, 0.9):
        for k in (fused, levels):
            k.kkt_solve_async(mu, 0.9, 0.8)
            assert k.sync()
        assert_same(snapshot(fused), snapshot(levels))


def test_a_timed_out_wait_falls_back_to_the_level_kernels():
    """A wait inside the whole-tree launch that runs out (a device shared with other work) is not a
    failed factorization: cxk_sync redoes the factor-and-solve on the level kernels, reports success,
    and the context stays on them (include/conex_kkt_hip.h, cxk_fused_tree_timed_out)."""
    prob = syn.lmi_problem(K=60, n=20, m=20, branching=4, overlap=5, seed=41)
    W = syn.scaling_points(60, 20, seed=42)
    k = syn.build(KktContext, prob, "lmi", device=0)
    o = syn.build(ol.Program, prob, "lmi")
    for i in range(k.K):
        k.set_W(i, W[i])
        o.set_W(i, W[i])
    k.set_cost(prob["b"])
    assert k.fused_tree()
    k.kkt_solve_async(0.7, 0.9, 0.8)
    assert k.sync() == 1
    y_fused = k.get_y().copy()
    k.kkt_solve_async(0.7, 0.9, 0.8)
    k.debug_force_fused_timeout()          # as if that launch had reported a wait that ran out
    assert k.sync() == 1                   # redone level by level, not reported as a failure
    assert not k.fused_tree()
    oko, yo = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert oko == 1
    assert np.linalg.norm(k.get_y() - yo) <= 1e-10 * np.linalg.norm(yo)
    assert np.linalg.norm(k.get_y() - y_fused) <= 1e-12 * np.linalg.norm(y_fused)
    k.kkt_solve_async(0.7, 0.9, 0.8)       # and the context keeps working on the level kernels
    assert k.sync() == 1
    assert np.linalg.norm(k.get_y() - yo) <= 1e-10 * np.linalg.norm(yo)


def test_a_timed_out_wait_inside_conex_maximize_redoes_the_iteration(monkeypatch):
    """program.cc: a time-out reported by cxk_factor_status makes the interior-point loop redo the
    iteration (W is untouched: TakeStep looked at the flag) instead of ending with 'Factorization
    failed'.  The hook arms the time-out through the environment for the second iteration."""
    import ctypes as C
    import conex_api as ca
    prob = syn.lmi_problem(K=40, n=20, m=20, branching=3, overlap=5, seed=43)
    L = ca.api()

    def solve(force_at):
        if force_at is None:
            monkeypatch.delenv("CXK_DEBUG_FUSED_TIMEOUT_AT", raising=False)
        else:
            monkeypatch.setenv("CXK_DEBUG_FUSED_TIMEOUT_AT", str(force_at))
        p = L.CONEX_CreateConeProgram()
        assert L.CONEX_SetNumberOfVariables(p, prob["num_vars"]) == 0
        for c, cl in enumerate(prob["cliques"]):
            a, cm = ca.colmajor(prob["A"][c]), ca.colmajor(prob["C"][c])
            v = np.ascontiguousarray(cl, dtype=np.int64)
            assert L.CONEX_AddSparseLMIConstraint(p, ca.dp(a), 20, 20, 20, ca.dp(cm), 20, 20,
                                                  v.ctypes.data_as(C.POINTER(C.c_long)), 20) == c
        cfg = ca.default_config()
        b = np.ascontiguousarray(prob["b"])
        y = np.zeros(len(b))
        ok = L.CONEX_Maximize(p, ca.dp(b), len(b), C.byref(cfg), ca.dp(y), len(b))
        st = ca.IterationStats()
        L.CONEX_GetIterationStats(p, C.byref(st), -1)
        L.CONEX_DeleteConeProgram(p)
        return ok, y, st.iteration_number + 1

    ok0, y0, it0 = solve(None)
    ok1, y1, it1 = solve(2)
    assert ok0 == 1 and ok1 == 1          # (without the redo: "Factorization failed", ok1 == 0)
    # from the redone iteration on the sweeps are the level kernels' (direction equal to 1e-15, not bit for
    # bit), and the raw Lanczos estimates amplify that near convergence: same optimum, +- an iteration
    assert abs(it1 - it0) <= 2
    assert abs(prob["b"] @ y1 - prob["b"] @ y0) <= 1e-6 * abs(prob["b"] @ y0)
    assert np.linalg.norm(y1 - y0) <= 1e-3 * np.linalg.norm(y0)


def test_kernel_clocks_report_the_launches_of_a_step():
    prob = syn.lmi_problem(K=100, n=20, m=20, branching=8, overlap=5, seed=31)
    W = syn.scaling_points(100, 20, seed=32)
    k = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(k.K):
        k.set_W(i, W[i])
    k.set_cost(prob["b"])
    k.kkt_solve_async(0.7, 0.9, 0.8)
    assert k.sync() == 1
    k.enable_timing(1)
    for _ in range(5):
        k.kkt_solve_async(0.7, 0.9, 0.8)
    assert k.sync() == 1
    k.enable_timing(False)
    na, ta = k.kernel_clock("assembly")
    nt, tt = k.kernel_clock("tree")
    assert na == 5 and nt == 5
    assert 1e-3 < ta < 1.0 and 1e-3 < tt < 1.0     # milliseconds: a few microseconds up to a millisecond
    assert k.kernel_clock("prepare")[0] == 0
