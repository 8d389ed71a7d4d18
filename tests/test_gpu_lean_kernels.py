"""The shape-specialised tree kernels (tree_factor_level / tree_backward_level and their two-shape
forms) restate the generic sweep with a straight-line load phase: same operations
in the same order, so their results must equal the generic kernels' BIT FOR BIT.  CXK_NO_LEAN=1
(read when a context is initialized) selects the generic kernels for the comparison; both paths
are also held against the oracle by the other GPU tests."""
import os

import numpy as np
import pytest

from conex_amd import KktContext
from conex_amd import synthetic as syn
from test_gpu_random_structures import build as build_random
from test_gpu_random_structures import random_program

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def level_kernels_only():
    """This module is about the LEVEL kernels: the whole-tree launch, which takes over where a tree
    allows it, has its own comparison against them (test_gpu_fused_tree.py)."""
    os.environ["CXK_NO_FUSED_TREE"] = "1"
    yield
    os.environ.pop("CXK_NO_FUSED_TREE", None)


def both_paths(make):
    """-> (lean context, generic context), same program."""
    os.environ.pop("CXK_NO_LEAN", None)
    lean = make()
    os.environ["CXK_NO_LEAN"] = "1"
    try:
        generic = make()
    finally:
        os.environ.pop("CXK_NO_LEAN", None)
    return lean, generic


def solve_twice(k, b):
    k.set_cost(b)
    out = []
    for mu in (0.7, 0.4):
        k.kkt_solve_async(mu, 0.9, 0.8)
        assert k.sync()
        out.append((k.get_y().copy(), k.slab().copy()))
        k.solve_rhs(-0.9, 0.8, 0.0)          # solve-only path on the stored factor
        assert k.sync()
        out.append((k.get_y().copy(), None))
    return out


def assert_same_bits(lean, generic, b):
    for (y1, s1), (y2, s2) in zip(solve_twice(lean, b), solve_twice(generic, b)):
        assert np.array_equal(y1, y2)
        if s1 is not None:
            assert np.array_equal(s1, s2, equal_nan=True)  # never-written padding may hold NaN under CXK_DEBUG_FILL_NAN


@pytest.mark.parametrize("K,branching", [(100, 8), (73, 3), (40, 1)])
def test_lmi_tree_lean_equals_generic(K, branching):
    prob = syn.lmi_problem(K=K, n=20, m=20, branching=branching, overlap=5, seed=11)
    W = syn.scaling_points(K, 20, seed=12)

    def make():
        k = syn.build(KktContext, prob, "lmi", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        return k

    lean, generic = both_paths(make)
    # a clique tree's leaves take their panels straight from the Schur blocks (the assembly rides in
    # the first factor level's launch); the generic path assembles in a launch of its own
    assert lean.fused_assembly() == (branching > 1) and not generic.fused_assembly()
    assert_same_bits(lean, generic, prob["b"])


def test_fused_assembly_equals_separate_assembly_launch():
    """tree_factor_level_asm against the same lean kernels behind assemble_gather
    (CXK_NO_FUSED_ASM=1): direction, factor, AW / AQc (through the step scalars) bit for bit, and a
    failed pivot in the first level is reported."""
    prob = syn.lmi_problem(K=230, n=20, m=20, branching=8, overlap=5, seed=21)
    W = syn.scaling_points(230, 20, seed=22)

    def make():
        k = syn.build(KktContext, prob, "lmi", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        return k

    fused = make()
    os.environ["CXK_NO_FUSED_ASM"] = "1"
    try:
        plain = make()
    finally:
        os.environ.pop("CXK_NO_FUSED_ASM", None)
    assert fused.fused_assembly() and not plain.fused_assembly()
    assert_same_bits(fused, plain, prob["b"])
    for k in (fused, plain):
        k.kkt_solve_async(0.7, 0.9, 0.8)
    assert np.array_equal(fused.step_scalars(), plain.step_scalars())
    # the interior-point loop's order of calls: cxk_assemble leaves the gather to the factor-and-solve
    # entry point that follows (right-hand side in either form); any other call in between takes
    # the separate launch.  Same bits every way.
    for call in ("factor_solve", "factor_direction", "scalars_first"):
        ys = []
        for k in (fused, plain):
            k.assemble()
            if call == "scalars_first":
                sc = k.step_scalars()          # needs AW / AQc of the assembled system: flushes the gather
            if call == "factor_direction":
                k._check(k.L.cxk_factor_direction_async(k.h, 0.7, 0.9, 0.8), "factor_direction")
            else:
                k._check(k.L.cxk_factor_solve_async(k.h, -0.9, 0.8, 0.0), "factor_solve")
            assert k.sync()
            ys.append((k.get_y().copy(), k.slab().copy(), k.step_scalars()))
        assert np.array_equal(ys[0][0], ys[1][0]) and np.array_equal(ys[0][1], ys[1][1], equal_nan=True)
        assert np.array_equal(ys[0][2], ys[1][2])
    # an indefinite scaling point on a leaf: the factorization must report failure on both paths
    bad = W[prob_leaf(fused)].copy()
    bad[0, 0] = -1e3
    for k in (fused, plain):
        k.set_W(prob_leaf(fused), bad)
        ok, _ = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        assert ok == 0
        k.set_W(prob_leaf(fused), W[prob_leaf(fused)])
        ok, _ = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        assert ok == 1


def prob_leaf(k):
    """A constraint eliminated at a leaf of the tree: the last one of an 8-ary tree in BFS order."""
    return k.K - 1


def test_soc_tree_lean_equals_generic():
    prob = syn.soc_problem(K=300, dim=10, m=10, overlap=2, tree=8)
    W = syn.soc_scaling_points(300, 10)

    def make():
        k = syn.build(KktContext, prob, "soc", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        return k

    lean, generic = both_paths(make)
    assert_same_bits(lean, generic, prob["b"])


def test_mixed_shapes_lean_equals_generic():
    # Hermitian order-12 cones (shape <24,8>) and second-order cones (<8,8>) on every level:
    # the two-shape launches
    prob = syn.mixed_problem(K=230)
    W = syn.mixed_scaling_points(prob)

    def make():
        k = syn.build(KktContext, prob, "mixed", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        return k

    lean, generic = both_paths(make)
    assert_same_bits(lean, generic, prob["b"])


@pytest.mark.parametrize("seed", range(12))
def test_random_structures_lean_equals_generic(seed):
    # ragged cliques: levels split into several segments, some without a lean kernel
    prob = random_program(2000 + seed)
    lean, generic = both_paths(lambda: build_random(KktContext, prob, device=0))
    assert_same_bits(lean, generic, prob["b"])


def test_backward_pair_launch_equals_level_by_level(monkeypatch):
    """tree_backward_pair (two downward levels in one launch, children behind a workgroup barrier)
    returns the bits of the level-by-level sweep: 20 solves each on two shapes, every one compared
    (the first version read the parent's solution through the scalar cache and was stale about one
    time in five)."""
    for K, n, m, br, ov in ((300, 8, 6, 4, 2), (120, 12, 9, 8, 3)):
        prob = syn.lmi_problem(K=K, n=n, m=m, branching=br, overlap=ov, seed=5 + K)
        W = syn.scaling_points(K, n, seed=6 + K)
        ys = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("CXK_NO_BACK_PAIRS", "1")
            else:
                monkeypatch.delenv("CXK_NO_BACK_PAIRS", raising=False)
            k = syn.build(KktContext, prob, "lmi", device=0)
            for i in range(k.K):
                k.set_W(i, W[i])
            out = []
            for _ in range(20):
                ok, y = k.kkt_solve(prob["b"], 0.5, 0.9, 0.8)
                assert ok == 1
                out.append(y.copy())
            ys.append(out)
        for a in ys[0] + ys[1]:
            assert np.array_equal(a, ys[1][0])
