"""A defect of the reference's assembly: reproduced by default (reference identity), corrected
behind cxk_set_reference_identity(ctx, 0) / CXK_REFERENCE_QUIRKS=0.

BindDiagonalBlock (supernodal_assembler.cc:72-91) aliases a constraint's Schur block G onto its
supernode's diagonal block ("direct_update") when the supernode has as many variables as the
constraint and their positions inside the constraint are strictly increasing -- without checking
that the first position is 0.  When the running-intersection fix (clique_ordering.cc:261-305) has
put a variable the constraint does not contain at the front of that supernode (position -1,
followed by 0, 1, .., m-2) the test passes and G lands one row and column off; the constraint's
last variable, which sits in the separator, never reaches its place.  The assembled matrix is then
not sum_c P_c^T G_c P_c.

The oracle restates the reference as written (so it has the defect) and offers the corrected test
behind cxo_set_strict_direct_update(1).  The HIP path does the same behind its own switch: as
written by default, every block scattered by position with reference identity off.  Found by the
randomised structure sweep (seed 1657 of tests/test_gpu_random_structures.py: 54 mixed cones,
N = 715; the first such case in 700 random programs).
"""
import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_random_structures import build, random_program, rel

SEED = 1657


def dense_from_slab(o, slab):
    N = o.N
    sizes = o.supernode_sizes()
    dof, oof = o.block_offsets()
    K = np.zeros((N, N))
    for e in range(o.K):
        lst = o.get_list(0, e)
        ns = int(sizes[e])
        nsep = len(lst) - ns
        D = slab[dof[e]:dof[e] + ns * ns].reshape(ns, ns).T
        for a in range(ns):
            for b in range(a + 1):
                K[lst[a], lst[b]] = K[lst[b], lst[a]] = D[a, b]
        if ns * nsep:
            B = slab[oof[e]:oof[e] + ns * nsep].reshape(nsep, ns).T
            for a in range(ns):
                for b in range(nsep):
                    K[lst[a], lst[ns + b]] = K[lst[ns + b], lst[a]] = B[a, b]
    return K


def scattered_blocks(o, prob):
    """sum_c P_c^T G_c P_c from the per-constraint Schur blocks."""
    p, _ = o.permutation()
    K = np.zeros((o.N, o.N))
    for c, cl in enumerate(prob["cliques"]):
        G, _, _, _ = o.constraint_schur(c)
        G = np.tril(G) + np.tril(G, -1).T
        lab = [int(p[v]) for v in cl]
        K[np.ix_(lab, lab)] += G
    return K


@pytest.fixture
def strict():
    ol.lib().cxo_set_strict_direct_update(1)
    yield
    ol.lib().cxo_set_strict_direct_update(0)


def test_reference_as_written_misplaces_a_block():
    prob = random_program(SEED)
    o = build(ol.Program, prob)
    o.assemble()
    K_ref = dense_from_slab(o, o.slab())
    K_sum = scattered_blocks(o, prob)
    bad = np.argwhere(np.abs(np.tril(K_ref - K_sum)) > 1e-9)
    assert len(bad) > 0                                   # the defect is there ...
    assert len({int(i) for i, _ in bad} | {int(j) for _, j in bad}) <= 4   # ... in one 3 + 1 variable supernode


def test_corrected_test_restores_the_sum_of_blocks(strict):
    prob = random_program(SEED)
    o = build(ol.Program, prob)
    o.assemble()
    assert np.abs(dense_from_slab(o, o.slab()) - scattered_blocks(o, prob)).max() <= 1e-12


@pytest.mark.gpu
def test_hip_path_with_the_correction_equals_the_corrected_reference(strict, monkeypatch):
    from conex_amd import KktContext
    monkeypatch.setenv("CXK_REFERENCE_QUIRKS", "0")
    prob = random_program(SEED)
    o, k = build(ol.Program, prob), build(KktContext, prob, device=0)
    o.assemble()
    k.assemble()
    K_sum = scattered_blocks(o, prob)
    assert np.abs(dense_from_slab(o, k.slab()) - K_sum).max() <= 1e-12
    assert np.abs(dense_from_slab(o, o.slab()) - K_sum).max() <= 1e-12
    ok_o, yo = o.kkt_solve(prob["b"], 0.3, 0.9, 0.8)
    k.set_cost(prob["b"])
    k.kkt_solve_async(0.3, 0.9, 0.8)
    assert ok_o == 1 and k.sync()
    assert rel(k.get_y(), yo) <= 1e-10


@pytest.mark.gpu
def test_corrected_hip_path_differs_from_the_reference_as_written_here(monkeypatch):
    from conex_amd import KktContext
    monkeypatch.setenv("CXK_REFERENCE_QUIRKS", "0")
    prob = random_program(SEED)
    o, k = build(ol.Program, prob), build(KktContext, prob, device=0)
    ok_o, yo = o.kkt_solve(prob["b"], 0.3, 0.9, 0.8)
    k.set_cost(prob["b"])
    k.kkt_solve_async(0.3, 0.9, 0.8)
    assert ok_o == 1 and k.sync()
    assert rel(k.get_y(), yo) > 1e-4      # what the correction is for: the reference's direction is off here


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [SEED, 2585])
def test_default_mode_reproduces_the_reference_as_written(monkeypatch, seed):
    """Reference identity (the default): the HIP path places the block where the reference does."""
    from conex_amd import KktContext
    monkeypatch.delenv("CXK_REFERENCE_QUIRKS", raising=False)
    prob = random_program(seed)
    o, k = build(ol.Program, prob), build(KktContext, prob, device=0)
    o.assemble()
    k.assemble()
    assert np.abs(dense_from_slab(o, k.slab()) - dense_from_slab(o, o.slab())).max() <= 1e-12
    ok_o, yo = o.kkt_solve(prob["b"], 0.3, 0.9, 0.8)
    k.set_cost(prob["b"])
    k.kkt_solve_async(0.3, 0.9, 0.8)
    assert ok_o == 1 and k.sync()
    assert rel(k.get_y(), yo) <= 1e-10
