"""Bit-exact parity of the product's host symbolic analysis (conex_amd/csrc/symbolic.cc,
inverted-index formulation) with the literal O(K^2) restatement in oracle/ -- SURVEY 8a rows
A2-A7.  Runs without a GPU: the context is created host-only (device = -1)."""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import KktContext
from conex_amd.synthetic import tree_cliques, chain_cliques


def _both(cliques, num_vars):
    o = ol.Program(num_vars)
    k = KktContext(num_vars, device=-1)
    for c in cliques:
        m = len(c)
        assert o.add_static(np.eye(m), c) == k.add_static(np.eye(m), c)
    o.initialize()
    k.initialize()
    return o, k


def _assert_same(o, k):
    assert o.K == k.K and o.N == k.N
    assert np.array_equal(o.order(), k.order())
    po, qo = o.permutation()
    pk, qk = k.permutation()
    assert np.array_equal(po, pk) and np.array_equal(qo, qk)
    assert np.array_equal(o.supernode_sizes(), k.supernode_sizes())
    for e in range(o.K):
        for which in range(5):
            assert np.array_equal(o.get_list(which, e), k.get_list(which, e)), (which, e)
        assert np.array_equal(o.ss_index(e), k.ss_index(e)), e
    assert o.slab_size() == k.slab_size()
    do, oo = o.block_offsets()
    dk, ok_ = k.block_offsets()
    assert np.array_equal(do, dk) and np.array_equal(oo, ok_)


LITERAL = [
    [[1, 2, 3, 5], [3, 4, 5], [4, 5, 6, 7], [8, 9], [1, 11]],
    [[0, 2, 3, 5], [3, 4, 5], [4, 5, 6, 7], [0, 11]],
    [[0, 1]],
    [[0, 1], [1, 2]],
    [[0, 1], [1, 2], [0, 3], [2, 3]],            # needs fill-in (clique_ordering_test.cc:110)
    [[0, 1], [0, 1, 2], [0, 1, 2, 3, 4]],        # non-maximal cliques (:126)
    [[0, 1, 2, 4, 7], [3, 4], [5, 6, 7]],
    [[0, 1, 5], [1, 2, 5], [3, 4, 5]],
    [[1, 0, 3], [1, 0, 2]],                      # out-of-order variables (assembly_test.cc:196)
]


@pytest.mark.parametrize("cliques", LITERAL)
def test_literal_clique_sets(cliques):
    nv = max(max(c) for c in cliques) + 1
    # the reference requires variables 0..max to be covered; pad with singleton use if not
    used = set(v for c in cliques for v in c)
    cl = [list(c) for c in cliques]
    for v in range(nv):
        if v not in used:
            cl.append([v])
    o, k = _both(cl, nv)
    _assert_same(o, k)


def _random_cliques(rng, K, nv, lo, hi):
    cliques = []
    for _ in range(K):
        m = int(rng.integers(lo, hi + 1))
        cliques.append(list(rng.choice(nv, size=min(m, nv), replace=False)))
    used = set(v for c in cliques for v in c)
    for v in range(nv):
        if v not in used:
            cliques[int(rng.integers(0, K))].append(v)
    return cliques


@pytest.mark.parametrize("seed", range(25))
def test_random_clique_sets(seed):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 30))
    nv = int(rng.integers(3, 40))
    cliques = _random_cliques(rng, K, nv, 1, 6)
    o, k = _both(cliques, nv)
    _assert_same(o, k)


@pytest.mark.parametrize("K,b", [(9, 8), (73, 8), (200, 3), (64, 2)])
def test_tree_structures(K, b):
    cliques, nv = tree_cliques(K, branching=b, clique_size=20, overlap=5)
    o, k = _both(cliques, nv)
    _assert_same(o, k)
    assert k.N == 20 + 15 * (K - 1)


def test_chain_structure():
    cliques, nv = chain_cliques(120, 10, 2)
    o, k = _both(cliques, nv)
    _assert_same(o, k)


def test_headline_structure_c4():
    """BASELINE config 4: 1000 cliques of 20, 8-ary tree, overlap 5 => N = 15005."""
    cliques, nv = tree_cliques(1000, 8, 20, 5)
    o, k = _both(cliques, nv)
    _assert_same(o, k)
    assert k.N == 15005


def test_rejects_duplicate_and_out_of_range_variables():
    k = KktContext(4, device=-1)
    o = ol.Program(4)
    assert k.add_static(np.eye(2), [1, 1]) == -1 == o.add_static(np.eye(2), [1, 1])
    assert k.add_static(np.eye(2), [1, 7]) == -1 == o.add_static(np.eye(2), [1, 7])
    assert k.add_static(np.eye(2), [1, 2]) == 0
