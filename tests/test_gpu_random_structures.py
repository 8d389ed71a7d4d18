"""Randomised structures: clique trees with ragged clique sizes and mixed cone kinds, so that every
sweep variant (register kernels <8,8> <16,8> <24,0> <24,8> <32,16>, LDS wavefront path,
workgroup-per-supernode path, merged backward ranges, single-workgroup top) and every assembly
kernel meets the oracle on the same seeded input.  Newton direction <= 1e-10 (north_star)."""
import numpy as np
import pytest

import oracle_lib as ol
from conex_amd import KktContext
from conex_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def random_program(seed):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(3, 60))
    branching = int(rng.integers(1, 6))
    cliques = []
    own_start, own_len = [], []
    nxt = 0
    kinds, data = [], []
    for c in range(K):
        size = int(rng.choice([3, 5, 8, 10, 14, 18, 22, 27, 35]))
        if c == 0:
            shared = []
        else:
            p = (c - 1) // branching
            ov = int(min(rng.integers(1, 7), own_len[p], size - 1))
            off = int(rng.integers(0, own_len[p] - ov + 1))
            shared = list(range(own_start[p] + off, own_start[p] + off + ov))
        fresh = size - len(shared)
        own_start.append(nxt)
        own_len.append(fresh)
        cl = shared + list(range(nxt, nxt + fresh))
        nxt += fresh
        order = rng.permutation(len(cl))            # variables out of order inside the clique
        cl = [cl[i] for i in order]
        cliques.append(cl)
        m = len(cl)
        kind = rng.choice(["lmi", "soc", "lin", "herm"], p=[0.4, 0.25, 0.25, 0.1])
        if kind == "lmi":
            n = int(rng.choice([2, 3, 5, 8]))
            while n * (n + 1) // 2 < m:            # keep the Schur block full rank
                n += 1
            A = rng.uniform(-1, 1, (m, n, n))
            A = 0.5 * (A + np.transpose(A, (0, 2, 1)))
            data.append((A, np.eye(n)))
        elif kind == "herm":
            n = 3
            while n * n < m:                       # complex Hermitian: n^2 real dimensions
                n += 1
            A = np.stack([syn.random_hermitian(rng, 2, n) for _ in range(m)])
            C = np.zeros((2, n, n))
            C[0] = np.eye(n)
            data.append((A, C))
        elif kind == "soc":
            dim = max(m, 3) + int(rng.integers(0, 3))
            A = rng.uniform(-1, 1, (dim + 1, m))
            cc = np.zeros(dim + 1)
            cc[0] = 1.0
            data.append((A, cc))
        else:
            rows = m + int(rng.integers(1, 6))
            data.append((rng.uniform(-1, 1, (rows, m)), np.abs(rng.uniform(0.5, 1.5, rows))))
        kinds.append(kind)
    b = rng.uniform(-1, 1, nxt)
    return dict(kinds=kinds, data=data, cliques=cliques, num_vars=nxt, b=b)


def build(cls, prob, **kw):
    p = cls(prob["num_vars"], **kw)
    for kind, (A, C), cl in zip(prob["kinds"], prob["data"], prob["cliques"]):
        add = {"lmi": p.add_lmi, "herm": p.add_hermitian, "soc": p.add_soc, "lin": p.add_linear}[kind]
        assert add(A, C, cl) >= 0
    p.initialize()
    return p


def rel(a, b):
    n = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (n if n > 0 else 1.0)


@pytest.mark.parametrize("seed", range(24))
def test_random_structure_newton_direction(seed):
    prob = random_program(1000 + seed)
    o, k = build(ol.Program, prob), build(KktContext, prob, device=0)
    assert np.array_equal(o.order(), k.order())
    rng = np.random.default_rng(seed)
    for it in range(2):
        ok_o, yo = o.kkt_solve(prob["b"], 0.3 + 0.3 * it, 0.9, 0.8)
        k.set_cost(prob["b"])
        k.kkt_solve_async(0.3 + 0.3 * it, 0.9, 0.8)
        ok_k = k.sync()
        assert ok_o == 1 and ok_k
        yk = k.get_y()
        assert rel(yk, yo) <= 1e-10
        # one damped update keeps both sides on the same interior point for the second pass
        c_weight = (0.3 + 0.3 * it) * 0.8
        io = o.prepare_step(yo, c_weight, 1.0)
        ik = k.prepare_step(yo, c_weight, 1.0)
        assert abs(ik[0] - io[0]) <= 1e-9 * max(1.0, abs(io[0]))
        step = min(1.0, 1.0 / (io[1] * io[1] + 1e-300)) * 0.5
        o.take_step(step)
        k.take_step(step)
