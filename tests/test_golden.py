"""Golden vectors (tests/golden/newton_vectors.npz, made by tests/golden/make_golden.py).

CPU suite: the oracle reproduces them bit for bit (pins the restatement against drift).
GPU suite: the HIP path matches them -- Newton direction <= 1e-10 relative norm (BASELINE.json
north_star), residual vectors <= 1e-12, traces / norms <= 1e-9 -- without touching the oracle.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "newton_vectors.npz"))
CASES = sorted(mg.cases().keys())


def rel(a, b):
    n = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (n if n > 0 else 1.0)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden_vectors_bit_for_bit(name):
    import oracle_lib as ol
    kind, prob, W, eqs = mg.cases()[name]
    res = mg.run(mg.build(ol.Program, kind, prob, eqs), W, prob["b"])
    for k, v in res.items():
        assert np.array_equal(v, GOLD[f"{name}/{k}"]), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_matches_golden_vectors(name):
    from conex_amd import KktContext
    kind, prob, W, eqs = mg.cases()[name]
    res = mg.run(mg.build(KktContext, kind, prob, eqs, device=0), W, prob["b"])
    assert rel(res["y"], GOLD[f"{name}/y"]) <= 1e-10
    assert rel(res["AW"], GOLD[f"{name}/AW"]) <= 1e-12
    assert rel(res["AQc"], GOLD[f"{name}/AQc"]) <= 1e-12
    assert rel(res["sc"], GOLD[f"{name}/sc"]) <= 1e-12
    assert rel(res["eig"], GOLD[f"{name}/eig"]) <= 1e-9
    assert rel(res["info"], GOLD[f"{name}/info"]) <= 1e-9
