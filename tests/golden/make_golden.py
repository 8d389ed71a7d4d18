#!/usr/bin/env python3
"""Regenerates tests/golden/newton_vectors.npz.

The reference (C++ over Eigen 3.3.9, un-vendored) can be neither built nor imported in this
image, so there is no reference process to record outputs from.  The golden
vectors are therefore outputs of the ORACLE (oracle/, the plain-C restatement pinned by the
reference's own known-answer tests in tests/test_oracle_kat.py, test_oracle_hermitian.py and
test_oracle_equality.py) on seeded inputs from conex_amd/synthetic.py:

    python tests/golden/make_golden.py

For every case: the permuted Newton direction y = K^-1 (k (b bs + AQc cs) - 2 AW), the
assembled residual vectors, and the PrepareStep / eigenvalue summaries.  tests/test_golden.py
checks that (a) the oracle still reproduces them bit for bit (CPU suite) and (b) the HIP path
matches them to the stated tolerances (GPU suite) -- the latter needs neither the oracle nor
/root/reference on the GPU box.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as ol  # noqa: E402
from conex_amd import synthetic as syn  # noqa: E402

K_MU, BS, CS = 0.7, 0.9, 0.8


def cases():
    """name -> (kind, problem dict, scaling points or None, extra equality blocks)"""
    out = {}
    out["lmi_tree"] = ("lmi", syn.lmi_problem(K=9, n=6, m=6, branching=8, overlap=2, seed=100 + 9),
                       syn.scaling_points(9, 6, seed=7 + 9), [])
    out["lmi_c4_shape"] = ("lmi", syn.lmi_problem(K=30, n=20, m=20, branching=8, overlap=5, seed=130),
                           syn.scaling_points(30, 20, seed=37), [])
    out["soc_chain"] = ("soc", syn.soc_problem(K=40, dim=10, m=10, overlap=2, seed=5),
                        syn.soc_scaling_points(40, 10, seed=6), [])
    out["lp_c1"] = ("lp", syn.lp_problem(rows=20, num_vars=10), None, [])
    out["herm_complex"] = ("herm", syn.hermitian_problem(K=7, n=6, d=2, m=5, branching=2, overlap=2, seed=544),
                           syn.hermitian_scaling_points(7, 6, 2, seed=37), [])
    out["herm_quaternion"] = ("herm", syn.hermitian_problem(K=3, n=12, d=4, m=8, branching=2, overlap=3, seed=588),
                              syn.hermitian_scaling_points(3, 12, 4, seed=43), [])
    prob = syn.lmi_problem(K=12, n=6, m=6, branching=3, overlap=2, seed=3)
    rng = np.random.default_rng(3)
    eqs = [(rng.uniform(-1, 1, (2, 6)), rng.uniform(-0.1, 0.1, 2), prob["cliques"][0]),
           (rng.uniform(-1, 1, (1, 6)), rng.uniform(-0.1, 0.1, 1), prob["cliques"][7]),
           (rng.uniform(-1, 1, (1, 3)), rng.uniform(-0.1, 0.1, 1), prob["cliques"][11][:3])]
    out["lmi_with_equalities"] = ("lmi", prob, syn.scaling_points(12, 6, seed=4), eqs)
    # round 2: shapes on the kernels added then -- an order without an lmi_schur_mfma instance (18: runs
    # zero-padded at 20), an order past the register kernels (28: batched GEMM assembly), an order 22
    # with more variables than the kernel's LDS images hold (batched GEMM as well)
    out["lmi_order18"] = ("lmi", syn.lmi_problem(K=10, n=18, m=12, branching=3, overlap=4, seed=618),
                          syn.scaling_points(10, 18, seed=619), [])
    out["lmi_order28"] = ("lmi", syn.lmi_problem(K=6, n=28, m=10, branching=2, overlap=3, seed=628),
                          syn.scaling_points(6, 28, seed=629), [])
    out["lmi_order22_m20"] = ("lmi", syn.lmi_problem(K=5, n=22, m=20, branching=2, overlap=5, seed=622),
                              syn.scaling_points(5, 22, seed=623), [])
    return out


def build(cls, kind, prob, eqs, **kw):
    if kind == "lp":
        p = cls(prob["A"].shape[1], **kw)
        p.add_linear(prob["A"], prob["c"])
    else:
        p = cls(prob["num_vars"], **kw)
        add = {"lmi": p.add_lmi, "soc": p.add_soc, "herm": p.add_hermitian}[kind]
        for c, cl in enumerate(prob["cliques"]):
            add(prob["A"][c], prob["C" if kind != "soc" else "c"][c], cl)
    for Ae, be, v in eqs:
        p.add_equality(Ae, be, v)
    p.initialize()
    return p


def run(p, W, b):
    """One KKT solve + step summaries through the common interface of oracle and KktContext."""
    if W is not None:
        for i in range(len(W)):
            p.set_W(i, W[i])
    p.assemble()
    AW, AQc, sc = p.residuals()
    assert p.factor() == 1
    N = p.N
    bb = np.zeros(N)
    bb[:len(b)] = b
    rhs = K_MU * (bb * BS + AQc * CS) - 2 * AW
    y = p.solve_inplace(rhs)
    eig = p.weighted_slack_eigenvalues(y, K_MU * CS)
    info = p.prepare_step(y, K_MU * CS, 1.0)
    return dict(y=y, AW=AW, AQc=AQc, sc=sc, eig=np.asarray(eig), info=np.asarray(info))


def main():
    out = {}
    for name, (kind, prob, W, eqs) in cases().items():
        o = build(ol.Program, kind, prob, eqs)
        res = run(o, W, prob["b"])
        for k, v in res.items():
            out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "newton_vectors.npz"), **out)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "newton_vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
