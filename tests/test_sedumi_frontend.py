"""SeDuMi-format front end (include/conex_sedumi.h, conex_amd/csrc/sedumi.cc): the reference's MATLAB
pipeline interfaces/matlab/conex.m + util/*.m as host C++ in front of the CONEX_* ABI.

CPU: the library's preprocessing against the numpy restatement of the .m files (oracle/cxo_sedumi.py),
itself pinned by the reference's literal test interfaces/matlab/test/test_extract_constraints.m.
GPU: whole solves -- a block-diagonal SDP hidden in large, symmetrically permuted LMIs is split into
its blocks and solved through the clique path; same optimum as the unsplit solve and as the CPU
oracle's interior-point loop on the same blocks."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cxo_sedumi as oracle  # noqa: E402
from conex_amd import sedumi  # noqa: E402


def literal_problem():
    """test_extract_constraints.m:2-13"""
    A = np.array([[1, 2, 2, 1, 0, 0, 0, 0],
                  [0, 0, 0, 0, 2, 1, 1, 2],
                  [1, 3, 3, 1, 2, -3, -3, 2]], dtype=float)
    A[A < .5] = 0
    return A, np.random.default_rng(1).normal(size=8), [2, 2]


def matlab_check(blocks, A, c, Ks):
    """DoTest of test_extract_constraints.m:15-38, restated: per block the variables are the rows of A
    with a nonzero in the block's columns and the matrices their rows reshaped."""
    s = 0
    for i, n in enumerate(Ks):
        Ai = A[:, s:s + n * n]
        var = np.flatnonzero(np.any(Ai != 0, axis=1))
        assert np.array_equal(blocks[i]["variables"], var)
        for k, v in enumerate(var):
            assert np.array_equal(blocks[i]["matrices"][:, :, k], Ai[v].reshape(n, n).T)
        assert np.array_equal(blocks[i]["affine"], c[s:s + n * n].reshape(n, n).T)
        s += n * n


def test_extract_constraint_matrices_literal():
    A, c, Ks = literal_problem()
    csym = oracle.symmetrize(c[None, :], Ks)[0]
    matlab_check(oracle.extract_constraint_matrices(A, csym, Ks), A, csym, Ks)      # the oracle is pinned ...
    got = sedumi.preprocess(A, np.ones(3), c, {"s": Ks}, blkdiag=0)               # ... and the library equals it
    matlab_check(got["blocks"], oracle.symmetrize(A, Ks), csym, Ks)


def hidden_blocks(seed, big, parts, m, density=0.6):
    """PSD blocks of orders `big`, each hiding diagonal sub-blocks of orders parts[i] under a random
    symmetric permutation; m constraint rows, each touching a random subset of the sub-blocks."""
    rng = np.random.default_rng(seed)
    N = sum(n * n for n in big)
    A = np.zeros((m, N))
    c = np.zeros(N)
    off = 0
    subs = []
    for n, ps in zip(big, parts):
        perm = rng.permutation(n)
        at = 0
        for p in ps:
            idx = perm[at:at + p]
            subs.append((off, n, idx))
            at += p
        off += n * n
    for off, n, idx in subs:
        Cm = np.zeros((n, n))
        Cm[np.ix_(idx, idx)] = np.eye(len(idx)) * 2 + 0.1
        c[off:off + n * n] += Cm.T.ravel()
    for r in range(m):
        chosen = [k for k in range(len(subs)) if rng.uniform() < density] or [int(rng.integers(len(subs)))]
        for k in chosen:
            off, n, idx = subs[k]
            R = rng.uniform(-1, 1, (len(idx), len(idx)))
            Mat = np.zeros((n, n))
            Mat[np.ix_(idx, idx)] = R + R.T
            A[r, off:off + n * n] += Mat.T.ravel()
    x0 = np.zeros(N)
    for off, n, idx in subs:
        X = np.zeros((n, n))
        X[np.ix_(idx, idx)] = np.eye(len(idx))
        x0[off:off + n * n] += X.T.ravel()
    return A, A @ x0, c, big, subs


def same_preprocessing(got, want):
    assert np.array_equal(got["kept_rows"], want["kept_rows"])
    assert np.array_equal(got["kept_cols"], want["kept_cols"])
    assert np.array_equal(got["b"], want["b"])
    assert len(got["blocks"]) == len(want["blocks"])
    for g, w in zip(got["blocks"], want["blocks"]):
        assert g["order"] == w["order"]
        assert np.array_equal(g["variables"], w["variables"])
        assert np.array_equal(g["matrices"], w["matrices"])
        assert np.array_equal(g["affine"], w["affine"])


@pytest.mark.parametrize("seed,big,parts,m", [(1, [5, 4], [[3, 2], [4]], 6), (2, [7], [[2, 2, 3]], 5),
                                              (3, [6, 6, 3], [[1, 5], [2, 2, 2], [3]], 9),
                                              (4, [8, 2], [[3, 3], [1]], 4), (5, [9], [[4, 5]], 12)])
def test_block_splitting_equals_the_matlab_restatement(seed, big, parts, m):
    A, b, c, Ks, subs = hidden_blocks(seed, big, parts, m)
    A = np.vstack([A, np.zeros((2, A.shape[1]))])                  # all-zero rows: CleanLinear drops them
    b = np.r_[b, 0, 0]
    want = oracle.preprocess(A, b, c, Ks, blkdiag=True)
    got = sedumi.preprocess(A, b, c, {"s": Ks}, blkdiag=1)
    same_preprocessing(got, want)
    # what it is for: every hidden sub-block comes out as a block of its own
    assert sorted(b2["order"] for b2 in got["blocks"]) == sorted(p for ps in parts for p in ps)
    assert not np.isin(got["kept_rows"], [len(b) - 1, len(b) - 2]).any()
    same_preprocessing(sedumi.preprocess(A, b, c, {"s": Ks}, blkdiag=0), oracle.preprocess(A, b, c, Ks, blkdiag=False))
    # unsymmetric data (Symmetrize averages it) that links two sub-blocks: they merge, on both sides
    A2 = A.copy()
    A2[0, 1] += 0.37
    same_preprocessing(sedumi.preprocess(A2, b, c, {"s": Ks}, blkdiag=1), oracle.preprocess(A2, b, c, Ks, blkdiag=True))


def test_single_psd_block_with_blkdiag_is_split_not_handed_over_whole():
    """The documented divergence from conex.m (include/conex_sedumi.h): with pars.blkdiag = 1 and ONE
    PSD block, conex.m:45-49 passes the unreduced matrices with the reduced order (a reshape error
    whenever the preprocessing did anything); this front end hands the solver the preprocessed
    blocks.  With conex.m's default (blkdiag off for a single block) the block stays whole."""
    A, b, c, Ks, _ = hidden_blocks(11, [8], [[3, 5]], 6)
    assert len(Ks) == 1
    split = sedumi.preprocess(A, b, c, {"s": Ks}, blkdiag=1)
    assert sorted(blk["order"] for blk in split["blocks"]) == [3, 5]
    for mode in (-1, 0):
        whole = sedumi.preprocess(A, b, c, {"s": Ks}, blkdiag=mode)
        assert [blk["order"] for blk in whole["blocks"]] == [8]


def test_triplets_add_up_and_bad_input_is_refused():
    import scipy.sparse as sp
    A, b, c, Ks, _ = hidden_blocks(7, [5], [[2, 3]], 4)
    half = sp.coo_matrix(A / 2)
    dup = sp.coo_matrix((np.r_[half.data, half.data], (np.r_[half.row, half.row], np.r_[half.col, half.col])), shape=A.shape)
    same_preprocessing(sedumi.preprocess(dup, b, c, {"s": Ks}), oracle.preprocess(A, b, c, Ks))
    with pytest.raises(ValueError):
        sedumi.preprocess(A, b, c, {"s": [4]})                    # N != sum K.s^2
    with pytest.raises(ValueError):
        sedumi.preprocess(A, b[:-1], c, {"s": Ks})


def test_library_exports_the_front_end():
    import re
    from conex_amd import load_library
    text = open(os.path.join(ROOT, "include", "conex_sedumi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(CONEX_[A-Za-z_0-9]+)\s*\(", text)))
    assert len(names) == 10
    L = load_library()
    for n in names:
        assert hasattr(L, n), n


def mats(x, Ks):
    out, off = [], 0
    for n in Ks:
        out.append(x[off:off + n * n].reshape(n, n).T)
        off += n * n
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("seed,big,parts,m", [(11, [12, 9], [[4, 5, 3], [6, 3]], 10), (12, [10, 10, 10], [[5, 5], [3, 7], [10]], 14)])
def test_split_solve_equals_unsplit_solve_and_the_oracle(seed, big, parts, m):
    import oracle_lib as ol
    A, b, c, Ks, subs = hidden_blocks(seed, big, parts, m, density=0.5)
    x, y, info = sedumi.conex(A, b, c, {"s": Ks}, errors=True)                    # default: block-diagonalised
    assert info.solved == 1 and info.num_blocks == len(subs)
    xu, yu, infou = sedumi.conex(A, b, c, {"s": Ks}, blkdiag=0, errors=True)      # the big LMIs as they are
    assert infou.solved == 1 and infou.num_blocks == len(big)
    assert abs(b @ y - b @ yu) <= 1e-6 * (1 + abs(b @ yu))
    # SeDuMi's optimality conditions: A x = b, slack and x PSD, no gap
    csym = oracle.symmetrize(c[None, :], Ks)[0]
    assert np.linalg.norm(A @ x - b) <= 1e-6 * (1 + np.linalg.norm(b))
    for S, X in zip(mats(csym - oracle.symmetrize(A, Ks).T @ y, Ks), mats(x, Ks)):
        assert np.linalg.eigvalsh((S + S.T) / 2).min() >= -1e-7
        assert np.linalg.eigvalsh((X + X.T) / 2).min() >= -1e-7
    assert info.errors[0] <= 1e-5 * (1 + abs(b @ y))
    # the same blocks through the CPU oracle's interior-point loop
    pre = oracle.preprocess(A, b, c, Ks, blkdiag=True)
    o = ol.Program(len(pre["b"]))
    for blk in pre["blocks"]:
        o.add_lmi(np.transpose(blk["matrices"], (2, 0, 1)), blk["affine"], blk["variables"])
    cfg = ol.default_config()
    cfg.prepare_dual_variables = 1
    cfg.inv_sqrt_mu_max = 1000
    cfg.infeasibility_threshold = 1e3
    cfg.max_iterations = 25
    cfg.divergence_upper_bound = 1
    cfg.final_centering_steps = 5
    oko, yo = o.solve(pre["b"], cfg)
    assert oko == 1
    assert np.linalg.norm(y[pre["kept_rows"]] - yo) <= 1e-6 * (1 + np.linalg.norm(yo))


@pytest.mark.gpu
def test_single_block_takes_the_dense_path_of_conex_m():
    """conex.m:46-50: one PSD block and no pars: one dense LMI over all rows (run_solver_comparison.m's
    shape: A_i = sym(randn), b = A vec(I), c = vec(I))."""
    rng = np.random.default_rng(5)
    n, m = 12, 8
    A = np.zeros((m, n * n))
    for i in range(m):
        R = rng.normal(size=(n, n))
        A[i] = (R + R.T).ravel()
    b = A @ np.eye(n).ravel()
    c = np.eye(n).ravel()
    x, y, info = sedumi.conex(A, b, c, {"s": [n]}, errors=True)
    assert info.solved == 1 and info.num_blocks == 1 and info.num_rows_kept == m
    assert np.linalg.norm(A @ x - b) <= 1e-6 * (1 + np.linalg.norm(b))
    S = (c - A.T @ y).reshape(n, n)
    assert np.linalg.eigvalsh(S).min() >= -1e-7 and info.errors[0] <= 1e-5 * (1 + abs(b @ y))
