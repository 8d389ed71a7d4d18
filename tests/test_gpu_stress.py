"""Repetition tests for the one unexplained failure recorded in round 1 (one of nine
runs of tests/test_gpu_solver.py failed around test_sparse_sdp_equals_dense_formulation, output
not kept).  Fifty in-process repetitions of that test and its neighbours, with the lean kernels
and the host spin-wait both on and off, and bit-equality of repeated solves: a race or a read of
unwritten memory would show up as a changed bit.  (The debugging aid that goes with it:
CXK_DEBUG_FILL_NAN=1 makes every scratch buffer of a context start as NaN.)
"""
import os

import numpy as np
import pytest

import test_gpu_solver as tgs
from conex_amd import KktContext
from conex_amd import synthetic as syn

pytestmark = pytest.mark.gpu

REPS = 50


@pytest.mark.parametrize("env", [{}, {"CXK_NO_LEAN": "1"}, {"CXK_NO_SPIN": "1"}])
def test_sparse_sdp_and_neighbours_50_times(env, monkeypatch):
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    for _ in range(REPS):
        tgs.test_lp_dense_optimality_and_dual_recovery()
        tgs.test_sparse_sdp_equals_dense_formulation()
        tgs.test_socp_matches_lmi_arrow_formulation()


def test_repeated_kkt_solves_are_bit_identical_across_fresh_contexts():
    """Ten fresh contexts of the same mixed program, five solves each: every direction equals the
    first one bit for bit (no atomics, fixed summation orders, no reads of unwritten memory)."""
    prob = syn.mixed_problem(K=230, seed=31)
    W = syn.mixed_scaling_points(prob, seed=32)
    ref = None
    for _ in range(10):
        k = syn.build(KktContext, prob, "mixed", device=0)
        for i in range(k.K):
            k.set_W(i, W[i])
        for _ in range(5):
            ok, y = k.kkt_solve(prob["b"], 0.5, 0.9, 0.8)
            assert ok == 1
            if ref is None:
                ref = y.copy()
            assert np.array_equal(y, ref)


def test_hermitian_newton_step_50_times():
    """Round 2: tests/test_gpu_parity.py::test_hermitian_newton_step[7-6-5-2-2-2] failed once (output not
    kept) in the run that followed a rebuild; 600 in-process repetitions, 16 repetitions of the whole
    file and the NaN-filled runs did not reproduce it.  Kept under repetition here."""
    import test_gpu_parity as tgp
    for _ in range(REPS):
        for d in (1, 2, 4):
            tgp.test_hermitian_newton_step(d, 7, 6, 5, 2, 2)
        tgp.test_hermitian_newton_step(2, 3, 12, 8, 2, 3)
