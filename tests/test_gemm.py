"""Batched fp64 MFMA GEMM (conex_amd/csrc/kernels_gemm.hip.h) against numpy.

The kernel replaces Eigen's GEMM at the reference call sites dense_lmi_constraint.cc:72-103,
psd_constraint.cc:13-28/45-84 and the supernode updates block_triangular_operations.cc:184-219
for orders that do not fit the LDS-resident kernels.  Tolerance: fp64 accumulation in a different
order than numpy's BLAS -> <= 1e-13 relative to the magnitude sum |A||B|.
"""
import numpy as np
import pytest

from conex_amd import kkt

pytestmark = pytest.mark.gpu


def check(M, N, K, batch, ta, tb, alpha=1.0, beta=0.0, lower_only=False, splits=1, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (batch, M, K))
    B = rng.uniform(-1, 1, (batch, K, N))
    C0 = rng.uniform(-1, 1, (batch, M, N))
    C, _ = kkt.gemm_f64(A, B, C0, ta=ta, tb=tb, alpha=alpha, beta=beta, lower_only=lower_only,
                        splits=splits)
    ref = alpha * (A @ B) + beta * C0
    scale = np.abs(alpha) * (np.abs(A) @ np.abs(B)) + np.abs(beta * C0) + 1e-300
    if lower_only:
        mask = np.tril(np.ones((M, N), bool))
        assert np.array_equal(C[:, ~mask], C0[:, ~mask])      # untouched above the diagonal
        err = np.max(np.abs(C - ref)[:, mask] / scale[:, mask])
    else:
        err = np.max(np.abs(C - ref) / scale)
    assert err <= 1e-13, err


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("shape", [(64, 64, 16), (1, 1, 1), (17, 33, 5), (200, 200, 200),
                                   (130, 70, 259), (51, 51, 1000)])
def test_layouts_and_ragged_shapes(shape, ta, tb):
    M, N, K = shape
    check(M, N, K, 2, ta, tb, seed=M + N + K)


def test_alpha_beta_accumulate():
    check(96, 80, 40, 3, False, True, alpha=-1.0, beta=1.0)      # C -= L L^T shape


def test_lower_only_syrk_shape():
    check(150, 150, 64, 2, False, True, alpha=-1.0, beta=1.0, lower_only=True)


@pytest.mark.parametrize("splits", [2, 7, 64])
def test_split_k_ordered_reduction(splits):
    check(51, 51, 4000, 2, True, False, splits=splits)
    # bit-reproducible: same call twice gives identical bits
    rng = np.random.default_rng(5)
    A = rng.uniform(-1, 1, (1, 51, 4000))
    B = rng.uniform(-1, 1, (1, 4000, 51))
    C1, _ = kkt.gemm_f64(A, B, ta=True, splits=splits)
    C2, _ = kkt.gemm_f64(A, B, ta=True, splits=splits)
    assert np.array_equal(C1, C2)


def test_rate_at_supernode_sizes_is_reported():
    """SYRK / GEMM in isolation at n_s in {64,128,200}; just make sure the timing
    path works and the kernel is not absurdly slow (> 1 TFLOP/s at the largest size)."""
    M = N = 200
    K = 200
    batch = 256
    rng = np.random.default_rng(1)
    A = rng.uniform(-1, 1, (batch, M, K))
    B = rng.uniform(-1, 1, (batch, K, N))
    _, ms = kkt.gemm_f64(A, B, tb=True, reps=5)
    tflops = 2.0 * M * N * K * batch / (ms * 1e-3) / 1e12
    assert tflops > 1.0, tflops
