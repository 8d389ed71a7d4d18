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


@pytest.mark.parametrize("shape", [(29, 29, 784, True), (21, 17, 100, False), (32, 32, 1024, True), (5, 31, 36, False),
                                   (16, 16, 400, True)])
@pytest.mark.parametrize("splits", [1, 3])
def test_gram_products_share_one_quadrant(shape, splits):
    """X^T Y of at most 32 x 32 (the LMI assembly's Gram matrices, kernels_lmi_large.hip.h): the four
    waves of a workgroup deal the k sub-steps of the one occupied tile quadrant among themselves."""
    M, N, K, lower = shape
    for batch in (1, 13):   # 13 workgroups: a grid that is not a multiple of the 8 XCDs
        check(M, N, K, batch, True, False, lower_only=lower, splits=splits, seed=K + batch)


@pytest.mark.parametrize("n", [64, 65, 200, 257])
def test_lower_only_launches_the_lower_tiles_only(n):
    """A square lower-only call: compact grid of T (T + 1) / 2 tiles, relabelled over the XCDs."""
    for batch in (1, 3):
        check(n, n, 48, batch, False, True, alpha=-1.0, beta=1.0, lower_only=True, seed=n + batch)
        check(n, n, 40, batch, True, False, lower_only=True, seed=n)


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


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("shape", [(128, 128, 40, 300), (120, 250, 50, 130), (256, 192, 64, 100), (128, 64, 24, 1100)])
def test_big_tile_kernel_on_the_shapes_it_is_chosen_for(shape, ta, tb):
    """gemm_f64_dma128 (128 x 128 / 128 x 64 tiles, gemm_mfma.hip ChooseTile): products without an old C,
    sides that fill their last tile, enough workgroups for the chip."""
    M, N, K, batch = shape
    check(M, N, K, batch, ta, tb, seed=M + N + K)


def test_big_tile_kernel_split_k_and_lower_only():
    check(128, 128, 400, 100, True, False, splits=3)
    check(128, 128, 48, 300, False, True, lower_only=True)
    check(256, 256, 48, 100, False, True, lower_only=True)


def test_big_tile_kernel_forced_onto_ragged_shapes():
    """CXK_GEMM_TILE=128 / 12864 (read once per process): every layout on ragged shapes through the big-tile
    kernel -- tiles with 1 .. 8 sub-tiles a side, dealt to the four wavefronts of a workgroup."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_gemm as t\n"
        "for ta in (False, True):\n"
        "    for tb in (False, True):\n"
        "        for (M, N, K) in ((200, 200, 200), (130, 70, 258), (17, 33, 6), (65, 129, 40), (300, 90, 32), (128, 16, 20)):\n"
        "            t.check(M, N, K, 3, ta, tb, seed=M + N)\n"
        "t.check(200, 200, 64, 2, False, True, lower_only=True)\n"
        "t.check(150, 150, 400, 2, True, False, splits=5)\n"
        "print('ok')\n" % (root, os.path.join(root, "tests")))
    for tile in ("128", "12864"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CXK_GEMM_TILE=tile), capture_output=True, text=True)
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]


RATE_FLOORS = [
    # (M, N, K, batch, ta, tb, alpha, beta, lower, floor in useful TFLOP/s)
    # measured on MI355X (profiles/r02/gemm_rates.jsonl): 39.1, 36.2, 45.6, 24.9, 23.5; floors leave ~20 %
    # for box-to-box variance (the chip holds a lower clock under fp64 MFMA load on random data:
    # 61-68 TFLOP/s sustained by a pure MFMA loop against 77.5 on constant operands)
    (128, 128, 200, 1024, True, False, 1.0, 0.0, False, 31.0),   # separator update U = off^T off, s = 128, n_s = 200
    (128, 128, 128, 1024, False, False, 1.0, 0.0, False, 29.0),  # 128^3
    (1024, 1024, 1024, 4, False, False, 1.0, 0.0, False, 36.0),  # 1024^3
    (128, 128, 128, 1024, False, True, -1.0, 1.0, True, 19.5),   # SYRK n_s = 128, s = 128 (HBM-bound: 5.4 flop/byte)
    (200, 200, 128, 512, False, True, -1.0, 1.0, True, 18.0),    # SYRK n_s = 200, s = 128 (3.1 tiles of 64)
]


@pytest.mark.parametrize("M,N,K,batch,ta,tb,alpha,beta,lower,floor", RATE_FLOORS)
def test_rate_at_supernode_sizes(M, N, K, batch, ta, tb, alpha, beta, lower, floor):
    """SURVEY 8(d) / north_star: the SYRK / GEMM trailing updates in isolation at supernode sizes.
    Asserts the rate (useful flops: the entries kept), not just that the timing path works."""
    rng = np.random.default_rng(1)
    A = rng.uniform(-1, 1, (batch, M, K))
    B = rng.uniform(-1, 1, (batch, K, N))
    C0 = rng.uniform(-1, 1, (batch, M, N)) if beta != 0 else None
    _, ms = kkt.gemm_f64(A, B, C0, ta=ta, tb=tb, alpha=alpha, beta=beta, lower_only=lower, reps=8)
    useful = (M * (N + 1) / 2 if lower else M * N) * K * 2.0 * batch
    tflops = useful / (ms * 1e-3) / 1e12
    assert tflops >= floor, tflops
