"""GPU parity of K4 (weighted Gram, fp64 MFMA) and the posterior update built on it."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import models_ref as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


@pytest.mark.parametrize('n,d', [(1, 1), (5, 3), (17, 16), (300, 8), (1000, 63), (1000, 64), (999, 65), (2000, 95),
                                 (2000, 96), (3000, 127), (3000, 128), (1500, 200), (700, 512),
                                 (40, 2049), (60, 2100), (30, 4200)])     # D > 2048: X^T (w*y) needs several chunk trips (k_gram_reduce2)
@pytest.mark.parametrize('weighted', [True, False])
def test_gram_matches_numpy(bc, n, d, weighted):
    rng = np.random.RandomState(n + d)
    Z = rng.randn(n, d + 1)
    w = rng.rand(n) * 2. if weighted else None
    G, v = bc.weighted_gram(Z, w)
    Gr, vr = M.linreg_xtwx(Z, w if weighted else np.ones(n))
    scale = np.abs(Gr).max()
    assert np.abs(G - Gr).max() <= 1e-12 * scale
    assert np.abs(v - vr).max() <= 1e-12 * max(1., np.abs(vr).max())
    assert np.array_equal(G, G.T)                                     # mirrored from the upper triangle: exactly symmetric


def test_f6_weighted_post_goldens(bc):
    g = load_golden('f6_weighted_post')
    for D in (8, 64):
        mu, L, Linv = bc.weighted_post(g['D%d_th0' % D], g['D%d_Sig0inv' % D], 1.7, g['D%d_Z' % D], g['D%d_w' % D])
        np.testing.assert_allclose(mu, g['D%d_mu' % D], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(L, g['D%d_L' % D], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(Linv, g['D%d_Linv' % D], rtol=1e-9, atol=1e-12)
        # the reference's mean is NOT the true posterior mean on this anisotropic design (quirk kept)
        mu_true = bc.weighted_post_corrected(g['D%d_th0' % D], g['D%d_Sig0inv' % D], 1.7, g['D%d_Z' % D], g['D%d_w' % D])[0]
        assert np.abs(mu_true - mu).max() > 1e-3
    mu, L, Linv = bc.gaussian_weighted_post(np.zeros(8), np.eye(8), g['g_Siginv'], g['g_X'], g['g_w'])
    np.testing.assert_allclose(mu, g['g_mu'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(L, g['g_L'], rtol=1e-12)


def test_gram_large_rows_splitk(bc):
    rng = np.random.RandomState(0)
    n, d = 200_000, 64
    Z = rng.randn(n, d + 1)
    w = rng.rand(n)
    G, v = bc.weighted_gram(Z, w)
    Gr, vr = M.linreg_xtwx(Z, w)
    assert np.abs(G - Gr).max() <= 1e-12 * np.abs(Gr).max()
    assert np.abs(v - vr).max() <= 1e-11 * np.abs(vr).max()
    G2, v2 = bc.weighted_gram(Z, w)
    assert np.array_equal(G, G2) and np.array_equal(v, v2)            # fixed split order: run-to-run deterministic


@pytest.mark.parametrize('n', [4096, 10_007, 100_001, 262_145])
@pytest.mark.parametrize('d', [127, 128])
def test_gram_lds_dma_kernel(bc, n, d, monkeypatch):
    """One unweighted diagonal tile whose rows hold >= 128 doubles (the drivers' full-data posterior at D = 128,
    model_linreg.py:25-34 with w = 1) runs k_gram_dma: slabs go from global memory straight into LDS, 32 rows at a time,
    double-buffered.  Row counts that end inside a slab (zero-filled rows), one that is a multiple of 32, one row more than
    a multiple of the split size.  Run-to-run deterministic; against the register-staged kernel (BC_GRAM_DMA=0) the results
    agree to rounding (the row splits, and with them the association of the partial sums, follow the slab size); both
    match NumPy."""
    rng = np.random.RandomState(n + d)
    Z = rng.randn(n, d + 1)
    Z[-1, -1] = 7.5                                       # the last row's y: the end of the array is read exactly, not past
    monkeypatch.setenv('BC_GRAM_DMA', '1')
    G, v = bc.weighted_gram(Z, None)
    G2, v2 = bc.weighted_gram(Z, None)
    assert np.array_equal(G, G2) and np.array_equal(v, v2)
    monkeypatch.setenv('BC_GRAM_DMA', '0')
    G0, v0 = bc.weighted_gram(Z, None)
    assert np.abs(G - G0).max() <= 1e-13 * np.abs(G0).max()
    np.testing.assert_allclose(v, v0, rtol=1e-12, atol=1e-12 * np.abs(v0).max())
    Gr, vr = M.linreg_xtwx(Z, np.ones(n))
    assert np.abs(G - Gr).max() <= 1e-12 * np.abs(Gr).max()
    assert np.abs(v - vr).max() <= 1e-11 * max(1., np.abs(vr).max())
    assert np.array_equal(G, G.T)
