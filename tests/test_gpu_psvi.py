"""GPU parity of the pseudo-coreset path: the x-gradient kernel bc_project_grad_x against the
oracle's restatement of the reference formulas, and BatchPSVICoreset (bpsvi.py:6-65) against golden F16."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import models_ref as M
from oracle import coreset_ref as C

pytestmark = pytest.mark.gpu
RTOL = 1e-11        # relative to the largest entry of the tensor (fp64; contraction order differs from BLAS)


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


def fixed(th):
    return lambda n, w, p: th


def close(a, b):
    assert a.shape == b.shape
    scale = max(1., float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=0., atol=RTOL * scale)


@pytest.mark.parametrize('m,s,d', [(1, 1, 1), (7, 33, 5), (64, 100, 128), (300, 100, 40), (5, 3, 513), (257, 64, 2)])
def test_grad_x_linreg(bc, m, s, d):
    rng = np.random.RandomState(m + s + d)
    z = rng.randn(m, d + 1)
    th = rng.randn(s, d)
    for sg in (1.0, 2.5):
        prj = bc.DeviceProjector(fixed(th), s, bc.likelihoods.LinearRegression(sg))
        lls, glls = prj.project(z, grad=True)
        rl, rg = C.project_grad(lambda a, b: M.linreg_loglik(a, b, sg), lambda a, b: M.linreg_grad_x_loglik(a, b, sg), z.copy(), th)
        close(np.asarray(lls), rl)
        close(glls, rg)


@pytest.mark.parametrize('m,s,d', [(1, 1, 1), (9, 12, 5), (128, 100, 100), (33, 7, 300)])
def test_grad_x_logistic(bc, m, s, d):
    rng = np.random.RandomState(3 * m + s + d)
    z = rng.randn(m, d) * 2.
    z[0] *= 80. / np.sqrt(d)                        # some m = -z.th beyond the branch at 100
    th = rng.randn(s, d)
    prj = bc.DeviceProjector(fixed(th), s, bc.likelihoods.LogisticRegression())
    lls, glls = prj.project(z, grad=True)
    rl, rg = C.project_grad(M.logistic_loglik, M.logistic_grad_z_loglik, z.copy(), th)
    close(np.asarray(lls), rl)
    close(glls, rg)


@pytest.mark.parametrize('m,s,d', [(1, 1, 1), (9, 12, 5), (100, 100, 100), (40, 24, 4)])
def test_grad_x_gauss(bc, m, s, d):
    rng = np.random.RandomState(5 * m + s + d)
    A = rng.randn(d, d)
    Si = np.linalg.inv(A.dot(A.T) + d * np.eye(d))
    ld = -np.linalg.slogdet(Si)[1]
    x = rng.randn(m, d) * 3.
    th = rng.randn(s, d)
    prj = bc.DeviceProjector(fixed(th), s, bc.likelihoods.GaussianLocation(Si, ld))
    lls, glls = prj.project(x, grad=True)
    rl, rg = C.project_grad(lambda a, b: M.gauss_loglik(a, b, Si, ld), lambda a, b: M.gauss_grad_x_loglik(a, b, Si), x.copy(), th)
    close(np.asarray(lls), rl)
    close(glls, rg)


def test_f16_formulas_on_device(bc):
    g = load_golden('f16_bpsvi')
    z, th = g['lin_z'], g['lin_th']
    prj = bc.DeviceProjector(fixed(th), th.shape[0], bc.likelihoods.LinearRegression(2.5))
    lls, glls = prj.project(z, grad=True)
    close(np.asarray(lls), g['lin_proj_lls'])
    close(glls, g['lin_proj_glls'])
    # the un-centred reference tensors, centred here the way projector.py:31 does it
    for nm, model, pts, key in (('log', bc.likelihoods.LogisticRegression(), g['log_z'], 'log_grad'),
                                ('gau', bc.likelihoods.GaussianLocation(g['gau_Siginv'], 0.), g['gau_x'], 'gau_grad')):
        ref = g[key].copy()
        ref -= ref.mean(axis=2)[:, :, np.newaxis]
        _, glls = bc.DeviceProjector(fixed(th), th.shape[0], model).project(pts, grad=True)
        close(glls, ref)


def test_grad_needs_a_model_with_one(bc):
    th = np.zeros((4, 3))
    prj = bc.BlackBoxProjector(fixed(th), 4, lambda a, b: np.zeros((a.shape[0], 4)))
    with pytest.raises(ValueError):
        prj.project(np.zeros((2, 3)), grad=True)
    with pytest.raises(ValueError, match='no x-gradient'):
        import ctypes as Cc
        from beta_cores_amd import _native as N
        dd = bc.DeviceData(np.zeros((2, 4)))
        out = np.zeros((2, 4, 4))
        N.call('bc_project_grad_x', bc.default_context().h, dd.h, 1, th.ctypes.data_as(Cc.c_void_p), 4,
               np.array([1., .5]).ctypes.data_as(Cc.c_void_p), 2, out.ctypes.data_as(Cc.c_void_p))   # beta-likelihood: no x-gradient


def problem(bc, g, tag, projector):
    """fixture F16's two models -> (data, projector factory)"""
    if tag == 'g':
        X, E, Si, ld = g['g_X'], g['g_E'], g['g_Siginv'], float(g['g_logdet'])
        d, S = X.shape[1], E.shape[0]

        def sampler(sz, wts, pts):
            if pts.shape[0] == 0:
                wts, pts = np.zeros(1), np.zeros((1, d))
            muw, LSigw, _ = bc.gaussian_weighted_post(np.zeros(d), np.eye(d), Si, pts, wts)
            return muw + E.dot(LSigw.T)
        if projector == 'device':
            return X, lambda: bc.DeviceProjector(sampler, S, bc.likelihoods.GaussianLocation(Si, ld))
        return X, lambda: bc.BlackBoxProjector(sampler, S, lambda x, t: M.gauss_loglik(x, t, Si, ld),
                                               lambda x, t: M.gauss_grad_x_loglik(x, t, Si))
    Z, E, sg = g['l_Z'], g['l_E'], float(g['l_sigsq'])
    D, S = Z.shape[1] - 1, E.shape[0]

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, D + 1))
        muw, LSigw, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), sg, pts, wts)
        return muw + E.dot(LSigw.T)
    if projector == 'device':
        return Z, lambda: bc.DeviceProjector(sampler, S, bc.likelihoods.LinearRegression(sg))
    return Z, lambda: bc.BlackBoxProjector(sampler, S, lambda a, b: M.linreg_loglik(a, b, sg), lambda a, b: M.linreg_grad_x_loglik(a, b, sg))


@pytest.mark.parametrize('tag', ['g', 'l'])
@pytest.mark.parametrize('mode,nsub', [('full', None), ('sub', 60)])
@pytest.mark.parametrize('projector', ['device', 'blackbox'])
def test_f16_batch_psvi(bc, tag, mode, nsub, projector):
    g = load_golden('f16_bpsvi')
    data, mkprj = problem(bc, g, tag, projector)
    np.random.seed(160)
    alg = bc.BatchPSVICoreset(data, mkprj(), opt_itrs=6, n_subsample_opt=nsub, step_sched=lambda m: lambda i: 0.5 / (1. + i))
    for sz in (3, 5):
        alg.build(1, sz)
        k = '%s_%s_%d' % (tag, mode, sz)
        np.testing.assert_array_equal(alg.idcs, g[k + '_idcs'])
        np.testing.assert_allclose(alg.wts, g[k + '_wts'], rtol=1e-7)
        np.testing.assert_allclose(alg.pts, g[k + '_pts'], rtol=1e-7, atol=1e-9)
        w, p, i = alg.get()
        assert w.shape[0] == p.shape[0] == i.shape[0] <= sz
    assert np.random.rand() == float(g['%s_%s_rng_after' % (tag, mode)])
    assert alg.error() == 0.


def test_batch_psvi_large_n_matches_oracle(bc):
    """N = 200k rows (pinned in HBM), S = 100: device K1 + K2 per gradient against the NumPy restatement."""
    rng = np.random.RandomState(7)
    N, d, S = 200_000, 8, 100
    Sig = 30. * np.eye(d)
    Si = np.linalg.inv(Sig)
    ld = np.linalg.slogdet(Sig)[1]
    X = rng.multivariate_normal(np.zeros(d), Sig, N)
    E = rng.randn(S, d)

    def sampler3(sz, wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, d))
        muw, LSigw, _ = M.gauss_weighted_post(np.zeros(d), np.eye(d), Si, pts, wts)
        return muw + E.dot(LSigw.T)
    sched = lambda m: lambda i: 0.3 / (1. + i)
    np.random.seed(5)
    alg = bc.BatchPSVICoreset(X, bc.DeviceProjector(sampler3, S, bc.likelihoods.GaussianLocation(Si, ld)), opt_itrs=4, step_sched=sched)
    alg.build(1, 6)
    np.random.seed(5)
    ref = C.RefBatchPSVI(X, lambda x, t: M.gauss_loglik(x, t, Si, ld), lambda x, t: M.gauss_grad_x_loglik(x, t, Si),
                         lambda w, p: sampler3(S, w, p), 4, step_sched=sched)
    ref.build(1, 6)
    np.testing.assert_array_equal(alg.idcs, ref.idcs)
    np.testing.assert_allclose(alg.wts, ref.wts, rtol=1e-6)
    np.testing.assert_allclose(alg.pts, ref.pts, rtol=1e-6, atol=1e-8)
