"""The store-free K1 (bc_project_colsum) and the fused gradient of the greedy-VI weight optimisation (bc_vi_gradient):
the gradient loop of BetaCoreset / SparseVI (bcores.py:141-146, sparsevi.py:129-134) needs only `vecs.sum(axis=0)` of the
N x S projection, so K1 keeps its column partials and writes no Phi.  The bar: the SAME bits as the column sums of the
materialised projection, and gradients / coresets equal to the general path's."""
import numpy as np
import pytest

from oracle import models_ref as M
from oracle import coreset_ref as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


def fixed(th):
    return lambda n, w, p: th


def _models(bc, rng, d):
    Sig = np.diag(rng.uniform(0.5, 2.0, d))
    return [
        ('linreg', bc.likelihoods.LinearRegression(1.3), d + 1, (None, 0.3)),
        ('logistic', bc.likelihoods.LogisticRegression(), d, (None, 0.2)),
        ('gauss', bc.likelihoods.GaussianLocation(np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]), d, (None, 0.5)),
    ]


@pytest.mark.parametrize('n,d,s', [(1, 3, 5), (127, 7, 16), (129, 33, 64), (1000, 12, 97), (5000, 64, 100), (3001, 20, 112),
                                   (777, 9, 200), (2000, 5, 256)])
def test_store_free_column_sums_are_bit_identical(bc, n, d, s):
    """bc_project_colsum == bc_project + bc_phi_colsum, bit for bit: every model, ragged shapes, every sample-tile variant
    of the kernel (S <= 64, 97..100, <= 112, <= 208, <= 256), with constant (all-zero-feature) rows in the data."""
    rng = np.random.RandomState(n + 7 * d + s)
    for name, model, dz, betas in _models(bc, rng, d):
        Z = rng.randn(n, dz)
        if n > 10:
            Z[rng.choice(n, 3, replace=False), :d] = 0.      # constant rows take the NumPy-order mean path
        th = rng.randn(s, d) * 0.4
        prj = bc.DeviceBetaProjector(fixed(th), s, model)
        dd = bc.DeviceData(Z)
        for beta in betas:
            full = prj.project(dd) if beta is None else prj.project_f(dd, beta)
            want = full.colsum()
            got = prj.colsum(dd, beta=beta)
            assert got is not None
            assert np.array_equal(got, want), (name, beta, np.abs(got - want).max())
            # and from a live (not resident) array, the way project() accepts one
            got2 = prj.colsum(Z, beta=beta)
            assert np.array_equal(got2, want), (name, beta)


def test_store_free_column_sums_at_1m_rows(bc):
    """The configuration-2 scale: N = 1 000 003 ragged rows, D = 32, S = 100, linear regression and its beta-likelihood:
    identical bits, and the wide case S > 256 is declined (None) rather than approximated."""
    rng = np.random.RandomState(11)
    n, d, s = 1_000_003, 32, 100
    Z = rng.randn(n, d + 1)
    Z[[5, 70001, n - 1], :d] = 0.
    th = rng.randn(s, d) * 0.3
    prj = bc.DeviceBetaProjector(fixed(th), s, bc.likelihoods.LinearRegression(1.0))
    dd = bc.DeviceData(Z)
    for beta in (None, 0.1):
        full = prj.project(dd) if beta is None else prj.project_f(dd, beta)
        want = full.colsum()
        del full
        for _ in range(2):                    # the cached stats-only buffers are reused on the second call
            got = prj.colsum(dd, beta=beta)
            assert np.array_equal(got, want), np.abs(got - want).max()
    # a smaller projection through the same cached buffers afterwards, then a larger S
    dd2 = bc.DeviceData(Z[:3000])
    assert np.array_equal(prj.colsum(dd2, beta=0.1), prj.project_f(dd2, 0.1).colsum())
    th2 = rng.randn(300, d) * 0.3
    prj2 = bc.DeviceBetaProjector(fixed(th2), 300, bc.likelihoods.LinearRegression(1.0))
    assert prj2.colsum(dd2, beta=0.1) is None


@pytest.mark.parametrize('m', [1, 7, 130, 300])
def test_fused_gradient_matches_host_algebra(bc, m):
    """bc_vi_gradient == -corevecs.dot(scale * vecs.sum(0) - w.dot(corevecs)) / S computed on the host from the
    materialised projections (bcores.py:144-146): residual and gradient within 1e-12 relative."""
    rng = np.random.RandomState(m)
    n, d, s = 20000, 24, 100
    for name, model, dz, betas in _models(bc, rng, d):
        Z = rng.randn(n, dz)
        th = rng.randn(s, d) * 0.3
        prj = bc.DeviceBetaProjector(fixed(th), s, model)
        dd = bc.DeviceData(Z)
        core = Z[rng.choice(n, m, replace=False)]
        w = rng.uniform(0., 3., m)
        for beta in betas:
            for scale in (1., 2.5):
                vecs = prj.project(dd) if beta is None else prj.project_f(dd, beta)
                cv = np.asarray(prj.project(core) if beta is None else prj.project_f(core, beta))
                resid = scale * vecs.colsum() - w.dot(cv)
                want = -cv.dot(resid) / s
                got, r = prj.vi_gradient(dd, core, w, scale, beta=beta, want_resid=True)
                np.testing.assert_allclose(r, resid, rtol=1e-12, atol=1e-12 * np.abs(resid).max())
                np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12 * np.abs(want).max())


def test_fused_gradient_survives_library_calls_between_begin_and_end(bc):
    """Whatever the host does between bc_vi_gradient_begin and _end -- including calls into this library on the same context
    (a K4 round trip, another projection that re-stages Theta through the same pinned area, a second projector's column
    sums) -- the pending gradient is the one that was enqueued.  The coreset rows' projection runs on a side stream beside
    the data rows' launch: this is the test that their results are joined before the algebra."""
    rng = np.random.RandomState(77)
    n, d, s, m = 300_000, 32, 100, 40
    Z = rng.randn(n, d + 1)
    th = rng.randn(s, d) * 0.3
    model = bc.likelihoods.LinearRegression(1.3)
    prj = bc.DeviceBetaProjector(fixed(th), s, model)
    other = bc.DeviceProjector(fixed(rng.randn(s, d)), s, model)
    dd = bc.DeviceData(Z)
    core = Z[rng.choice(n, m, replace=False)]
    w = rng.uniform(0., 3., m)
    want, want_r = prj.vi_gradient(dd, core, w, 1.7, beta=0.2, want_resid=True)
    side = {}

    def meddle():
        side['gram'] = bc.weighted_gram(core, w)
        side['phi'] = other.project(Z[:5000]).colsum()
        side['cs'] = other.colsum(dd)
    for _ in range(3):
        got, got_r = prj.vi_gradient(dd, core, w, 1.7, beta=0.2, want_resid=True, overlap=meddle)
        assert np.array_equal(got, want) and np.array_equal(got_r, want_r)
    G, v = side['gram']
    np.testing.assert_allclose(G, (w[:, None] * core[:, :d]).T.dot(core[:, :d]), rtol=1e-12, atol=1e-12)
    assert np.array_equal(side['cs'], other.project(dd).colsum())


def make_sampler(Z, E):
    D = Z.shape[1] - 1

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
        mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)
    return sampler


@pytest.mark.parametrize('kind', ['bcores', 'svi'])
@pytest.mark.parametrize('n', [3000, 9000])
def test_fused_optimise_equals_general_path_and_oracle(bc, kind, n):
    """The whole construction through the fused gradient (default) and through the general path (fused_gradient=False:
    every gradient materialises Phi) give the same selections and weights to 1e-9, and both match the oracle; n = 9000
    runs on pinned (resident) rows, n = 3000 on a live array uploaded per call."""
    rng = np.random.RandomState(5 + n)
    D, S, its = 8, 64, 6
    X = rng.randn(n, D)
    y = X.dot(rng.randn(D)) + rng.randn(n)
    out = rng.choice(n, n // 10, replace=False)
    y[out] = rng.normal(10., .5, out.shape[0])
    Z = np.hstack((X, y[:, None]))
    E = rng.randn(S, D)
    sampler = make_sampler(Z, E)
    sched = lambda i: 0.1 / (1. + i)
    model = bc.likelihoods.LinearRegression(1.0)

    def make(fused):
        if kind == 'bcores':
            return bc.BetaCoreset(Z, bc.DeviceBetaProjector(sampler, S, model), opt_itrs=its, step_sched=sched, beta=0.1,
                                  learn_beta=False, fused_gradient=fused)
        return bc.SparseVICoreset(Z, bc.DeviceProjector(sampler, S, model), opt_itrs=its, step_sched=sched, fused_gradient=fused)
    if kind == 'bcores':
        ref = C.RefGreedyVI(Z, lambda pts, th: C.project_f(lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0), pts, th, 0.1),
                            lambda w, p: sampler(S, w, p), its, sched)
    else:
        ref = C.RefGreedyVI(Z, lambda pts, th: C.project(lambda z, t: M.linreg_loglik(z, t, 1.0), pts, th),
                            lambda w, p: sampler(S, w, p), its, sched)
    a, b = make(True), make(False)
    calls = {'n': 0}
    orig = a.ll_projector.vi_gradient

    def counted(*args, **kw):
        calls['n'] += 1
        return orig(*args, **kw)
    a.ll_projector.vi_gradient = counted
    for m in range(6):
        a.build(1, m + 1)
        b.build(1, m + 1)
        ref.build(1)
        np.testing.assert_array_equal(a.idcs, b.idcs)
        np.testing.assert_allclose(a.wts, b.wts, rtol=1e-9, atol=1e-13)
        np.testing.assert_array_equal(a.idcs, ref.idcs)
        np.testing.assert_allclose(a.wts, ref.wts, rtol=1e-5, atol=1e-12)
    assert calls['n'] == 6 * its          # every gradient of every build step went through the native call


def test_rank_order_sum_kernel_on_fabricated_gather(bc):
    """k_sum_rank_order (the device side of bc_comm_sum_doubles / bc_phi_colsum_all) on a fabricated [world][count]
    gathered buffer: world = 3 and 8, count not a multiple of the block size -- bit-identical to adding the ranks'
    vectors in rank order on the host (ShardComm.sum_in_rank_order)."""
    import ctypes as Ct
    from beta_cores_amd import _native as N
    from beta_cores_amd.device import _ptr
    ctx = bc.default_context()
    rng = np.random.RandomState(3)
    for world, count in [(3, 1000), (8, 104), (2, 1), (5, 70001)]:
        g = rng.randn(world, count) * 10.0 ** rng.uniform(-8, 8, size=(world, count))
        want = g[0].copy()
        for r in range(1, world):
            want = want + g[r]
        got = np.empty(count)
        N.call('bc_comm_rank_order_sum_selftest', ctx.h, _ptr(np.ascontiguousarray(g)), world, count, _ptr(got))
        assert np.array_equal(got, want), (world, count)


@pytest.mark.parametrize('kind', ['bcores', 'svi'])
def test_prefetching_sampler_keeps_results_and_rng_position(bc, kind):
    """bc.samplers.LinregPosteriorSampler draws the NEXT call's normals while the GPU works on the current gradient
    (bc_vi_gradient_begin / _end).  Same stream, same order, nothing consumed on anyone else's behalf: the coreset, the
    weights and the global RNG position after the builds equal those of the reference-style closure (and the oracle's)."""
    rng = np.random.RandomState(31)
    n, D, S, its = 6000, 9, 48, 7
    X = rng.randn(n, D)
    y = X.dot(rng.randn(D)) + rng.randn(n)
    Z = np.hstack((X, y[:, None]))
    model = bc.likelihoods.LinearRegression(1.0)
    sched = lambda i: 0.1 / (1. + i)

    def closure(sz, wts, pts):                      # zellner_neural_linear/main.py:119-124
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, D + 1))
        mu, L, _ = bc.weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + np.random.randn(sz, D).dot(L.T)

    def run(sampler):
        np.random.seed(5)
        if kind == 'bcores':
            alg = bc.BetaCoreset(Z, bc.DeviceBetaProjector(sampler, S, model), opt_itrs=its, step_sched=sched, beta=0.1, learn_beta=False)
        else:
            alg = bc.SparseVICoreset(Z, bc.DeviceProjector(sampler, S, model), opt_itrs=its, step_sched=sched)
        out = []
        for m in range(5):
            alg.build(1, m + 1)
            out.append((alg.idcs.copy(), alg.wts.copy()))
        return out, np.random.rand()
    pre = bc.samplers.LinregPosteriorSampler(np.zeros(D), np.eye(D), 1.0)
    hits = {'n': 0}
    orig = pre.prefetch

    def counted():
        hits['n'] += 1
        orig()
    pre.prefetch = counted
    a, ra = run(pre)
    b, rb = run(closure)
    assert hits['n'] == 5 * (its - 1)
    assert ra == rb                                 # the global stream stands where the closure leaves it
    for (ia, wa), (ib, wb) in zip(a, b):
        np.testing.assert_array_equal(ia, ib)
        assert np.array_equal(wa, wb)               # the very same normals, so the very same bits
