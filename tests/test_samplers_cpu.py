"""Host logic of bc.samplers (no GPU): the logistic drivers' Laplace sampler against golden F19 (the reference's
get_laplace, util/opt.py:9-33 == zellner_logreg/main.py:86-111) and the prefetch contract."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.mark.parametrize('S', [37, 100])
@pytest.mark.parametrize('diag', [False, True])
def test_logistic_laplace_matches_reference(S, diag):
    from beta_cores_amd.samplers import logistic_laplace
    g = load_golden('f19_logistic_greedy_vi')
    tag = 'S%d_' % S
    Z = g[tag + 'Z']
    mu, L, Li = logistic_laplace(g[tag + 'lap_w'], Z[g[tag + 'lap_rows']], np.zeros(Z.shape[1]), diag)
    # scipy's BFGS is shared, not restated: same routine, same inputs -> the reference's numbers to rounding
    np.testing.assert_allclose(mu, g['%slap%d_mu' % (tag, diag)], rtol=1e-9, atol=1e-12)
    Lg, Lig = g['%slap%d_L' % (tag, diag)], g['%slap%d_Li' % (tag, diag)]
    np.testing.assert_allclose(L, np.diag(Lg) if diag else Lg, rtol=1e-9, atol=1e-12)    # util/opt.py:27-29 hands back vectors
    np.testing.assert_allclose(Li, np.diag(Lig) if diag else Lig, rtol=1e-9, atol=1e-12)


def test_logistic_laplace_sampler_prior_and_stream():
    """An empty coreset gives the prior N(0, I) (main.py:140-142); the normals come from the given stream, one S x D block
    per call, and a prefetched block is the one the next call uses."""
    from beta_cores_amd.samplers import LogisticLaplaceSampler
    D, S = 5, 7
    smp = LogisticLaplaceSampler(np.zeros(D), rng=np.random.RandomState(3), solver='newton')
    th = smp(S, np.array([]), np.array([]))
    ref = np.random.RandomState(3)
    e0 = ref.randn(S, D)
    np.testing.assert_allclose(th, e0, rtol=0, atol=1e-7)          # mode 0 (to BFGS's tolerance), LSig = I
    smp.prefetch()
    smp.prefetch()                                                 # second one is a no-op
    th1 = smp(S, np.array([]), np.array([]))
    np.testing.assert_allclose(th1, ref.randn(S, D), rtol=0, atol=1e-7)
    th2 = smp(S, np.array([]), np.array([]))
    np.testing.assert_allclose(th2, ref.randn(S, D), rtol=0, atol=1e-7)


def test_prefetched_block_of_another_shape_is_an_error():
    from beta_cores_amd.samplers import LogisticLaplaceSampler
    smp = LogisticLaplaceSampler(np.zeros(4), rng=np.random.RandomState(0), solver='newton')
    smp(6, np.array([]), np.array([]))
    smp.prefetch()
    with pytest.raises(RuntimeError, match='prefetched'):
        smp(9, np.array([]), np.array([]))


def test_no_look_ahead_with_the_reference_mode_search(monkeypatch):
    """With solver='bfgs' a failing minimize() draws its restart perturbation BEFORE the sample normals (main.py:96-101, :144):
    prefetch() must not take the normals from the stream ahead of it.  A mode search that fails once, then succeeds: the
    stream is consumed as perturbation-then-normals, exactly like the reference's closure."""
    import beta_cores_amd.samplers as SM
    D, S = 4, 3
    rng = np.random.RandomState(11)
    Z = rng.randn(9, D)
    w = np.full(9, 2.)
    real = SM.minimize
    fails = [1]

    def flaky(*a, **k):
        if fails[0] > 0:
            fails[0] -= 1
            raise FloatingPointError('injected')
        return real(*a, **k)
    monkeypatch.setattr(SM, 'minimize', flaky)
    smp = SM.LogisticLaplaceSampler(np.full(D, 0.3), rng=np.random.RandomState(5))
    smp._shape = (S, D)
    smp.prefetch()                                                  # must be a no-op for 'bfgs'
    assert smp._ahead is None
    th = smp(S, w, Z)
    ref = np.random.RandomState(5)
    mu0 = np.full(D, 0.3)
    mu0 = mu0 + np.sqrt((mu0 ** 2).sum()) * 0.1 * ref.randn(D)     # the restart's draw comes first
    mu, L, _ = SM.logistic_laplace(w, Z, mu0)
    np.testing.assert_allclose(th, mu + ref.randn(S, D).dot(L.T), rtol=1e-9, atol=1e-12)


def test_logistic_beta_constant_at_zero_is_numpys():
    """likelihoods.LogisticRegression hands K1 the beta-likelihood's value at m = 0 computed by NumPy's array power --
    the bits the reference's N x S evaluation has for a data row z = 0 (golden F20 holds such rows: their centred value
    is c - mean(S copies of c), recomputed here from the constant alone)."""
    from beta_cores_amd.likelihoods import LogisticRegression
    g = load_golden('f20_logistic_beta_constant_rows')
    for S in (16, 100, 200):
        for beta in (0.1, 0.2, 0.5):
            c = LogisticRegression.beta_value_at_zero(beta)
            row = np.full((1, S), c)
            row -= row.mean(axis=1)[:, np.newaxis]
            np.testing.assert_array_equal(row[0], g['S%d_b%g_phi_const' % (S, beta)][0])
    p = LogisticRegression().params(0.1)
    assert p.shape == (2,) and p[0] == 0.1 and p[1] == LogisticRegression.beta_value_at_zero(0.1)
    for bad in (0., -0.5, 33.):        # the device body's power series covers 0 < beta <= 32 (csrc/bc_k1_math.h)
        with pytest.raises(ValueError):
            LogisticRegression().params(bad)
    assert LogisticRegression().params(32.)[0] == 32.


@pytest.mark.parametrize('S', [37, 100])
def test_newton_mode_is_the_reference_mode(S):
    """solver='newton' reaches the same (unique) mode as the reference's BFGS call: within BFGS's own stopping tolerance of
    the golden, and with a smaller gradient than the golden's point has."""
    from beta_cores_amd.samplers import logistic_laplace, _lr_grad_log_joint
    g = load_golden('f19_logistic_greedy_vi')
    tag = 'S%d_' % S
    Z = g[tag + 'Z']
    w, rows = g[tag + 'lap_w'], Z[g[tag + 'lap_rows']]
    mu, L, Li = logistic_laplace(w, rows, np.zeros(Z.shape[1]), False, solver='newton')
    np.testing.assert_allclose(mu, g[tag + 'lap0_mu'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(L, g[tag + 'lap0_L'], rtol=1e-4, atol=2e-6)
    gn = np.abs(_lr_grad_log_joint(rows[w > 0], mu, w[w > 0])).max()
    gb = np.abs(_lr_grad_log_joint(rows[w > 0], g[tag + 'lap0_mu'], w[w > 0])).max()
    assert gn <= 1e-12 and gn <= gb
    # a peaked posterior (weights N/M, the shape of a pre-initialised coreset): Newton still lands on BFGS's answer
    rng = np.random.RandomState(5)
    Zb = rng.randn(60, 12) * np.where(rng.rand(60) < 0.5, 1., -1.)[:, None]
    wb = np.full(60, 1e4)
    mu_n = logistic_laplace(wb, Zb, np.zeros(12), False, solver='newton')[0]
    mu_b = logistic_laplace(wb, Zb, np.zeros(12), False, solver='bfgs')[0]
    np.testing.assert_allclose(mu_n, mu_b, rtol=1e-4, atol=1e-6)
    with pytest.raises(ValueError):
        logistic_laplace(w, rows, np.zeros(Z.shape[1]), False, solver='lbfgs')


def test_newton_warm_start_reaches_the_same_mode():
    """LogisticLaplaceSampler(solver='newton') starts each mode search at the previous call's mode; the maximiser is unique, so
    the samples are the cold-start ones to rounding."""
    from beta_cores_amd.samplers import LogisticLaplaceSampler
    rng = np.random.RandomState(9)
    D, M, S = 10, 40, 6
    Z = rng.randn(M, D) * np.where(rng.rand(M) < 0.5, 1., -1.)[:, None]

    class E:
        def randn(self, n, d):
            return np.ones((n, d))
    warm = LogisticLaplaceSampler(np.zeros(D), rng=E(), solver='newton')
    for k in range(4):
        w = np.full(M, 50.) + 5. * k * rng.rand(M)
        cold = LogisticLaplaceSampler(np.zeros(D), rng=E(), solver='newton')
        np.testing.assert_allclose(warm(S, w, Z), cold(S, w, Z), rtol=1e-10, atol=1e-12)
    assert warm._mode is not None and cold._mode is not None


@pytest.mark.parametrize('S,sn,nm', [(37, 'laplace', 'svi'), (100, 'laprng', 'bcores'), (37, 'lapdiag', 'bcores')])
def test_newton_sampler_in_the_oracle_loop_against_the_goldens(S, sn, nm):
    """What solver='newton' does to the coreset, asked of the reference's goldens on the host alone (the oracle's greedy-VI
    loop + the product's sampler): selections exact, weights within 1e-5 -- 1.76e-5 for one weight of the first case, whose
    golden value carries BFGS's stopping noise (tests/test_gpu_coresets.py runs all ten cases through the device)."""
    from oracle import models_ref as M, coreset_ref as C
    from beta_cores_amd.samplers import LogisticLaplaceSampler
    g = load_golden('f19_logistic_greedy_vi')
    tag = 'S%d_' % S
    Z, E = g[tag + 'Z'], g[tag + 'E']
    beta, opt_itrs = float(g['beta']), int(g['opt_itrs'])

    class FixedNormals:
        def randn(self, n, d):
            return E
    smp = LogisticLaplaceSampler(np.zeros(Z.shape[1]), diag=(sn == 'lapdiag'), rng=None if sn == 'laprng' else FixedNormals(),
                                 solver='newton')
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(M.logistic_beta_lik, pts, th, beta)
    else:
        proj = lambda pts, th: C.project(M.logistic_loglik, pts, th)
    np.random.seed(190)
    worst = 0.
    with np.errstate(all='ignore'):
        smp(S, np.array([]), np.zeros((0, Z.shape[1])))       # the projector constructor draws once (projector.py:18,46)
        alg = C.RefGreedyVI(Z, proj, lambda w, p: smp(S, w, p), opt_itrs, lambda i: 0.5 / (1. + i))
        for m in range(5):
            alg.build(1, m + 1)
            np.testing.assert_array_equal(alg.idcs, g['%s%s_%s_allidcs_%d' % (tag, sn, nm, m)])
            gw = g['%s%s_%s_allw_%d' % (tag, sn, nm, m)]
            np.testing.assert_allclose(alg.wts, gw, rtol=2.5e-5, atol=1e-12)
            big = np.abs(gw) > 1e-9
            worst = max(worst, float(np.max(np.abs(alg.wts[big] - gw[big]) / np.abs(gw[big]), initial=0.)))
    assert worst <= 1e-5 or (S, sn, nm) == (37, 'laplace', 'svi')
    if sn == 'laprng':
        assert np.random.rand() == float(g['%s%s_%s_rng_after' % (tag, sn, nm)])
