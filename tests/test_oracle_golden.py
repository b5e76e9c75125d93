"""Pin the CPU oracle against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import snnls_ref as O
from oracle import models_ref as M
from oracle import coreset_ref as C
from conftest import load_golden

ALGS = dict(giga=O.RefGIGA, fw=O.RefFrankWolfe, omp=O.RefOrthoPursuit)


def stepwise(solver, steps):
    sel = np.full(steps, -1, dtype=np.int64)
    err = np.zeros(steps)
    lim = np.zeros(steps, dtype=np.int8)
    W = np.zeros((steps, solver.n))
    for m in range(steps):
        n0 = len(solver.trace)
        solver.build(1)
        if len(solver.trace) > n0:
            sel[m] = solver.trace[-1][0]
        err[m] = solver.error()
        lim[m] = solver.hit_limit
        W[m] = solver.weights()
    return sel, err, lim, W


F1 = load_golden('f1_snnls')


@pytest.mark.parametrize('case', list(F1['cases']))
@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
def test_f1_snnls_known_answers(case, alg):
    X = F1[case + '_X']
    steps = F1['%s_%s_sel' % (case, alg)].shape[0]
    sel, err, lim, W = stepwise(ALGS[alg](X.T, X.sum(axis=0)), steps)
    # same container, same BLAS as the generator -> expect identical selections and weights
    np.testing.assert_array_equal(sel, F1['%s_%s_sel' % (case, alg)])
    np.testing.assert_array_equal(lim, F1['%s_%s_lim' % (case, alg)])
    np.testing.assert_allclose(W, F1['%s_%s_W' % (case, alg)], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(err, F1['%s_%s_err' % (case, alg)], rtol=1e-9, atol=1e-12)


def test_f2_formulas_bit_identical():
    g = load_golden('f2_formulas')
    Z, th = g['lin_Z'], g['lin_th']
    for sig in (1.0, 2.5):
        assert np.array_equal(M.linreg_loglik(Z, th, sig), g['lin_ll_sig%g' % sig])
        for beta in (0.1, 0.2, 0.5):
            assert np.array_equal(M.linreg_beta_lik(Z, th, beta, sig), g['lin_bl_sig%g_b%g' % (sig, beta)])
    Zl, thl = g['log_Z'], g['log_th']
    assert np.array_equal(M.logistic_loglik(Zl, thl), g['log_ll'])
    for beta in (0.1, 0.2, 0.5):
        got = M.logistic_beta_lik(Zl, thl, beta)
        assert np.array_equal(got, g['log_bl_b%g' % beta])
        assert np.all(np.isfinite(got))
        # saturation limits: +1 for m -> +inf, -1/beta for m -> -inf
        assert got[4, 0] == 1.0 and abs(got[5, 0] + 1. / beta) < 1e-12
    X, thg = g['gau_X'], g['gau_th']
    for nm in ('iso', 'full'):
        Si, ld = g['gau_%s_Siginv' % nm], float(g['gau_%s_logdet' % nm])
        assert np.array_equal(M.gauss_loglik(X, thg, Si, ld), g['gau_%s_ll' % nm])
        for beta in (0.1, 0.5):
            assert np.array_equal(M.gauss_beta_lik(X, thg, beta, Si, ld), g['gau_%s_bl_b%g' % (nm, beta)])
            assert np.array_equal(M.gauss_beta_grad(X, thg, beta, Si, ld), g['gau_%s_bg_b%g' % (nm, beta)])


def _hilbert(data, ll, th, steps, solver):
    h = C.RefHilbert(data, ll, th, solver)
    sel = np.full(steps, -1, dtype=np.int64)
    err = np.zeros(steps)
    for m in range(steps):
        n0 = len(h.solver.trace)
        h.build(1, m + 1)
        if len(h.solver.trace) > n0:
            sel[m] = h.solver.trace[-1][0]
        err[m] = h.error()
    return h, sel, err


@pytest.mark.parametrize('nm', ['ll', 'bl'])
@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
def test_f3_hilbert_linreg(nm, alg):
    g = load_golden('f3_hilbert_linreg')
    Z, th = g['Z'], g['th']
    ll = (lambda z, t: M.linreg_loglik(z, t, 1.0)) if nm == 'll' else (lambda z, t: M.linreg_beta_lik(z, t, 0.1, 1.0))
    assert np.array_equal(C.project(ll, Z, th), g['phi_' + nm])
    key = '%s_%s_' % (nm, alg)
    steps = g[key + 'sel'].shape[0]
    h, sel, err = _hilbert(Z, ll, th, steps, ALGS[alg])
    np.testing.assert_array_equal(sel, g[key + 'sel'])
    np.testing.assert_allclose(err, g[key + 'err'], rtol=1e-9)
    np.testing.assert_array_equal(h.idcs, g[key + 'idcs'])
    np.testing.assert_allclose(h.wts, g[key + 'wts'], rtol=1e-12)
    np.testing.assert_array_equal(h.solver.b, g[key + 'b'])
    h.optimize()
    np.testing.assert_array_equal(h.idcs, g[key + 'opt_idcs'])
    np.testing.assert_allclose(h.wts, g[key + 'opt_wts'], rtol=1e-9)


def test_f4_hilbert_logistic_and_gauss():
    g = load_golden('f4_hilbert_logistic_gauss')
    Z, th = g['log_Z'], g['log_th']
    for nm, ll in (('ll', M.logistic_loglik), ('bl', lambda z, t: M.logistic_beta_lik(z, t, 0.1))):
        key = 'log_%s_' % nm
        steps = g[key + 'sel'].shape[0]
        h, sel, err = _hilbert(Z, ll, th, steps, O.RefGIGA)
        np.testing.assert_array_equal(sel, g[key + 'sel'])
        np.testing.assert_array_equal(h.idcs, g[key + 'idcs'])
        np.testing.assert_allclose(h.wts, g[key + 'wts'], rtol=1e-12)
    X, thg, Si, ld = g['gau_X'], g['gau_th'], g['gau_Siginv'], float(g['gau_logdet'])
    key = 'gau_ll_'
    steps = g[key + 'sel'].shape[0]
    h, sel, err = _hilbert(X, lambda x, t: M.gauss_loglik(x, t, Si, ld), thg, steps, O.RefGIGA)
    np.testing.assert_array_equal(sel, g[key + 'sel'])
    np.testing.assert_array_equal(h.idcs, g[key + 'idcs'])
    np.testing.assert_allclose(h.wts, g[key + 'wts'], rtol=1e-12)


@pytest.mark.parametrize('nm', ['bcores', 'svi'])
def test_f5_greedy_vi(nm):
    g = load_golden('f5_greedy_vi')
    Z, E = g['Z'], g['E']
    D = Z.shape[1] - 1
    beta, opt_itrs = float(g['beta']), int(g['opt_itrs'])

    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
        mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0), pts, th, beta)
    else:
        proj = lambda pts, th: C.project(lambda z, t: M.linreg_loglik(z, t, 1.0), pts, th)
    alg = C.RefGreedyVI(Z, proj, sampler, opt_itrs, lambda i: 0.1 / (1. + i))
    for m in range(5):
        alg.build(1)
        np.testing.assert_array_equal(alg.idcs, g['%s_allidcs_%d' % (nm, m)])
        np.testing.assert_allclose(alg.wts, g['%s_allw_%d' % (nm, m)], rtol=1e-10, atol=1e-14)
        w, p, i = alg.get()
        np.testing.assert_array_equal(i, g['%s_idcs_%d' % (nm, m)])


@pytest.mark.parametrize('nm', ['bcores', 'svi'])
def test_f8_grouped_vi(nm):
    g = load_golden('f8_grouped_vi')
    Z, E = g['Z'], g['E']
    groups = [list(r) for r in g['groups']]
    D = Z.shape[1] - 1
    opt_itrs = int(g['opt_itrs'])

    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
        mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0), pts, th, 0.1)
    else:
        proj = lambda pts, th: C.project(lambda z, t: M.linreg_loglik(z, t, 1.0), pts, th)
    alg = C.RefGreedyVI(Z, proj, sampler, opt_itrs, lambda i: 0.1 / (1. + i), groups=groups)
    for m in range(4):
        alg.build(1)
        np.testing.assert_array_equal(alg.idcs, g['%s_allidcs_%d' % (nm, m)])
        np.testing.assert_array_equal(alg.selected_groups, g['%s_groups_%d' % (nm, m)])
        np.testing.assert_allclose(alg.wts, g['%s_allw_%d' % (nm, m)], rtol=1e-10, atol=1e-14)


def test_f6_weighted_post():
    g = load_golden('f6_weighted_post')
    for D in (8, 64):
        mu, L, Linv = M.linreg_weighted_post(g['D%d_th0' % D], g['D%d_Sig0inv' % D], 1.7, g['D%d_Z' % D], g['D%d_w' % D])
        assert np.array_equal(mu, g['D%d_mu' % D]) and np.array_equal(L, g['D%d_L' % D]) and np.array_equal(Linv, g['D%d_Linv' % D])
    mu, L, Linv = M.gauss_weighted_post(np.zeros(8), np.eye(8), g['g_Siginv'], g['g_X'], g['g_w'])
    assert np.array_equal(mu, g['g_mu']) and np.array_equal(L, g['g_L'])


def test_f7_nn_opt():
    g = load_golden('f7_nn_opt')
    Q, c, x0 = g['Q'], g['c'], g['x0']
    grd = lambda x: Q.dot(x) - c
    assert np.array_equal(C.nn_opt(x0, grd, opt_itrs=50, step_sched=lambda i: 0.5 / (1. + i)), g['nn'])
    assert np.array_equal(C.partial_nn_opt(x0, grd, np.arange(0, 12, 2), opt_itrs=50, step_sched=lambda i: 0.5 / (1. + i)), g['pnn'])


# ------------------------------------------------------------------ F9-F15 (round 2)
def _gauss_sampler(mu0, Sig0inv, Siginv, dz, S, E=None):
    """sampler_w of zellner_gaussian/main.py:90-95; E = None draws from the global RNG."""
    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, dz))
        muw, LSigw, _ = M.gauss_weighted_post(mu0, Sig0inv, Siginv, pts, wts)
        e = np.random.randn(S, muw.shape[0]) if E is None else E
        return muw + e.dot(LSigw.T)
    return sampler


@pytest.mark.parametrize('nm', ['bcores', 'svi'])
def test_f9_subsampled_gaussian(nm):
    g = load_golden('f9_subsampled_gaussian')
    X, Si, ld = g['X'], g['Siginv'], float(g['logdet'])
    d, S = X.shape[1], 40
    np.random.seed(90)
    sampler = _gauss_sampler(np.zeros(d), np.eye(d), Si, d, S)
    sampler(np.array([]), np.array([]))                 # the projector constructor draws once (projector.py:18,46)
    if nm == 'bcores':
        proj = lambda p, th: C.project_f(lambda x, t, b: M.gauss_beta_lik(x, t, b, Si, ld), p, th, .1)
    else:
        proj = lambda p, th: C.project(lambda x, t: M.gauss_loglik(x, t, Si, ld), p, th)
    alg = C.RefGreedyVI(X, proj, sampler, 8, lambda i: 0.1 / (1. + i), n_subsample_select=150, n_subsample_opt=60)
    for m in range(6):
        alg.build(1, m + 1)
        np.testing.assert_array_equal(alg.idcs, g['%s_allidcs_%d' % (nm, m)])
        np.testing.assert_allclose(alg.wts, g['%s_allw_%d' % (nm, m)], rtol=1e-10, atol=1e-14)
    assert np.random.rand() == float(g['%s_rng_after' % nm])


@pytest.mark.parametrize('nm', ['imp', 'unif'])
def test_f10_sampling_solvers(nm):
    g = load_golden('f10_sampling')
    phi = g['phi']
    cls = O.RefImportanceSampling if nm == 'imp' else O.RefUniformSampling
    np.random.seed(100)
    steps = g[nm + '_sel'].shape[0]
    sel, err, lim, W = stepwise(cls(phi.T, phi.sum(axis=0)), steps)
    np.testing.assert_array_equal(sel, g[nm + '_sel'])
    np.testing.assert_array_equal(lim, g[nm + '_lim'])
    np.testing.assert_allclose(W, g[nm + '_W'], rtol=1e-13, atol=0)
    np.testing.assert_allclose(err, g[nm + '_err'], rtol=1e-10)
    assert np.random.rand() == float(g[nm + '_rng_after'])
    np.random.seed(101)
    h = C.RefHilbert(g['Z'], lambda z, t: M.linreg_loglik(z, t, 1.0), g['th'], cls)
    h.build(steps, steps)
    np.testing.assert_array_equal(h.idcs, g[nm + '_h_idcs'])
    np.testing.assert_allclose(h.wts, g[nm + '_h_wts'], rtol=1e-13)
    np.testing.assert_allclose(h.error(), float(g[nm + '_h_err']), rtol=1e-10)


@pytest.mark.parametrize('nm', ['bcores', 'svi'])
def test_f11_grouped_subsampled(nm):
    g = load_golden('f11_grouped_subsampled')
    Z, E = g['Z'], g['E']
    groups = [list(r) for r in g['groups']]
    D = Z.shape[1] - 1

    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
        mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0), pts, th, 0.1)
    else:
        proj = lambda pts, th: C.project(lambda z, t: M.linreg_loglik(z, t, 1.0), pts, th)
    np.random.seed(110)
    alg = C.RefGreedyVI(Z, proj, sampler, int(g['opt_itrs']), lambda i: 0.1 / (1. + i), groups=groups,
                        n_subsample_select=8, n_subsample_opt=50, size_check_always=(nm == 'svi'))
    for m in range(5):
        alg.build(1, 12 * (m + 1))
        np.testing.assert_array_equal(alg.idcs, g['%s_allidcs_%d' % (nm, m)])
        np.testing.assert_array_equal(alg.selected_groups, g['%s_groups_%d' % (nm, m)])
        np.testing.assert_allclose(alg.wts, g['%s_allw_%d' % (nm, m)], rtol=1e-10, atol=1e-14)
    assert np.random.rand() == float(g['%s_rng_after' % nm])


def np_mean_of_constant(c, n):
    """NumPy's pairwise sum of n copies of c, divided by n -- the model the K1 kernel implements for constant
    rows (csrc/bc_internal.h: bc_np_sum_const_*)."""
    def leaf(n):
        if n < 8:
            r = 0.
            for _ in range(n):
                r = r + c
            return r
        r = c
        for _ in range(1, n // 8):
            r = r + c
        res = ((r + r) + (r + r)) + ((r + r) + (r + r))
        for _ in range(n - (n // 8) * 8):
            res = res + c
        return res

    def rec(n):
        if n <= 128:
            return leaf(n)
        n2 = n // 2
        n2 -= n2 % 8
        return rec(n2) + rec(n - n2)
    return rec(n) / n


def test_constant_row_mean_model_matches_numpy():
    rng = np.random.RandomState(0)
    for S in list(range(1, 300)) + [500, 1000, 1024, 4097]:
        cs = np.concatenate((rng.randn(40) * 10. ** rng.uniform(-3, 3, 40), [-np.log(2.), 0., 1., -0.5 * np.log(2 * np.pi)]))
        a = np.repeat(cs[:, None], S, axis=1)
        want = a.mean(axis=1)
        got = np.array([np_mean_of_constant(float(c), S) for c in cs])
        assert np.array_equal(want, got), S


@pytest.mark.parametrize('S', [100, 200])
def test_f12_constant_rows(S):
    g = load_golden('f12_constant_rows')
    za = g['zero_at']
    for tag, ll, algs in (('lin_S%d_' % S, lambda z, t: M.linreg_loglik(z, t, 1.0), ('giga', 'fw')),
                          ('log_S%d_' % S, M.logistic_loglik, ('giga',))):
        Z, th = g[tag + 'Z'], g[tag + 'th']
        phi = C.project(ll, Z, th)
        assert np.array_equal(phi[za], g[tag + 'phi_const'])
        kept = np.sqrt((phi ** 2).sum(axis=1)) > 0.
        assert np.array_equal(kept, g[tag + 'kept'])
        # every constant row: c - mean(c x S) with the pairwise-sum model
        raw = ll(Z[za], th)
        for r in range(len(za)):
            assert np.all(raw[r] == raw[r, 0])
            assert phi[za[r], 0] == raw[r, 0] - np_mean_of_constant(float(raw[r, 0]), S)
        for an in algs:
            steps = g['%s%s_sel' % (tag, an)].shape[0]
            h, sel, err = _hilbert(Z, ll, th, steps, ALGS[an])
            np.testing.assert_array_equal(sel, g['%s%s_sel' % (tag, an)])
            np.testing.assert_array_equal(h.idcs, g['%s%s_idcs' % (tag, an)])
            np.testing.assert_allclose(h.wts, g['%s%s_wts' % (tag, an)], rtol=1e-12)


@pytest.mark.parametrize('S', [16, 100])
@pytest.mark.parametrize('nm', ['bcores', 'svi'])
def test_f13_greedy_vi_zero_rows(S, nm):
    g = load_golden('f13_greedy_vi_zero_rows')
    Z, E = g['S%d_Z' % S], g['S%d_E' % S]
    D = Z.shape[1] - 1

    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
        mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0), pts, th, 0.1)
    else:
        proj = lambda pts, th: C.project(lambda z, t: M.linreg_loglik(z, t, 1.0), pts, th)
    alg = C.RefGreedyVI(Z, proj, sampler, 5, lambda i: 0.1 / (1. + i))
    for m in range(4):
        alg.build(1, m + 1)
        np.testing.assert_array_equal(alg.idcs, g['S%d_%s_allidcs_%d' % (S, nm, m)])
        np.testing.assert_array_equal(alg.wts, g['S%d_%s_allw_%d' % (S, nm, m)])


def run_zellner_gaussian_oracle(nm, params):
    """examples/zellner_gaussian/main.py:33-167 on the oracle classes (statement order = RNG order)."""
    N, d, M_, opt_itrs, n_sub_opt, n_sub_sel, proj_dim, tr = [int(v) for v in params]
    pihat_noise, i0 = 0.75, 0.1
    np.random.seed(tr)
    mu0, Sig0 = np.zeros(d), np.eye(d)
    Sig = 500 * np.eye(d)
    th = np.zeros(d)
    Sig0inv, Siginv = np.linalg.inv(Sig0), np.linalg.inv(Sig)
    logdetSig = np.linalg.slogdet(Sig)[1]
    X = np.random.multivariate_normal(th, Sig, N)
    mup, LSigp, LSigpInv = M.gauss_weighted_post(mu0, Sig0inv, Siginv, X, np.ones(X.shape[0]))
    Sigp, SigpInv = LSigp.dot(LSigp.T), LSigpInv.dot(LSigpInv.T)
    Xc = np.concatenate((X, np.random.multivariate_normal(th + 200, 0.5 * Sig, int(N / 50.)),
                         np.random.multivariate_normal(th + 150, 0.1 * Sig, int(N / 50.)),
                         np.random.multivariate_normal(th, 10 * Sig, int(N / 10.))))
    th_opt = mup + np.random.randn(proj_dim, d).dot(LSigp.T)                       # prj_optimal's constructor draw
    U = np.random.rand()
    muhat = U * mup + (1. - U) * mu0
    Sighat = U * Sigp + (1. - U) * Sig0
    muhat += pihat_noise * np.sqrt((muhat ** 2).sum()) * np.random.randn(muhat.shape[0])
    Sighat *= np.exp(-2 * pihat_noise * np.fabs(np.random.randn()))
    LSighat = np.linalg.cholesky(Sighat)
    th_real = mup + np.random.randn(proj_dim, d).dot(LSighat.T)                    # prj_realistic's
    sampler_w = _gauss_sampler(mu0, Sig0inv, Siginv, d, proj_dim)
    sampler_w(np.array([]), np.array([]))                                         # prj_w's
    sampler_w(np.array([]), np.array([]))                                         # prj_bw's
    ll = lambda x, t: M.gauss_loglik(x, t, Siginv, logdetSig)
    sched = lambda i: i0 / (1. + i)
    if nm == 'BCORES':
        proj = lambda p, t: C.project_f(lambda x, tt, b: M.gauss_beta_lik(x, tt, b, Siginv, logdetSig), p, t, .1)
        alg = C.RefGreedyVI(Xc, proj, sampler_w, opt_itrs, sched, n_subsample_select=n_sub_sel, n_subsample_opt=n_sub_opt)
    elif nm == 'SVI':
        alg = C.RefGreedyVI(Xc, lambda p, t: C.project(ll, p, t), sampler_w, opt_itrs, sched,
                            n_subsample_select=n_sub_sel, n_subsample_opt=n_sub_opt, size_check_always=True)
    elif nm == 'BPSVI':
        alg = C.RefBatchPSVI(Xc, ll, lambda x, t: M.gauss_grad_x_loglik(x, t, Siginv), sampler_w, opt_itrs,
                             n_subsample_opt=n_sub_opt, step_sched=lambda m: sched,
                             projector_draw=False)                                 # (prj_w's constructor draw was made above)
    else:
        alg = C.RefHilbert(Xc, ll, th_opt if nm == 'GIGAO' else th_real)
    w, p, idl = [np.array([0.])], [np.zeros((1, d))], [np.zeros(0, dtype=np.int64)]
    fork_state = np.random.get_state()
    for m in range(1, M_ + 1):
        if nm == 'BPSVI':
            np.random.set_state(fork_state)                                        # main.py:126-135: forked pool children
        alg.build(1, m)
        got = alg.get()
        w.append(got[0].copy()); p.append(got[1].copy()); idl.append(got[2].copy())
    if nm == 'BPSVI':
        np.random.set_state(fork_state)
        run_zellner_gaussian_oracle.last_points = p
    rkl, fkl = np.zeros(M_ + 1), np.zeros(M_ + 1)
    for m in range(M_ + 1):
        muw, LSigw, LSigwInv = M.gauss_weighted_post(mu0, Sig0inv, Siginv, p[m], w[m])
        rkl[m] = M.gaussian_KL(muw, LSigw.dot(LSigw.T), mup, SigpInv)
        fkl[m] = M.gaussian_KL(mup, Sigp, muw, LSigwInv.dot(LSigwInv.T))
    return w, idl, rkl, fkl, np.random.rand()


@pytest.mark.parametrize('nm', ['BCORES', 'SVI', 'GIGAO', 'GIGAR'])
def test_f14_zellner_gaussian_driver(nm):
    g = load_golden('f14_zellner_gaussian_driver')
    w, idl, rkl, fkl, rng_after = run_zellner_gaussian_oracle(nm, g['params'])
    for m in range(len(w)):
        np.testing.assert_array_equal(idl[m], g['%s_idcs_%d' % (nm, m)])
        np.testing.assert_allclose(w[m], g['%s_w_%d' % (nm, m)], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(rkl, g[nm + '_rkl'], rtol=1e-8)
    np.testing.assert_allclose(fkl, g[nm + '_fkl'], rtol=1e-8)
    assert rng_after == float(g[nm + '_rng_after'])


@pytest.mark.parametrize('tag,nsub', [('full', None), ('sub', 80)])
def test_f15_learn_beta(tag, nsub):
    g = load_golden('f15_learn_beta')
    X, E, Si, ld = g['X'], g['E'], g['Siginv'], float(g['logdet'])
    d, S = X.shape[1], E.shape[0]
    sampler = _gauss_sampler(np.zeros(d), np.eye(d), Si, d, S, E=E)
    proj = lambda p, th, b: C.project_f(lambda x, t, bb: M.gauss_beta_lik(x, t, bb, Si, ld), p, th, b)
    bgrad = lambda p, th, b: C.project_f(lambda x, t, bb: M.gauss_beta_grad(x, t, bb, Si, ld), p, th, b)
    mk = lambda **kw: C.RefGreedyVI(X, proj, sampler, 8, lambda i: 0.1 / (1. + i), n_subsample_select=nsub,
                                    n_subsample_opt=nsub, beta=.3, learn_beta=True, beta_grad=bgrad, **kw)
    np.random.seed(150)
    alg = mk()
    alg.build(1, 1)
    np.testing.assert_array_equal(alg.idcs, g[tag + '_one_allidcs'])
    np.testing.assert_allclose(alg.wts, g[tag + '_one_allw'], rtol=1e-10)
    np.testing.assert_allclose(alg.beta, float(g[tag + '_one_beta']), rtol=1e-10)
    ii = g['init_idcs']
    alg = mk(wts=np.full(6, X.shape[0] / 6.), idcs=ii, pts=X[ii])
    alg.build(1, 7)
    for m in range(3):
        if m > 0:
            alg.optimize()
        np.testing.assert_array_equal(alg.idcs, g['%s_init_allidcs_%d' % (tag, m)])
        np.testing.assert_allclose(alg.wts, g['%s_init_allw_%d' % (tag, m)], rtol=1e-10)
        np.testing.assert_allclose(alg.beta, float(g['%s_init_beta_%d' % (tag, m)]), rtol=1e-10)
    assert np.random.rand() == float(g['%s_rng_after' % tag])


def test_f16_x_gradient_formulas():
    g = load_golden('f16_bpsvi')
    z, th = g['lin_z'], g['lin_th']
    for sg in (1.0, 2.5):
        np.testing.assert_allclose(M.linreg_grad_x_loglik(z, th, sg), g['lin_grad_%g' % sg], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(M.logistic_grad_z_loglik(g['log_z'], th), g['log_grad'], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(M.gauss_grad_x_loglik(g['gau_x'], th, g['gau_Siginv']), g['gau_grad'], rtol=1e-13, atol=1e-15)
    lls, glls = C.project_grad(lambda a, b: M.linreg_loglik(a, b, 2.5), lambda a, b: M.linreg_grad_x_loglik(a, b, 2.5), z, th)
    np.testing.assert_allclose(lls, g['lin_proj_lls'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(glls, g['lin_proj_glls'], rtol=1e-12, atol=1e-13)
    assert np.abs(glls.mean(axis=2)).max() < 1e-13          # centred over the coordinate axis, as the reference writes it


def f16_problem(g, tag):
    """(data, loglik, grad_loglik, sampler(wts, pts)) of fixture F16's two models"""
    if tag == 'g':
        X, E, Si, ld = g['g_X'], g['g_E'], g['g_Siginv'], float(g['g_logdet'])
        d = X.shape[1]
        return (X, lambda x, t: M.gauss_loglik(x, t, Si, ld), lambda x, t: M.gauss_grad_x_loglik(x, t, Si),
                _gauss_sampler(np.zeros(d), np.eye(d), Si, d, E.shape[0], E=E))
    Z, E, sg = g['l_Z'], g['l_E'], float(g['l_sigsq'])
    D = Z.shape[1] - 1

    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, D + 1))
        muw, LSigw, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), sg, pts, wts)
        return muw + E.dot(LSigw.T)
    return Z, lambda a, b: M.linreg_loglik(a, b, sg), lambda a, b: M.linreg_grad_x_loglik(a, b, sg), sampler


@pytest.mark.parametrize('tag', ['g', 'l'])
@pytest.mark.parametrize('mode,nsub', [('full', None), ('sub', 60)])
def test_f16_batch_psvi(tag, mode, nsub):
    g = load_golden('f16_bpsvi')
    data, ll, gl, sampler = f16_problem(g, tag)
    np.random.seed(160)
    alg = C.RefBatchPSVI(data, ll, gl, sampler, 6, n_subsample_opt=nsub, step_sched=lambda m: lambda i: 0.5 / (1. + i))
    for sz in (3, 5):
        alg.build(1, sz)
        k = '%s_%s_%d' % (tag, mode, sz)
        np.testing.assert_array_equal(alg.idcs, g[k + '_idcs'])
        np.testing.assert_allclose(alg.wts, g[k + '_wts'], rtol=1e-10)
        np.testing.assert_allclose(alg.pts, g[k + '_pts'], rtol=1e-9, atol=1e-11)
    assert np.random.rand() == float(g['%s_%s_rng_after' % (tag, mode)])


def test_f18_zellner_gaussian_bpsvi():
    """the driver's BPSVI row on the oracle's RefBatchPSVI (F18; RAND is a host class of the product, pinned by F17 and
    tests/test_sampling_coreset_cpu.py)"""
    g = load_golden('f18_zellner_gaussian_bpsvi_rand')
    w, idl, rkl, fkl, rng_after = run_zellner_gaussian_oracle('BPSVI', g['params'])
    p = run_zellner_gaussian_oracle.last_points
    for m in range(1, len(w)):
        np.testing.assert_array_equal(idl[m], g['BPSVI_idcs_%d' % m])
        np.testing.assert_allclose(w[m], g['BPSVI_w_%d' % m], rtol=1e-9)
        np.testing.assert_allclose(p[m], g['BPSVI_p_%d' % m], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(rkl, g['BPSVI_rkl'], rtol=1e-7)
    np.testing.assert_allclose(fkl, g['BPSVI_fkl'], rtol=1e-7)
    assert rng_after == float(g['BPSVI_rng_after'])



# ---- F19 / F20: the greedy-VI coresets on the logistic model (BASELINE config 3) and its constant rows
def _f19_sampler(sn, Z, E, th_fixed):
    D = Z.shape[1]

    def lap(diag, normals):
        def sampler(wts, pts):
            if pts.shape[0] == 0:
                wts, pts = np.zeros(1), np.zeros((1, D))
            mu, L, _ = M.logistic_laplace(wts, pts, np.zeros(D), diag)
            return mu + normals(E.shape[0], D).dot(L.T)
        return sampler
    return dict(fixed=lambda wts, pts: th_fixed, laplace=lap(False, lambda n, d: E),
                laprng=lap(False, lambda n, d: np.random.randn(n, d)), lapdiag=lap(True, lambda n, d: E))[sn]


@pytest.mark.parametrize('S', [37, 100])
@pytest.mark.parametrize('sn,nm', [('fixed', 'bcores'), ('fixed', 'svi'), ('laplace', 'bcores'), ('laplace', 'svi'),
                                   ('laprng', 'bcores'), ('laprng', 'svi'), ('lapdiag', 'bcores')])
def test_f19_logistic_greedy_vi(S, sn, nm):
    g = load_golden('f19_logistic_greedy_vi')
    tag = 'S%d_' % S
    Z, E, th_fixed = g[tag + 'Z'], g[tag + 'E'], g[tag + 'th_fixed']
    beta, opt_itrs = float(g['beta']), int(g['opt_itrs'])
    sampler = _f19_sampler(sn, Z, E, th_fixed)
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(M.logistic_beta_lik, pts, th, beta)
    else:
        proj = lambda pts, th: C.project(M.logistic_loglik, pts, th)
    np.random.seed(190)
    with np.errstate(all='ignore'):
        sampler(np.array([]), np.array([]))                 # the projector constructor draws once (projector.py:18,46)
        alg = C.RefGreedyVI(Z, proj, sampler, opt_itrs, lambda i: 0.5 / (1. + i))
        for m in range(5):
            alg.build(1, m + 1)
            np.testing.assert_array_equal(alg.idcs, g['%s%s_%s_allidcs_%d' % (tag, sn, nm, m)])
            np.testing.assert_allclose(alg.wts, g['%s%s_%s_allw_%d' % (tag, sn, nm, m)], rtol=1e-9, atol=1e-13)
    if sn == 'laprng':
        assert np.random.rand() == float(g['%s%s_%s_rng_after' % (tag, sn, nm)])


@pytest.mark.parametrize('S', [37, 100])
@pytest.mark.parametrize('diag', [False, True])
def test_f19_laplace_fit(S, diag):
    g = load_golden('f19_logistic_greedy_vi')
    tag = 'S%d_' % S
    Z = g[tag + 'Z']
    mu, L, Li = M.logistic_laplace(g[tag + 'lap_w'], Z[g[tag + 'lap_rows']], np.zeros(Z.shape[1]), diag)
    np.testing.assert_allclose(mu, g['%slap%d_mu' % (tag, diag)], rtol=1e-9, atol=1e-12)
    Lg = g['%slap%d_L' % (tag, diag)]
    np.testing.assert_allclose(L, np.diag(Lg) if diag else Lg, rtol=1e-9, atol=1e-12)       # util/opt.py:27-29 returns vectors
    Lig = g['%slap%d_Li' % (tag, diag)]
    np.testing.assert_allclose(Li, np.diag(Lig) if diag else Lig, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('S', [16, 100, 200])
def test_f20_logistic_beta_constant_rows(S):
    g = load_golden('f20_logistic_beta_constant_rows')
    Z, th, zero_at = g['S%d_Z' % S], g['S%d_th' % S], g['zero_at']
    with np.errstate(all='ignore'):
        for beta in (0.1, 0.2, 0.5):
            phi = C.project_f(M.logistic_beta_lik, Z, th, beta)
            np.testing.assert_array_equal(phi[zero_at], g['S%d_b%g_phi_const' % (S, beta)])
            np.testing.assert_array_equal(np.sqrt((phi ** 2).sum(axis=1)) > 0., g['S%d_b%g_norm_pos' % (S, beta)])
        for beta in ((0.1,) if S != 100 else (0.1, 0.5)):
            alg = C.RefGreedyVI(Z, lambda pts, t: C.project_f(M.logistic_beta_lik, pts, t, beta), lambda w, p: th, 5,
                                lambda i: 0.5 / (1. + i))
            for m in range(4):
                alg.build(1, m + 1)
                np.testing.assert_array_equal(alg.idcs, g['S%d_b%g_allidcs_%d' % (S, beta, m)])
                np.testing.assert_allclose(alg.wts, g['S%d_b%g_allw_%d' % (S, beta, m)], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize('nm', ['bcores', 'svi'])
def test_f21_logistic_subsampled(nm):
    """The logistic drivers' actual wiring (zellner_logreg/main.py:152-160): sub-sampled tangent spaces, Laplace sampler on the
    global NumPy stream (its randn before the sub-sample's randint, bcores.py:39 then :53)."""
    g = load_golden('f21_logistic_subsampled')
    Z = g['Z']
    D, S = Z.shape[1], 40

    def sampler(wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, D))
        mu, L, _ = M.logistic_laplace(wts, pts, np.zeros(D), False)
        return mu + np.random.randn(S, D).dot(L.T)
    if nm == 'bcores':
        proj = lambda pts, th: C.project_f(M.logistic_beta_lik, pts, th, 0.1)
    else:
        proj = lambda pts, th: C.project(M.logistic_loglik, pts, th)
    np.random.seed(210)
    with np.errstate(all='ignore'):
        sampler(np.array([]), np.array([]))                 # the projector constructor draws once (projector.py:18,46)
        alg = C.RefGreedyVI(Z, proj, sampler, 8, lambda i: 0.5 / (1. + i), n_subsample_select=150, n_subsample_opt=60)
        for m in range(6):
            alg.build(1, m + 1)
            np.testing.assert_array_equal(alg.idcs, g['%s_allidcs_%d' % (nm, m)])
            np.testing.assert_allclose(alg.wts, g['%s_allw_%d' % (nm, m)], rtol=1e-9, atol=1e-13)
    assert np.random.rand() == float(g['%s_rng_after' % nm])


def test_f22_logistic_beta_exp_overflow():
    """model_lr.py:85 past np.exp's overflow with small beta: (1 + inf)**(-beta) == 0 exactly, the value jumps to 1."""
    g = load_golden('f22_logistic_beta_overflow')
    with np.errstate(over='ignore'):
        for beta in (0.01, 0.05, 0.1):
            got = M.logistic_beta_lik(g['Z'], g['th'], beta)
            np.testing.assert_array_equal(got, g['bl_b%g' % beta])
    i709, i710 = list(g['mags']).index(709.78), list(g['mags']).index(709.79)
    assert g['bl_b0.01'][i710, 0] == 1.0 and 0.916 < g['bl_b0.01'][i709, 0] < 0.917
