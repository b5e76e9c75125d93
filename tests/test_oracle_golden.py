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
