#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference lives at /root/reference and
never travels):   python tests/golden/make_golden.py

The stock ``import bayesiancoresets`` fails with an ordinary
ModuleNotFoundError (missing ``dpbpsvi`` / ``iwg`` modules), so the reference's
sub-modules are imported under empty stub parent packages.
Nothing in /root/reference is modified, copied or byte-compiled.  The fixtures
are data only: seeded inputs and the outputs the reference produced for them.
"""
import os
import sys
import types
import io
import contextlib

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
import numpy as np
import scipy

REF = os.environ.get('BC_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    root = _stub('bayesiancoresets', REF + '/bayesiancoresets')
    root.util = _stub('bayesiancoresets.util', REF + '/bayesiancoresets/util', TOL=1e-12)
    _stub('bayesiancoresets.coreset', REF + '/bayesiancoresets/coreset')
    _stub('iwg')
    sys.path.insert(0, REF + '/examples/common')
    import bayesiancoresets.util.errors  # noqa
    import bayesiancoresets.util.opt as opt
    import bayesiancoresets.snnls as snnls
    import bayesiancoresets.coreset.hilbert as hilbert
    import bayesiancoresets.coreset.bcores as bcores
    import bayesiancoresets.coreset.sparsevi as sparsevi
    import bayesiancoresets.coreset.projector as projector
    import model_linreg, model_neurlinr, model_lr, gaussian
    return types.SimpleNamespace(opt=opt, snnls=snnls, hilbert=hilbert, bcores=bcores, sparsevi=sparsevi,
                                 projector=projector, linreg=model_linreg, neurlinr=model_neurlinr,
                                 lr=model_lr, gaussian=gaussian)


R = import_reference()
META = dict(numpy=np.__version__, scipy=scipy.__version__)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    arrs['meta_numpy'] = np.array(META['numpy'])
    arrs['meta_scipy'] = np.array(META['scipy'])
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024.))


# ---------------------------------------------------------------- F1: SNNLS known-answer
def gendata(rng, N, D, dist):
    """Same five designs as the reference's (stale) tests/test_snnls/test_deterministic.py:18-35."""
    if dist == 'gauss':
        return rng.normal(0., 1., (N, D))
    if dist == 'bin':
        return (rng.rand(N, D) > 0.5).astype(float)
    if dist == 'gauss_colinear':
        x = rng.normal(0., 1., D)
        return (rng.rand(N) * 2. - 1.)[:, None] * x
    if dist == 'bin_colinear':
        x = (rng.rand(D) > 0.5).astype(float)
        return (rng.rand(N) * 2. - 1.)[:, None] * x
    x = np.zeros((N, N))
    x[np.arange(N), np.arange(N)] = 1. / float(N)
    return x


def run_solver_stepwise(alg_cls, X, steps):
    """build(1) x steps, recording the selected index of every consumed iteration."""
    s = alg_cls(X.T, X.sum(axis=0))
    picks = []
    orig = s._select

    def logged():
        f = orig()
        picks.append(int(f))
        return f
    s._select = logged
    sel = np.full(steps, -1, dtype=np.int64)
    err = np.zeros(steps)
    lim = np.zeros(steps, dtype=np.int8)
    W = np.zeros((steps, X.shape[0]))
    for m in range(steps):
        n0 = len(picks)
        s.build(1)
        if len(picks) > n0:
            sel[m] = picks[-1]
        err[m] = s.error()
        lim[m] = s.reached_numeric_limit
        W[m] = s.weights()
    return sel, err, lim, W


def f1_snnls():
    rng = np.random.RandomState(324)    # test_deterministic.py:8
    algs = dict(giga=R.snnls.GIGA, fw=R.snnls.FrankWolfe, omp=R.snnls.OrthoPursuit)
    out = {}
    cases = []
    for N in (10, 100):
        for D in (3, 10):
            for dist in ('gauss', 'bin', 'gauss_colinear', 'bin_colinear', 'axis_aligned'):
                X = gendata(rng, N, D, dist)
                if np.any(np.sqrt((X ** 2).sum(axis=1)) == 0):
                    continue   # zero rows -> ValueError in the ctor (giga.py:11-12); covered in tests directly
                tag = '%s_N%d_D%d' % (dist, N, D)
                out[tag + '_X'] = X
                steps = min(N, 25)
                for an, cls in algs.items():
                    sel, err, lim, W = run_solver_stepwise(cls, X, steps)
                    out['%s_%s_sel' % (tag, an)] = sel
                    out['%s_%s_err' % (tag, an)] = err
                    out['%s_%s_lim' % (tag, an)] = lim
                    out['%s_%s_W' % (tag, an)] = W
                cases.append(tag)
    out['cases'] = np.array(cases)
    save('f1_snnls', **out)


# ---------------------------------------------------------------- F2: likelihood formulas
def f2_formulas():
    rng = np.random.RandomState(2)
    out = {}
    # linreg / beta-linreg
    N, D, S = 64, 12, 24
    X = rng.randn(N, D)
    thstar = rng.randn(D)
    y = X.dot(thstar) + rng.randn(N)
    y[:6] = rng.normal(10., .5, 6)                       # outliers (model_neurlinr.py:63)
    Z = np.hstack((X, y[:, None]))
    th = thstar + 0.3 * rng.randn(S, D)
    out['lin_Z'], out['lin_th'] = Z, th
    for sig in (1.0, 2.5):
        out['lin_ll_sig%g' % sig] = R.linreg.gaussian_loglikelihood(Z, th, sig)
        assert np.array_equal(out['lin_ll_sig%g' % sig], R.neurlinr.neurlinr_loglikelihood(Z, th, sig))
        for beta in (0.1, 0.2, 0.5):
            out['lin_bl_sig%g_b%g' % (sig, beta)] = R.neurlinr.neurlinr_beta_likelihood(Z, th, beta, sig)
    # logistic / beta-logistic with forced margins |m| in {0.5, 120, 800}
    D = 16
    Xl = rng.randn(N, D)
    thl = rng.randn(S, D) / np.sqrt(D)
    yl = np.where(rng.rand(N) < 0.5, 1., -1.)
    Zl = yl[:, None] * Xl
    for r, mag in zip(range(6), (0.5, -0.5, 120., -120., 800., -800.)):
        Zl[r] = -mag * thl[0] / (thl[0] ** 2).sum()      # m = -z.th0 = mag for sample 0
    out['log_Z'], out['log_th'] = Zl, thl
    out['log_ll'] = R.lr.log_likelihood(Zl, thl)
    with np.errstate(over='ignore'):
        for beta in (0.1, 0.2, 0.5):
            out['log_bl_b%g' % beta] = R.lr.beta_likelihood(Zl, thl, beta)
    # gaussian location
    d, Sg = 8, 40
    Sig = 500. * np.eye(d)
    A = rng.randn(d, d)
    Sig_full = A.dot(A.T) + d * np.eye(d)
    Xg = rng.multivariate_normal(np.zeros(d), Sig, N)
    thg = rng.randn(Sg, d) * 3.
    for nm, Sg_ in (('iso', Sig), ('full', Sig_full)):
        Siginv = np.linalg.inv(Sg_)
        logdet = np.linalg.slogdet(Sg_)[1]
        out['gau_%s_Siginv' % nm] = Siginv
        out['gau_%s_logdet' % nm] = np.array(logdet)
        out['gau_%s_ll' % nm] = quiet(R.gaussian.gaussian_loglikelihood, Xg, thg, Siginv, logdet)
        for beta in (0.1, 0.5):
            out['gau_%s_bl_b%g' % (nm, beta)] = R.gaussian.gaussian_beta_likelihood(Xg, thg, beta, Siginv, logdet)
            out['gau_%s_bg_b%g' % (nm, beta)] = R.gaussian.gaussian_beta_gradient(Xg, thg, beta, Siginv, logdet)
    out['gau_X'], out['gau_th'] = Xg, thg
    save('f2_formulas', **out)


# ---------------------------------------------------------------- helpers: synthetic models
def linreg_problem(rng, N, D, S, outlier_frac=0.1):
    X = rng.randn(N, D)
    thstar = rng.randn(D)
    y = X.dot(thstar) + rng.randn(N)
    no = int(outlier_frac * N)
    if no:
        y[rng.choice(N, no, replace=False)] = rng.normal(10., .5, no)
    Z = np.hstack((X, y[:, None]))
    mu, L, Linv = R.linreg.weighted_post(np.zeros(D), np.eye(D), 1.0, Z, np.ones(N))
    E = rng.randn(S, D)
    th = mu + E.dot(L.T)
    return Z, th, E


def hilbert_run(data, prj, steps, snnls_cls=None):
    kw = dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    if snnls_cls is not None:
        kw['snnls'] = snnls_cls
    h = R.hilbert.HilbertCoreset(data, prj, **kw)
    picks = []
    orig = h.snnls._select

    def logged():
        f = orig()
        picks.append(int(f))
        return f
    h.snnls._select = logged
    sel = np.full(steps, -1, dtype=np.int64)
    err = np.zeros(steps)
    for m in range(steps):
        n0 = len(picks)
        h.build(1, m + 1)
        if len(picks) > n0:
            sel[m] = picks[-1]
        err[m] = h.error()
    wts, pts, idcs = h.get()
    w_final = h.snnls.weights()
    h.optimize()
    wo, po, io_ = h.get()
    return dict(sel=sel, err=err, wts=wts, idcs=idcs, w_dense=w_final, opt_wts=wo, opt_idcs=io_,
                opt_err=np.array(h.error()), b=h.snnls.b.copy())


def f3_hilbert_linreg():
    rng = np.random.RandomState(3)
    N, D, S, steps = 1000, 8, 50, 50
    Z, th, E = linreg_problem(rng, N, D, S)
    out = dict(Z=Z, th=th)
    for nm, ll in (('ll', lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0)),
                   ('bl', lambda z, t: R.neurlinr.neurlinr_beta_likelihood(z, t, 0.1, 1.0))):
        prj = R.projector.BlackBoxProjector(lambda n, w, p: th, S, ll)
        phi = prj.project(Z)
        out['phi_' + nm] = phi
        for an, cls in (('giga', R.snnls.GIGA), ('fw', R.snnls.FrankWolfe), ('omp', R.snnls.OrthoPursuit)):
            res = hilbert_run(Z, prj, steps if an != 'omp' else 20, cls)
            for k, v in res.items():
                out['%s_%s_%s' % (nm, an, k)] = v
    save('f3_hilbert_linreg', **out)


def f4_hilbert_logistic_gauss():
    rng = np.random.RandomState(4)
    out = {}
    # logistic, D=16
    N, D, S, steps = 1000, 16, 50, 40
    X = rng.randn(N, D)
    thstar = np.ones(D) / np.sqrt(D)
    yl = np.where(rng.rand(N) <= 1. / (1. + np.exp(-X.dot(thstar))), 1., -1.)
    Z = yl[:, None] * X
    th = thstar + 0.1 * rng.randn(S, D)
    out['log_Z'], out['log_th'] = Z, th
    with np.errstate(over='ignore'):
        for nm, ll in (('ll', R.lr.log_likelihood), ('bl', lambda z, t: R.lr.beta_likelihood(z, t, 0.1))):
            prj = R.projector.BlackBoxProjector(lambda n, w, p: th, S, ll)
            res = hilbert_run(Z, prj, steps)
            for k, v in res.items():
                out['log_%s_%s' % (nm, k)] = v
    # gaussian location, config-1 plumbing shrunk: d=8, S=200, three outlier clusters (zellner_gaussian/main.py:33-54)
    N, d, S = 1000, 8, 200
    Sig = 500. * np.eye(d)
    Siginv = np.linalg.inv(Sig)
    logdet = np.linalg.slogdet(Sig)[1]
    X = rng.multivariate_normal(np.zeros(d), Sig, N)
    mup, LSigp, _ = R.gaussian.weighted_post(np.zeros(d), np.eye(d), Siginv, X, np.ones(N))
    Xc = np.concatenate((X,
                         rng.multivariate_normal(np.zeros(d) + 200, 0.5 * Sig, N // 50),
                         rng.multivariate_normal(np.zeros(d) + 150, 0.1 * Sig, N // 50),
                         rng.multivariate_normal(np.zeros(d), 10 * Sig, N // 10)))
    thg = mup + rng.randn(S, d).dot(LSigp.T)
    out['gau_X'], out['gau_th'], out['gau_Siginv'], out['gau_logdet'] = Xc, thg, Siginv, np.array(logdet)
    prj = R.projector.BlackBoxProjector(lambda n, w, p: thg, S,
                                        lambda x, t: quiet(R.gaussian.gaussian_loglikelihood, x, t, Siginv, logdet))
    res = hilbert_run(Xc, prj, steps)
    for k, v in res.items():
        out['gau_ll_%s' % k] = v
    save('f4_hilbert_logistic_gauss', **out)


# ---------------------------------------------------------------- F5: BetaCoreset / SparseVI
def f5_greedy_vi():
    rng = np.random.RandomState(5)
    N, D, S = 400, 6, 30
    Z, _, E = linreg_problem(rng, N, D, S)
    out = dict(Z=Z, E=E)
    opt_itrs, builds = 10, 5

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, Z.shape[1]))
        mu, L, _ = R.linreg.weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)

    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    beta = 0.1
    bl = lambda z, t, b: R.neurlinr.neurlinr_beta_likelihood(z, t, b, 1.0)
    ll = lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0)
    prj_b = R.projector.BetaBlackBoxProjector(sampler, S, bl, ll, None)
    bco = R.bcores.BetaCoreset(Z, prj_b, opt_itrs=opt_itrs, step_sched=lambda i: 0.1 / (1. + i),
                               beta=beta, learn_beta=False, **fresh())
    prj_s = R.projector.BlackBoxProjector(sampler, S, ll)
    svi = R.sparsevi.SparseVICoreset(Z, prj_s, opt_itrs=opt_itrs, step_sched=lambda i: 0.1 / (1. + i), **fresh())
    for nm, alg in (('bcores', bco), ('svi', svi)):
        for m in range(builds):
            quiet(alg.build, 1, m + 1)
            got = alg.get()
            out['%s_wts_%d' % (nm, m)] = got[0].copy()
            out['%s_idcs_%d' % (nm, m)] = got[2].copy()
            out['%s_allw_%d' % (nm, m)] = alg.wts.copy()
            out['%s_allidcs_%d' % (nm, m)] = alg.idcs.copy()
    out['beta'] = np.array(beta)
    out['opt_itrs'] = np.array(opt_itrs)
    save('f5_greedy_vi', **out)


def f8_grouped_vi():
    """Grouped (batch) selection, bcores.py:46-50,91-123 / sparsevi.py:43-47,93-126, full-data mode."""
    rng = np.random.RandomState(8)
    N, D, S = 240, 5, 24
    Z, _, E = linreg_problem(rng, N, D, S)
    perm = rng.permutation(N)
    groups = [sorted(perm[i:i + 12].tolist()) for i in range(0, N, 12)]     # 20 groups of 12 rows
    out = dict(Z=Z, E=E, groups=np.array(groups))
    opt_itrs, builds = 6, 4

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, Z.shape[1]))
        mu, L, _ = R.linreg.weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)

    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    bl = lambda z, t, b: R.neurlinr.neurlinr_beta_likelihood(z, t, b, 1.0)
    ll = lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0)
    bco = R.bcores.BetaCoreset(Z, R.projector.BetaBlackBoxProjector(sampler, S, bl, ll, None), opt_itrs=opt_itrs,
                               step_sched=lambda i: 0.1 / (1. + i), beta=0.1, learn_beta=False, groups=groups, **fresh())
    svi = R.sparsevi.SparseVICoreset(Z, R.projector.BlackBoxProjector(sampler, S, ll), opt_itrs=opt_itrs,
                                     step_sched=lambda i: 0.1 / (1. + i), groups=groups, **fresh())
    for nm, alg in (('bcores', bco), ('svi', svi)):
        for m in range(builds):
            quiet(alg.build, 1, 12 * (m + 1))
            out['%s_allw_%d' % (nm, m)] = alg.wts.copy()
            out['%s_allidcs_%d' % (nm, m)] = alg.idcs.copy()
            out['%s_groups_%d' % (nm, m)] = np.array([int(g) for g in alg.selected_groups])
    out['opt_itrs'] = np.array(opt_itrs)
    save('f8_grouped_vi', **out)


def f9_subsampled_gaussian():
    """The zellner_gaussian recipe shrunk (main.py:33-105): Gaussian location model, sub-sampled
    selection / optimisation, samplers drawing from the GLOBAL NumPy RNG -- pins the RNG call order
    (sampler before randint, bcores.py:39 then :53)."""
    np.random.seed(9)
    N, d, S = 600, 6, 40
    Sig = 500. * np.eye(d)
    Siginv = np.linalg.inv(Sig)
    logdet = np.linalg.slogdet(Sig)[1]
    X = np.random.multivariate_normal(np.zeros(d), Sig, N)
    Xc = np.concatenate((X, np.random.multivariate_normal(np.zeros(d) + 200, 0.5 * Sig, N // 50),
                         np.random.multivariate_normal(np.zeros(d), 10 * Sig, N // 10)))
    mu0, Sig0inv = np.zeros(d), np.eye(d)

    def sampler_w(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, Xc.shape[1]))
        muw, LSigw, _ = R.gaussian.weighted_post(mu0, Sig0inv, Siginv, pts, wts)
        return muw + np.random.randn(sz, muw.shape[0]).dot(LSigw.T)

    ll = lambda x, th: quiet(R.gaussian.gaussian_loglikelihood, x, th, Siginv, logdet)
    bl = lambda x, th, beta: R.gaussian.gaussian_beta_likelihood(x, th, beta, Siginv, logdet)
    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    out = dict(X=Xc, Siginv=Siginv, logdet=np.array(logdet))
    for nm in ('bcores', 'svi'):
        np.random.seed(90)
        if nm == 'bcores':
            prj = R.projector.BetaBlackBoxProjector(sampler_w, S, bl, ll, None)
            alg = R.bcores.BetaCoreset(Xc, prj, opt_itrs=8, n_subsample_opt=60, n_subsample_select=150,
                                       step_sched=lambda i: 0.1 / (1. + i), beta=.1, learn_beta=False, **fresh())
        else:
            prj = R.projector.BlackBoxProjector(sampler_w, S, ll)
            alg = R.sparsevi.SparseVICoreset(Xc, prj, opt_itrs=8, n_subsample_opt=60, n_subsample_select=150,
                                             step_sched=lambda i: 0.1 / (1. + i), **fresh())
        for m in range(6):
            quiet(alg.build, 1, m + 1)
            out['%s_allw_%d' % (nm, m)] = alg.wts.copy()
            out['%s_allidcs_%d' % (nm, m)] = alg.idcs.copy()
        out['%s_rng_after' % nm] = np.array(np.random.rand())       # the RNG stream position must match too
    save('f9_subsampled_gaussian', **out)


# ---------------------------------------------------------------- F6/F7
def f6_weighted_post():
    rng = np.random.RandomState(6)
    out = {}
    for D in (8, 64):
        N = 300
        X = rng.randn(N, D) * (1. + np.arange(D) / D)          # anisotropic on purpose
        y = X.dot(rng.randn(D)) + rng.randn(N)
        Z = np.hstack((X, y[:, None]))
        w = rng.rand(N) * 2.
        th0 = rng.randn(D) * .1
        Sig0inv = np.eye(D) * 0.5
        mu, L, Linv = R.linreg.weighted_post(th0, Sig0inv, 1.7, Z, w)
        out['D%d_Z' % D], out['D%d_w' % D], out['D%d_th0' % D], out['D%d_Sig0inv' % D] = Z, w, th0, Sig0inv
        out['D%d_mu' % D], out['D%d_L' % D], out['D%d_Linv' % D] = mu, L, Linv
    d = 8
    Siginv = np.linalg.inv(500. * np.eye(d))
    Xg = rng.randn(200, d) * 20
    wg = rng.rand(200)
    mu, L, Linv = R.gaussian.weighted_post(np.zeros(d), np.eye(d), Siginv, Xg, wg)
    out['g_X'], out['g_w'], out['g_Siginv'], out['g_mu'], out['g_L'], out['g_Linv'] = Xg, wg, Siginv, mu, L, Linv
    save('f6_weighted_post', **out)


def f7_nn_opt():
    rng = np.random.RandomState(7)
    n = 12
    Q = rng.randn(n, n)
    Q = Q.dot(Q.T) + np.eye(n)
    c = rng.randn(n) * 3
    x0 = np.abs(rng.randn(n))
    grd = lambda x: Q.dot(x) - c
    out = dict(Q=Q, c=c, x0=x0)
    out['nn'] = R.opt.nn_opt(x0, grd, opt_itrs=50, step_sched=lambda i: 0.5 / (1. + i))
    out['pnn'] = R.opt.partial_nn_opt(x0, grd, np.arange(0, n, 2), opt_itrs=50, step_sched=lambda i: 0.5 / (1. + i))
    save('f7_nn_opt', **out)


if __name__ == '__main__':
    f1_snnls()
    f2_formulas()
    f3_hilbert_linreg()
    f4_hilbert_logistic_gauss()
    f5_greedy_vi()
    f6_weighted_post()
    f7_nn_opt()
    f8_grouped_vi()
    f9_subsampled_gaussian()
