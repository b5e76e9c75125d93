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
    import bayesiancoresets.coreset.bpsvi as bpsvi
    import bayesiancoresets.coreset.sampling as csampling
    import model_linreg, model_neurlinr, model_lr, gaussian
    return types.SimpleNamespace(opt=opt, snnls=snnls, hilbert=hilbert, bcores=bcores, sparsevi=sparsevi,
                                 projector=projector, bpsvi=bpsvi, csampling=csampling, linreg=model_linreg, neurlinr=model_neurlinr,
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


# ---------------------------------------------------------------- F10: sampling "solvers"
def f10_sampling():
    """ImportanceSampling / UniformSampling (snnls/sampling.py:6-37), draws from the GLOBAL NumPy RNG."""
    rng = np.random.RandomState(10)
    N, D, S, steps = 300, 6, 40, 30
    Z, th, _ = linreg_problem(rng, N, D, S)
    prj = R.projector.BlackBoxProjector(lambda n, w, p: th, S, lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0))
    phi = prj.project(Z)
    out = dict(Z=Z, th=th, phi=phi)
    for nm, cls in (('imp', R.snnls.ImportanceSampling), ('unif', R.snnls.UniformSampling)):
        np.random.seed(100)
        sel, err, lim, W = run_solver_stepwise(cls, phi, steps)
        out[nm + '_sel'], out[nm + '_err'], out[nm + '_lim'], out[nm + '_W'] = sel, err, lim, W
        out[nm + '_rng_after'] = np.array(np.random.rand())
        # the same through HilbertCoreset(snnls=...), one build(steps) call
        np.random.seed(101)
        h = R.hilbert.HilbertCoreset(Z, prj, snnls=cls, wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
        h.build(steps, steps)
        wts, pts, idcs = h.get()
        out[nm + '_h_wts'], out[nm + '_h_idcs'], out[nm + '_h_err'] = wts, idcs, np.array(h.error())
    save('f10_sampling', **out)


# ---------------------------------------------------------------- F11: grouped AND sub-sampled tangent space
def f11_grouped_subsampled():
    """bcores.py:56-61 / sparsevi.py:54-59: `groups` together with n_subsample_select (random groups in the
    selection step) and n_subsample_opt (random rows in the gradient steps, bcores.py:51-55).  The sampler is
    deterministic (fixed E); the sub-sampling draws come from the global RNG and their order is pinned."""
    rng = np.random.RandomState(11)
    N, D, S = 240, 5, 24
    Z, _, E = linreg_problem(rng, N, D, S)
    perm = rng.permutation(N)
    groups = [sorted(perm[i:i + 12].tolist()) for i in range(0, N, 12)]     # 20 groups of 12 rows
    out = dict(Z=Z, E=E, groups=np.array(groups))
    opt_itrs, builds = 6, 5

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, Z.shape[1]))
        mu, L, _ = R.linreg.weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)

    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    bl = lambda z, t, b: R.neurlinr.neurlinr_beta_likelihood(z, t, b, 1.0)
    ll = lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0)
    for nm in ('bcores', 'svi'):
        np.random.seed(110)
        if nm == 'bcores':
            alg = R.bcores.BetaCoreset(Z, R.projector.BetaBlackBoxProjector(sampler, S, bl, ll, None), opt_itrs=opt_itrs,
                                       n_subsample_select=8, n_subsample_opt=50, step_sched=lambda i: 0.1 / (1. + i),
                                       beta=0.1, learn_beta=False, groups=groups, **fresh())
        else:
            alg = R.sparsevi.SparseVICoreset(Z, R.projector.BlackBoxProjector(sampler, S, ll), opt_itrs=opt_itrs,
                                             n_subsample_select=8, n_subsample_opt=50, step_sched=lambda i: 0.1 / (1. + i),
                                             groups=groups, **fresh())
        for m in range(builds):
            quiet(alg.build, 1, 12 * (m + 1))
            out['%s_allw_%d' % (nm, m)] = alg.wts.copy()
            out['%s_allidcs_%d' % (nm, m)] = alg.idcs.copy()
            out['%s_groups_%d' % (nm, m)] = np.array([int(g) for g in alg.selected_groups])
        out['%s_rng_after' % nm] = np.array(np.random.rand())
    out['opt_itrs'] = np.array(opt_itrs)
    save('f11_grouped_subsampled', **out)


# ---------------------------------------------------------------- F12: constant rows (all-zero features)
def f12_constant_rows():
    """Data rows with all-zero features project to S equal numbers c; `lls -= lls.mean(axis=1)` (projector.py:26)
    leaves c - mean, which NumPy's pairwise mean makes non-zero for most (c, S) with S not a power of two.  Such
    rows keep a positive norm, survive hilbert.py:16 and no index shift happens; rows whose residue IS exactly zero
    are dropped and shift every later index (hilbert.py:16 vs :32).  S = 100 and S = 200, linreg (x_i = 0, assorted y)
    and logistic (z_i = 0: every value is -log 2)."""
    rng = np.random.RandomState(12)
    out = {}
    N, D, steps = 600, 8, 30
    zero_at = np.array([0, 3, 4, 17, 100, 101, 257, 300, 311, 389, 450, 512, 555, 580, 599])
    for S in (100, 200):
        Z, th, _ = linreg_problem(rng, N, D, S)
        Z[zero_at, :D] = 0.
        Z[zero_at[:4], D] = [0., 1., -2., 0.5]                       # exactly representable y among the rest
        prj = R.projector.BlackBoxProjector(lambda n, w, p: th, S, lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0))
        phi = prj.project(Z)
        tag = 'lin_S%d_' % S
        out[tag + 'Z'], out[tag + 'th'], out[tag + 'phi_const'] = Z, th, phi[zero_at]      # the constant rows of Phi
        out[tag + 'kept'] = np.sqrt((phi ** 2).sum(axis=1)) > 0.         # hilbert.py:16
        for an, cls in (('giga', R.snnls.GIGA), ('fw', R.snnls.FrankWolfe)):
            res = hilbert_run(Z, prj, steps, cls)
            for k in ('sel', 'err', 'wts', 'idcs'):
                out['%s%s_%s' % (tag, an, k)] = res[k]
        # logistic
        Dl = 16
        X = rng.randn(N, Dl)
        thstar = np.ones(Dl) / np.sqrt(Dl)
        yl = np.where(rng.rand(N) <= 1. / (1. + np.exp(-X.dot(thstar))), 1., -1.)
        Zl = yl[:, None] * X
        Zl[zero_at] = 0.
        thl = thstar + 0.1 * rng.randn(S, Dl)
        prjl = R.projector.BlackBoxProjector(lambda n, w, p: thl, S, R.lr.log_likelihood)
        phil = prjl.project(Zl)
        tag = 'log_S%d_' % S
        out[tag + 'Z'], out[tag + 'th'], out[tag + 'phi_const'] = Zl, thl, phil[zero_at]
        out[tag + 'kept'] = np.sqrt((phil ** 2).sum(axis=1)) > 0.
        res = hilbert_run(Zl, prjl, steps)
        for k in ('sel', 'err', 'wts', 'idcs'):
            out['%sgiga_%s' % (tag, k)] = res[k]
    out['zero_at'] = zero_at
    save('f12_constant_rows', **out)


# ---------------------------------------------------------------- F13: greedy VI with all-zero projection rows
def f13_greedy_vi_zero_rows():
    """BetaCoreset / SparseVI never drop all-zero rows of the tangent space: the filter at bcores.py:67-68 needs
    select=True with groups=None, and the ungrouped _select calls _get_projection with select=False (bcores.py:76).
    A zero row's correlation is 0/0 = NaN; np.argmax returns the first NaN, `corrs.max() > x` is False.  S = 16
    makes the constant rows exactly zero (the mean of 16 equal doubles is exact); S = 100 keeps them as tiny
    residue rows.  Both recorded."""
    rng = np.random.RandomState(13)
    N, D = 300, 5
    out = {}
    for S in (16, 100):
        Z, _, E = linreg_problem(rng, N, D, S)
        zero_at = np.array([7, 40, 41, 200])
        Z[zero_at, :D] = 0.
        out['S%d_Z' % S], out['S%d_E' % S] = Z, E

        def sampler(sz, wts, pts, Z=Z, E=E):
            if pts.shape[0] == 0:
                wts = np.zeros(1)
                pts = np.zeros((1, Z.shape[1]))
            mu, L, _ = R.linreg.weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
            return mu + E.dot(L.T)

        fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
        bl = lambda z, t, b: R.neurlinr.neurlinr_beta_likelihood(z, t, b, 1.0)
        ll = lambda z, t: R.linreg.gaussian_loglikelihood(z, t, 1.0)
        with np.errstate(invalid='ignore', divide='ignore'):
            for nm in ('bcores', 'svi'):
                if nm == 'bcores':
                    alg = R.bcores.BetaCoreset(Z, R.projector.BetaBlackBoxProjector(sampler, S, bl, ll, None), opt_itrs=5,
                                               step_sched=lambda i: 0.1 / (1. + i), beta=0.1, learn_beta=False, **fresh())
                else:
                    alg = R.sparsevi.SparseVICoreset(Z, R.projector.BlackBoxProjector(sampler, S, ll), opt_itrs=5,
                                                     step_sched=lambda i: 0.1 / (1. + i), **fresh())
                for m in range(4):
                    quiet(alg.build, 1, m + 1)
                    out['S%d_%s_allw_%d' % (S, nm, m)] = alg.wts.copy()
                    out['S%d_%s_allidcs_%d' % (S, nm, m)] = alg.idcs.copy()
    out['zero_at'] = zero_at
    save('f13_greedy_vi_zero_rows', **out)


# ---------------------------------------------------------------- F14: the zellner_gaussian driver, shrunk
def f14_zellner_gaussian_driver():
    """examples/zellner_gaussian/main.py:33-167 at N = 400, d = 6, M = 8, proj_dim = 40: the data recipe, the four
    projector constructors (each draws its first Theta from the global RNG when constructed, projector.py:18,46), the
    BCORES / SVI / GIGAO / GIGAR objects, the build(1, m) loop and the KL metrics -- statement order as in the script,
    so the global RNG stream is the script's.  BPSVI / RAND construct without touching the RNG and are left out."""
    G = R.gaussian
    out = {}
    M, opt_itrs, n_subsample_opt, n_subsample_select, proj_dim, pihat_noise, i0 = 8, 30, 50, 150, 40, 0.75, 0.1
    N, d = 400, 6
    for nm in ('BCORES', 'SVI', 'GIGAO', 'GIGAR'):
        tr = 3
        np.random.seed(tr)
        mu0, Sig0 = np.zeros(d), np.eye(d)
        Sig = 500 * np.eye(d)
        th = np.zeros(d)
        Sig0inv, Siginv = np.linalg.inv(Sig0), np.linalg.inv(Sig)
        logdetSig = np.linalg.slogdet(Sig)[1]
        X = np.random.multivariate_normal(th, Sig, N)
        mup, LSigp, LSigpInv = G.weighted_post(mu0, Sig0inv, Siginv, X, np.ones(X.shape[0]))
        Sigp = LSigp.dot(LSigp.T)
        SigpInv = LSigpInv.dot(LSigpInv.T)
        Xc = np.concatenate((X, np.random.multivariate_normal(th + 200, 0.5 * Sig, int(N / 50.)),
                             np.random.multivariate_normal(th + 150, 0.1 * Sig, int(N / 50.)),
                             np.random.multivariate_normal(th, 10 * Sig, int(N / 10.))))
        log_likelihood = lambda x, t: quiet(G.gaussian_loglikelihood, x, t, Siginv, logdetSig)
        beta_likelihood = lambda x, t, beta: G.gaussian_beta_likelihood(x, t, beta, Siginv, logdetSig)
        grad_beta = lambda x, t, beta: G.gaussian_beta_gradient(x, t, beta, Siginv, logdetSig)
        sampler_optimal = lambda n, w, pts: mup + np.random.randn(n, mup.shape[0]).dot(LSigp.T)
        prj_optimal = R.projector.BlackBoxProjector(sampler_optimal, proj_dim, log_likelihood, None)
        U = np.random.rand()
        muhat = U * mup + (1. - U) * mu0
        Sighat = U * Sigp + (1. - U) * Sig0
        muhat += pihat_noise * np.sqrt((muhat ** 2).sum()) * np.random.randn(muhat.shape[0])
        Sighat *= np.exp(-2 * pihat_noise * np.fabs(np.random.randn()))
        LSighat = np.linalg.cholesky(Sighat)
        sampler_realistic = lambda n, w, pts: mup + np.random.randn(n, mup.shape[0]).dot(LSighat.T)
        prj_realistic = R.projector.BlackBoxProjector(sampler_realistic, proj_dim, log_likelihood, None)

        def sampler_w(sz, wts, pts):
            if pts.shape[0] == 0:
                wts = np.zeros(1)
                pts = np.zeros((1, Xc.shape[1]))
            muw, LSigw, _ = G.weighted_post(mu0, Sig0inv, Siginv, pts, wts)
            return muw + np.random.randn(sz, muw.shape[0]).dot(LSigw.T)

        prj_w = R.projector.BlackBoxProjector(sampler_w, proj_dim, log_likelihood, None)
        prj_bw = R.projector.BetaBlackBoxProjector(sampler_w, proj_dim, beta_likelihood, log_likelihood, grad_beta)
        fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
        sched = lambda i: i0 / (1. + i)
        if nm == 'SVI':
            alg = R.sparsevi.SparseVICoreset(Xc, prj_w, opt_itrs=opt_itrs, n_subsample_opt=n_subsample_opt,
                                             n_subsample_select=n_subsample_select, step_sched=sched, **fresh())
        elif nm == 'BCORES':
            alg = R.bcores.BetaCoreset(Xc, prj_bw, opt_itrs=opt_itrs, n_subsample_opt=n_subsample_opt,
                                       n_subsample_select=n_subsample_select, step_sched=sched, beta=.1, learn_beta=False,
                                       **fresh())
        elif nm == 'GIGAO':
            alg = R.hilbert.HilbertCoreset(Xc, prj_optimal, **fresh())
        else:
            alg = R.hilbert.HilbertCoreset(Xc, prj_realistic, **fresh())
        w = [np.array([0.])]
        p = [np.zeros((1, Xc.shape[1]))]
        idl = [np.zeros(0, dtype=np.int64)]
        for m in range(1, M + 1):
            quiet(alg.build, 1, m)
            got = quiet(alg.get)
            w.append(got[0].copy())
            p.append(got[1].copy())
            idl.append(got[2].copy())
        rklw, fklw = np.zeros(M + 1), np.zeros(M + 1)
        for m in range(M + 1):
            muw, LSigw, LSigwInv = G.weighted_post(mu0, Sig0inv, Siginv, p[m], w[m])
            Sigw = LSigw.dot(LSigw.T)
            rklw[m] = G.gaussian_KL(muw, Sigw, mup, SigpInv)
            fklw[m] = G.gaussian_KL(mup, Sigp, muw, LSigwInv.dot(LSigwInv.T))
        out[nm + '_rkl'], out[nm + '_fkl'] = rklw, fklw
        for m in range(M + 1):
            out['%s_w_%d' % (nm, m)] = w[m]
            out['%s_idcs_%d' % (nm, m)] = idl[m]
        out[nm + '_rng_after'] = np.array(np.random.rand())
        if nm == 'GIGAO':
            out['Xc'] = Xc
    out['params'] = np.array([N, d, M, opt_itrs, n_subsample_opt, n_subsample_select, proj_dim, 3])
    save('f14_zellner_gaussian_driver', **out)


# ---------------------------------------------------------------- F15: learn_beta=True
def _get_projection_ii(self, n_subsample, w, p, beta):
    """NOT reference code: the method bcores.py:131 calls does not exist anywhere in the reference tree, so
    learn_beta=True cannot run as shipped.  This is the sibling of _get_projection (bcores.py:37-72) its call
    site implies: same tangent space, plus the beta-gradient of the CORESET rows from
    project_f(..., grad=True) (projector.py:56-61; `betagrads.dot(resid)` at bcores.py:134 must be M-long).
    It is attached to the reference class for this fixture only."""
    got = self._get_projection(n_subsample, w, p, beta)
    vecs, sum_scaling, sub_idcs, corevecs = got
    if self.pts.size > 0:
        corevecs, betagrads = self.ll_projector.project_f(self.pts, beta, grad=True)
    else:
        betagrads = np.zeros((0, vecs.shape[1]))
    return vecs, sum_scaling, sub_idcs, corevecs, betagrads


def f15_learn_beta():
    """BetaCoreset(learn_beta=True), bcores.py:127-140, Gaussian location model (the one model for which the
    reference ships a beta-gradient, gaussian.py:46-62), full-data tangent space, deterministic sampler."""
    rng = np.random.RandomState(15)
    N, d, S = 300, 5, 32
    Sig = 500. * np.eye(d)
    Siginv = np.linalg.inv(Sig)
    logdet = np.linalg.slogdet(Sig)[1]
    X = rng.multivariate_normal(np.zeros(d), Sig, N)
    # outliers near enough that exp(-beta q / 2) does not underflow into constant rows (F13 covers those)
    Xc = np.concatenate((X, rng.multivariate_normal(np.zeros(d) + 40, 0.5 * Sig, N // 50),
                         rng.multivariate_normal(np.zeros(d), 4 * Sig, N // 10)))
    E = rng.randn(S, d)
    mu0, Sig0inv = np.zeros(d), np.eye(d)

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, Xc.shape[1]))
        muw, LSigw, _ = R.gaussian.weighted_post(mu0, Sig0inv, Siginv, pts, wts)
        return muw + E.dot(LSigw.T)

    ll = lambda x, t: quiet(R.gaussian.gaussian_loglikelihood, x, t, Siginv, logdet)
    bl = lambda x, t, beta: R.gaussian.gaussian_beta_likelihood(x, t, beta, Siginv, logdet)
    bg = lambda x, t, beta: R.gaussian.gaussian_beta_gradient(x, t, beta, Siginv, logdet)
    out = dict(X=Xc, E=E, Siginv=Siginv, logdet=np.array(logdet))
    R.bcores.BetaCoreset._get_projection_ii = _get_projection_ii
    # A second quirk limits what the reference can run even then: `self.wts = xf[:-1]` (bcores.py:139) is a VIEW, and
    # the next _select that appends a point dies in `self.wts.resize` ("does not own its data", bcores.py:85).  So the
    # path is pinned by (a) one build from scratch and (b) a pre-initialised 6-point coreset: one build (select appends
    # the 7th point to owned arrays, then the beta-learning optimisation) followed by two more _optimize() calls.
    init_idcs = np.sort(rng.choice(Xc.shape[0], 6, replace=False)).astype(np.int64)
    out['init_idcs'] = init_idcs
    try:
        for tag, nsub in (('full', None), ('sub', 80)):
            np.random.seed(150)
            prj = R.projector.BetaBlackBoxProjector(sampler, S, bl, ll, bg)
            mk = lambda **kw: R.bcores.BetaCoreset(Xc, prj, opt_itrs=8, n_subsample_opt=nsub, n_subsample_select=nsub,
                                                   step_sched=lambda i: 0.1 / (1. + i), beta=.3, learn_beta=True, **kw)
            alg = mk(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
            quiet(alg.build, 1, 1)
            out[tag + '_one_allw'], out[tag + '_one_allidcs'], out[tag + '_one_beta'] = alg.wts.copy(), alg.idcs.copy(), np.array(alg.beta)
            alg = mk(wts=np.full(6, Xc.shape[0] / 6.), idcs=init_idcs.copy(), pts=Xc[init_idcs].copy())
            quiet(alg.build, 1, 7)
            for m in range(3):
                if m > 0:
                    quiet(alg._optimize)
                out['%s_init_allw_%d' % (tag, m)] = alg.wts.copy()
                out['%s_init_allidcs_%d' % (tag, m)] = alg.idcs.copy()
                out['%s_init_beta_%d' % (tag, m)] = np.array(alg.beta)
            out['%s_rng_after' % tag] = np.array(np.random.rand())
    finally:
        del R.bcores.BetaCoreset._get_projection_ii
    save('f15_learn_beta', **out)


def f16_bpsvi():
    """BatchPSVICoreset (bpsvi.py:6-65): pseudo-points initialised by np.random.choice, then `opt_itrs` projected-ADAM
    steps on (weights, points) whose gradient needs project(data) [K1 + K2] and project(points, grad=True)
    (projector.py:27-32: the x-gradient tensor, centred over its LAST axis).  Also the three x-gradient formulas
    themselves (model_linreg.py:12-17, model_lr.py:107-114, gaussian.py:17-20) on seeded inputs."""
    rng = np.random.RandomState(16)
    out = {}
    # ---- formulas
    M, S, D = 9, 12, 5
    z = rng.randn(M, D + 1)
    th = rng.randn(S, D)
    out['lin_z'], out['lin_th'] = z, th
    for sg in (1.0, 2.5):
        out['lin_grad_%g' % sg] = R.linreg.gaussian_grad_x_loglikelihood(z, th, sg)
    zl = rng.randn(M, D) * 2.
    zl[0] *= 80.                                   # m = -z.th beyond the branch at 100 on some samples
    out['log_z'] = zl
    out['log_grad'] = R.lr.grad_z_log_likelihood(zl, th)
    A = rng.randn(D, D)
    Sig = A.dot(A.T) + D * np.eye(D)
    Siginv = np.linalg.inv(Sig)
    xg = rng.randn(M, D) * 3.
    out['gau_x'], out['gau_Siginv'] = xg, Siginv
    out['gau_grad'] = R.gaussian.gaussian_grad_x_loglikelihood(xg, th, Siginv)
    # centred as the projector does it
    prj = R.projector.BlackBoxProjector(lambda n, w, p: th, S, lambda a, b: R.linreg.gaussian_loglikelihood(a, b, 2.5),
                                        lambda a, b: R.linreg.gaussian_grad_x_loglikelihood(a, b, 2.5))
    lls, glls = prj.project(z.copy(), grad=True)
    out['lin_proj_lls'], out['lin_proj_glls'] = lls, glls

    # ---- the coreset, Gaussian location model (the zellner recipe shrunk) and linear regression
    N, d, Sp = 240, 4, 24
    Sigm = 50. * np.eye(d)
    Sinv = np.linalg.inv(Sigm)
    logdet = np.linalg.slogdet(Sigm)[1]
    X = np.concatenate((rng.multivariate_normal(np.zeros(d), Sigm, N), rng.multivariate_normal(np.zeros(d) + 15., 0.5 * Sigm, N // 10)))
    mu0, Sig0inv = np.zeros(d), np.eye(d)
    Eg = rng.randn(Sp, d)

    def sampler_g(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, d))
        muw, LSigw, _ = R.gaussian.weighted_post(mu0, Sig0inv, Sinv, pts, wts)
        return muw + Eg.dot(LSigw.T)

    llg = lambda x, t: quiet(R.gaussian.gaussian_loglikelihood, x, t, Sinv, logdet)
    glg = lambda x, t: R.gaussian.gaussian_grad_x_loglikelihood(x, t, Sinv)
    out.update(g_X=X, g_E=Eg, g_Siginv=Sinv, g_logdet=np.array(logdet))
    Dl = 5
    Xl = rng.randn(300, Dl)
    yl = Xl.dot(rng.randn(Dl)) + 0.7 * rng.randn(300)
    Zl = np.hstack((Xl, yl[:, None]))
    El = rng.randn(Sp, Dl)
    sigsq = 1.3

    def sampler_l(sz, wts, pts):
        if pts.shape[0] == 0:
            wts = np.zeros(1)
            pts = np.zeros((1, Dl + 1))
        muw, LSigw, _ = R.linreg.weighted_post(np.zeros(Dl), np.eye(Dl), sigsq, pts, wts)
        return muw + El.dot(LSigw.T)

    lll = lambda a, b: R.linreg.gaussian_loglikelihood(a, b, sigsq)
    gll = lambda a, b: R.linreg.gaussian_grad_x_loglikelihood(a, b, sigsq)
    out.update(l_Z=Zl, l_E=El, l_sigsq=np.array(sigsq))
    for tag, data, smp, ll, gl in (('g', X, sampler_g, llg, glg), ('l', Zl, sampler_l, lll, gll)):
        for mode, nsub in (('full', None), ('sub', 60)):
            np.random.seed(160)
            prj = R.projector.BlackBoxProjector(smp, Sp, ll, gl)
            alg = R.bpsvi.BatchPSVICoreset(data, prj, opt_itrs=6, n_subsample_opt=nsub,
                                           step_sched=lambda m: lambda i: 0.5 / (1. + i))
            for sz in (3, 5):
                quiet(alg.build, 1, sz)
                k = '%s_%s_%d' % (tag, mode, sz)
                out[k + '_wts'], out[k + '_pts'], out[k + '_idcs'] = alg.wts.copy(), alg.pts.copy(), alg.idcs.copy()
            out['%s_%s_rng_after' % (tag, mode)] = np.array(np.random.rand())
    save('f16_bpsvi', **out)


def f17_uniform_sampling_coreset():
    """UniformSamplingCoreset (coreset/sampling.py:5-52), the RAND baseline of the drivers: rows, or whole groups, drawn
    from the global RNG; build(1, m) for growing m as main.py:140-151 calls it, plus a coreset handed initial points."""
    rng = np.random.RandomState(17)
    N, d = 60, 3
    X = rng.randn(N, d)
    out = dict(X=X)
    np.random.seed(170)
    alg = R.csampling.UniformSamplingCoreset(X, wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    alg.cts, alg.ct_idcs = [], []          # 'wts' in kw makes the constructor take the "initialised" branch: same empty lists
    for m in range(1, 13):
        alg.build(1, m)
        w, p, i = alg.get()
        out['rows_w_%d' % m], out['rows_p_%d' % m], out['rows_i_%d' % m] = w.copy(), p.copy(), i.copy()
    out['rows_rng_after'] = np.array(np.random.rand())
    np.random.seed(171)
    alg = R.csampling.UniformSamplingCoreset(X, wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    alg.build(9, 9)                        # several draws in one call, repeats counted
    out['rows9_w'], out['rows9_i'] = alg.wts.copy(), alg.idcs.copy()
    init = np.array([5, 17, 40])
    np.random.seed(172)
    alg = R.csampling.UniformSamplingCoreset(X, wts=np.full(3, N / 3.), idcs=init.copy(), pts=X[init].copy())
    alg.build(4, 7)
    out['init_idcs'] = init
    out['init_w'], out['init_i'], out['init_p'] = alg.wts.copy(), alg.idcs.copy(), alg.pts.copy()
    groups = [list(range(g * 5, g * 5 + 5)) for g in range(N // 5)]
    out['groups'] = np.array(groups)
    np.random.seed(173)
    alg = R.csampling.UniformSamplingCoreset(X, groups=groups, wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    for t in range(4):
        alg.build(1, 5 * (t + 1))
        out['grp_w_%d' % t], out['grp_i_%d' % t], out['grp_p_%d' % t] = alg.wts.copy(), alg.idcs.copy(), alg.pts.copy()
    out['grp_sel'] = np.array(alg.selected_groups)
    save('f17_uniform_sampling_coreset', **out)


def f18_zellner_gaussian_bpsvi_rand():
    """The rest of the driver's algorithm table (main.py:101-113): BPSVI, RAND, PRIOR on the recipe of F14.  The script
    builds BPSVI for m = 1..M in a multiprocessing pool (main.py:126-135): every build runs in a forked child that starts
    from the parent's RNG state, and which child gets which m is up to the scheduler.  Pinned here in the one
    deterministic reading of that: every m is built from the parent's state at the fork (state restored per m)."""
    G = R.gaussian
    out = {}
    M, opt_itrs, n_subsample_opt, proj_dim, pihat_noise, i0 = 6, 25, 50, 40, 0.75, 0.1
    N, d = 400, 6
    for nm in ('BPSVI', 'RAND', 'PRIOR'):
        tr = 3
        np.random.seed(tr)
        mu0, Sig0 = np.zeros(d), np.eye(d)
        Sig = 500 * np.eye(d)
        th = np.zeros(d)
        Sig0inv, Siginv = np.linalg.inv(Sig0), np.linalg.inv(Sig)
        logdetSig = np.linalg.slogdet(Sig)[1]
        X = np.random.multivariate_normal(th, Sig, N)
        mup, LSigp, LSigpInv = G.weighted_post(mu0, Sig0inv, Siginv, X, np.ones(X.shape[0]))
        Sigp = LSigp.dot(LSigp.T)
        SigpInv = LSigpInv.dot(LSigpInv.T)
        Xc = np.concatenate((X, np.random.multivariate_normal(th + 200, 0.5 * Sig, int(N / 50.)),
                             np.random.multivariate_normal(th + 150, 0.1 * Sig, int(N / 50.)),
                             np.random.multivariate_normal(th, 10 * Sig, int(N / 10.))))
        log_likelihood = lambda x, t: quiet(G.gaussian_loglikelihood, x, t, Siginv, logdetSig)
        grad_log_likelihood = lambda x, t: G.gaussian_grad_x_loglikelihood(x, t, Siginv)
        beta_likelihood = lambda x, t, beta: G.gaussian_beta_likelihood(x, t, beta, Siginv, logdetSig)
        grad_beta = lambda x, t, beta: G.gaussian_beta_gradient(x, t, beta, Siginv, logdetSig)
        sampler_optimal = lambda n, w, pts: mup + np.random.randn(n, mup.shape[0]).dot(LSigp.T)
        prj_optimal = R.projector.BlackBoxProjector(sampler_optimal, proj_dim, log_likelihood, grad_log_likelihood)
        U = np.random.rand()
        muhat = U * mup + (1. - U) * mu0
        Sighat = U * Sigp + (1. - U) * Sig0
        muhat += pihat_noise * np.sqrt((muhat ** 2).sum()) * np.random.randn(muhat.shape[0])
        Sighat *= np.exp(-2 * pihat_noise * np.fabs(np.random.randn()))
        LSighat = np.linalg.cholesky(Sighat)
        sampler_realistic = lambda n, w, pts: mup + np.random.randn(n, mup.shape[0]).dot(LSighat.T)
        prj_realistic = R.projector.BlackBoxProjector(sampler_realistic, proj_dim, log_likelihood, grad_log_likelihood)

        def sampler_w(sz, wts, pts):
            if pts.shape[0] == 0:
                wts = np.zeros(1)
                pts = np.zeros((1, Xc.shape[1]))
            muw, LSigw, _ = G.weighted_post(mu0, Sig0inv, Siginv, pts, wts)
            return muw + np.random.randn(sz, muw.shape[0]).dot(LSigw.T)

        prj_w = R.projector.BlackBoxProjector(sampler_w, proj_dim, log_likelihood, grad_log_likelihood)
        prj_bw = R.projector.BetaBlackBoxProjector(sampler_w, proj_dim, beta_likelihood, log_likelihood, grad_beta)
        fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
        w = [np.array([0.])]
        p = [np.zeros((1, Xc.shape[1]))]
        if nm == 'BPSVI':
            alg = R.bpsvi.BatchPSVICoreset(Xc, prj_w, opt_itrs=opt_itrs, n_subsample_opt=n_subsample_opt,
                                           step_sched=lambda m: lambda i: i0 / (1. + i), **fresh())
            state = np.random.get_state()
            for m in range(1, M + 1):
                np.random.set_state(state)              # a forked child starts from the parent's stream
                quiet(alg.build, 1, m)
                got = alg.get()
                w.append(got[0].copy())
                p.append(got[1].copy())
                out['BPSVI_idcs_%d' % m] = got[2].copy()
            np.random.set_state(state)                  # ... and the parent's own stream never moved
        elif nm == 'RAND':
            alg = R.csampling.UniformSamplingCoreset(Xc, **fresh())
            for m in range(1, M + 1):
                alg.build(1, m)
                got = alg.get()
                w.append(got[0].copy())
                p.append(got[1].copy())
                out['RAND_idcs_%d' % m] = got[2].copy()
        else:
            for m in range(1, M + 1):
                w.append(np.array([0.]))
                p.append(np.zeros((1, Xc.shape[1])))    # (main.py:151 writes Y.shape[0], an undefined name: PRIOR cannot run as shipped)
        rklw, fklw = np.zeros(M + 1), np.zeros(M + 1)
        for m in range(M + 1):
            muw, LSigw, LSigwInv = G.weighted_post(mu0, Sig0inv, Siginv, p[m], w[m])
            Sigw = LSigw.dot(LSigw.T)
            rklw[m] = G.gaussian_KL(muw, Sigw, mup, SigpInv)
            fklw[m] = G.gaussian_KL(mup, Sigp, muw, LSigwInv.dot(LSigwInv.T))
        out[nm + '_rkl'], out[nm + '_fkl'] = rklw, fklw
        for m in range(M + 1):
            out['%s_w_%d' % (nm, m)] = w[m]
            out['%s_p_%d' % (nm, m)] = p[m]
        out[nm + '_rng_after'] = np.array(np.random.rand())
    out['params'] = np.array([N, d, M, opt_itrs, n_subsample_opt, 0, proj_dim, 3])
    save('f18_zellner_gaussian_bpsvi_rand', **out)


# ---------------------------------------------------------------- F19: the greedy-VI coresets on the LOGISTIC model (BASELINE config 3)
def logistic_problem(rng, N, D, S, zero_row=None):
    """Z = y*x rows (model_lr.py:29) with a few rows scaled so that |m| = |z.theta| passes the branch at 100
    (model_lr.py:76) and the beta-likelihood's overflow limits (model_lr.py:85), optionally one all-zero row."""
    X = rng.randn(N, D)
    ths = 2. * np.ones(D) / np.sqrt(D)
    y = np.where(rng.rand(N) <= 1. / (1. + np.exp(-X.dot(ths))), 1., -1.)
    Z = y[:, None] * X
    big = np.array([5, 77, 201, 290])
    Z[big] *= np.array([60., -90., 150., -40.])[:, None]
    if zero_row is not None:
        Z[zero_row] = 0.
    E = rng.randn(S, D)
    return Z, ths, E


def f19_logistic_greedy_vi():
    """BetaCoreset (model_lr.beta_likelihood, beta = 0.1) and SparseVICoreset (model_lr.log_likelihood), full-data mode,
    5 builds x opt_itrs 10:
      * `fixed`   -- the sampler returns the same Theta every call,
      * `laplace` -- the reference's own wiring (zellner_logreg/main.py:139-144): get_laplace (util/opt.py:9-33) on the
                     weighted coreset, theta = mu_w + E.LSig_w^T with the normals E fixed,
      * `laprng`  -- the same with np.random.randn from the global stream; the stream position afterwards is recorded,
      * `lapdiag` -- diag=True Laplace (main.py:105-108), bcores only.
    S = 37 with an all-zero data row (its constant projection row keeps a residue of a few ulp under both likelihoods, so
    it stays a harmless candidate -- if its constant were off by one bit it would become a NaN row and win every argmax);
    S = 100 without one (at S = 100 that row centres to exactly 0 for beta = 0.1: the NaN case, golden F20)."""
    out = {}
    opt_itrs, builds, beta = 10, 5, 0.1
    sched = lambda i: 0.5 / (1. + i)
    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    for S, N, D, zero_row in ((37, 500, 6, 123), (100, 400, 10, None)):
        rng = np.random.RandomState(1900 + S)
        Z, ths, E = logistic_problem(rng, N, D, S, zero_row)
        th_fixed = ths + 0.5 * E
        tag = 'S%d_' % S
        out[tag + 'Z'], out[tag + 'E'], out[tag + 'th_fixed'] = Z, E, th_fixed
        mu0 = np.zeros(D)

        def lap(diag, normals):
            def sampler_w(sz, w, pts):
                if pts.shape[0] == 0:
                    w = np.zeros(1)
                    pts = np.zeros((1, Z.shape[1]))
                muw, LSigw, _ = R.opt.get_laplace(w, pts, mu0, diag)
                if diag:                      # main.py:105-108 hands back matrices where util/opt.py:27-29 has vectors
                    LSigw = np.diag(LSigw)
                return muw + normals(sz, muw.shape[0]).dot(LSigw.T)
            return sampler_w
        samplers = dict(fixed=lambda sz, w, pts: th_fixed,
                        laplace=lap(False, lambda n, d: E),
                        laprng=lap(False, lambda n, d: np.random.randn(n, d)),
                        lapdiag=lap(True, lambda n, d: E))
        with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
            for sn, sampler in samplers.items():
                for nm in ('bcores', 'svi'):
                    if sn == 'lapdiag' and nm == 'svi':
                        continue
                    np.random.seed(190)
                    if nm == 'bcores':
                        prj = R.projector.BetaBlackBoxProjector(sampler, S, R.lr.beta_likelihood, R.lr.log_likelihood, None)
                        alg = R.bcores.BetaCoreset(Z, prj, opt_itrs=opt_itrs, step_sched=sched, beta=beta, learn_beta=False, **fresh())
                    else:
                        prj = R.projector.BlackBoxProjector(sampler, S, R.lr.log_likelihood)
                        alg = R.sparsevi.SparseVICoreset(Z, prj, opt_itrs=opt_itrs, step_sched=sched, **fresh())
                    for m in range(builds):
                        quiet(alg.build, 1, m + 1)
                        out['%s%s_%s_allw_%d' % (tag, sn, nm, m)] = alg.wts.copy()
                        out['%s%s_%s_allidcs_%d' % (tag, sn, nm, m)] = alg.idcs.copy()
                    out['%s%s_%s_theta_last' % (tag, sn, nm)] = np.array(prj.samples)
                    if sn == 'laprng':
                        out['%s%s_%s_rng_after' % (tag, sn, nm)] = np.array(np.random.rand())
            # the sampler by itself: Laplace fit of a weighted handful of rows (both Hessian forms)
            w5 = np.array([0.7, 0., 2.5, 1.1, 0.3])
            p5 = Z[[1, 5, 9, 77, 200]]
            for diag in (False, True):
                mu, L, Li = R.opt.get_laplace(w5, p5, mu0, diag)
                out['%slap%d_mu' % (tag, diag)], out['%slap%d_L' % (tag, diag)], out['%slap%d_Li' % (tag, diag)] = mu, L, Li
            out[tag + 'lap_w'], out[tag + 'lap_rows'] = w5, np.array([1, 5, 9, 77, 200])
    out['beta'], out['opt_itrs'] = np.array(beta), np.array(opt_itrs)
    save('f19_logistic_greedy_vi', **out)


# ---------------------------------------------------------------- F20: constant rows under the logistic beta-likelihood
def f20_logistic_beta_constant_rows():
    """A data row z = 0 projects to S copies of c(beta) = -((b+1)/b 2^-b - 2 2^(-b-1)), the powers by np.power
    (model_lr.py:85).  Whether `c - mean` (projector.py:55) is exactly 0 depends on the last bit of c and on S; recorded
    for S = 16 / 100 / 200 and beta = 0.1 / 0.2 / 0.5, plus BetaCoreset runs where the row is a NaN candidate (S = 16:
    exact zero) and where it keeps a residue (S = 100 with beta = 0.5, S = 200 with beta = 0.1)."""
    rng = np.random.RandomState(20)
    N, D = 300, 5
    zero_at = np.array([7, 40, 41, 200])
    out = dict(zero_at=zero_at)
    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        for S in (16, 100, 200):
            Z, ths, E = logistic_problem(rng, N, D, S)
            Z[zero_at] = 0.
            th = ths + 0.5 * E
            out['S%d_Z' % S], out['S%d_th' % S] = Z, th
            for beta in (0.1, 0.2, 0.5):
                prj = R.projector.BetaBlackBoxProjector(lambda n, w, p: th, S, R.lr.beta_likelihood, R.lr.log_likelihood, None)
                phi = prj.project_f(Z, beta)
                out['S%d_b%g_phi_const' % (S, beta)] = phi[zero_at]
                out['S%d_b%g_norm_pos' % (S, beta)] = np.sqrt((phi ** 2).sum(axis=1)) > 0.
            for beta in ((0.1,) if S != 100 else (0.1, 0.5)):
                prj = R.projector.BetaBlackBoxProjector(lambda n, w, p: th, S, R.lr.beta_likelihood, R.lr.log_likelihood, None)
                alg = R.bcores.BetaCoreset(Z, prj, opt_itrs=5, step_sched=lambda i: 0.5 / (1. + i), beta=beta, learn_beta=False, **fresh())
                for m in range(4):
                    quiet(alg.build, 1, m + 1)
                    out['S%d_b%g_allw_%d' % (S, beta, m)] = alg.wts.copy()
                    out['S%d_b%g_allidcs_%d' % (S, beta, m)] = alg.idcs.copy()
    save('f20_logistic_beta_constant_rows', **out)


# ---------------------------------------------------------------- F21: the logistic drivers' ACTUAL wiring: sub-sampled tangent spaces
def f21_logistic_subsampled():
    """examples/zellner_logreg/main.py:152-160 builds its coresets with n_subsample_opt / n_subsample_select and the Laplace
    sampler on the global NumPy stream: BetaCoreset (beta-likelihood) and SparseVICoreset (log-likelihood) in that mode --
    selections, weights and the stream position after 6 builds (the sampler's randn is drawn BEFORE the randint of the
    sub-sample, bcores.py:39 then :53)."""
    rng = np.random.RandomState(2100)
    N, D, S = 700, 8, 40
    Z, ths, _ = logistic_problem(rng, N, D, S)
    mu0 = np.zeros(D)

    def sampler_w(sz, w, pts):
        if pts.shape[0] == 0:
            w = np.zeros(1)
            pts = np.zeros((1, Z.shape[1]))
        muw, LSigw, _ = R.opt.get_laplace(w, pts, mu0, False)
        return muw + np.random.randn(sz, muw.shape[0]).dot(LSigw.T)
    fresh = lambda: dict(wts=np.array([]), idcs=np.array([], dtype=np.int64), pts=np.array([]))
    out = dict(Z=Z)
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        for nm in ('bcores', 'svi'):
            np.random.seed(210)
            if nm == 'bcores':
                prj = R.projector.BetaBlackBoxProjector(sampler_w, S, R.lr.beta_likelihood, R.lr.log_likelihood, None)
                alg = R.bcores.BetaCoreset(Z, prj, opt_itrs=8, n_subsample_opt=60, n_subsample_select=150,
                                           step_sched=lambda i: 0.5 / (1. + i), beta=.1, learn_beta=False, **fresh())
            else:
                prj = R.projector.BlackBoxProjector(sampler_w, S, R.lr.log_likelihood)
                alg = R.sparsevi.SparseVICoreset(Z, prj, opt_itrs=8, n_subsample_opt=60, n_subsample_select=150,
                                                 step_sched=lambda i: 0.5 / (1. + i), **fresh())
            for m in range(6):
                quiet(alg.build, 1, m + 1)
                out['%s_allw_%d' % (nm, m)] = alg.wts.copy()
                out['%s_allidcs_%d' % (nm, m)] = alg.idcs.copy()
            out['%s_rng_after' % nm] = np.array(np.random.rand())
    save('f21_logistic_subsampled', **out)


# ---------------------------------------------------------------- F22: np.exp's overflow inside the logistic beta-likelihood
def f22_logistic_beta_overflow():
    """model_lr.py:81-86 at |m| around log(DBL_MAX) = 709.78 with SMALL beta: past it np.exp(m) is inf and the reference's
    (1 + inf)**(-beta) is exactly 0 -- a jump of c0 * exp(-beta * 709.78) (0.08 at beta = 0.01) that a smooth evaluation
    misses.  Rows whose sample-0 margin is forced to the listed values; beta in {0.01, 0.05, 0.1}."""
    rng = np.random.RandomState(22)
    N, D, S = 40, 8, 20
    X = rng.randn(N, D)
    th = rng.randn(S, D) / np.sqrt(D)
    y = np.where(rng.rand(N) < 0.5, 1., -1.)
    Z = y[:, None] * X
    mags = (700., 709., 709.7, 709.78, 709.79, 710., 720., 745.2, 760., 799., 801., 1500.,      # (none within rounding of the jump)
            -700., -709.79, -710., -745.2, -746., -1500.)
    for r, mag in enumerate(mags):
        Z[r] = -mag * th[0] / (th[0] ** 2).sum()
    out = dict(Z=Z, th=th, mags=np.array(mags))
    with np.errstate(over='ignore'):
        for beta in (0.01, 0.05, 0.1):
            out['bl_b%g' % beta] = R.lr.beta_likelihood(Z, th, beta)
    out['m'] = -Z.dot(th.T)
    save('f22_logistic_beta_overflow', **out)


if __name__ == '__main__':
    only = set(sys.argv[1:])
    if only:
        for fn in sorted(only):
            globals()[fn]()
        sys.exit(0)
    f1_snnls()
    f2_formulas()
    f3_hilbert_linreg()
    f4_hilbert_logistic_gauss()
    f5_greedy_vi()
    f6_weighted_post()
    f7_nn_opt()
    f8_grouped_vi()
    f9_subsampled_gaussian()
    f10_sampling()
    f11_grouped_subsampled()
    f12_constant_rows()
    f13_greedy_vi_zero_rows()
    f14_zellner_gaussian_driver()
    f15_learn_beta()
    f16_bpsvi()
    f17_uniform_sampling_coreset()
    f18_zellner_gaussian_bpsvi_rand()
    f19_logistic_greedy_vi()
    f20_logistic_beta_constant_rows()
    f21_logistic_subsampled()
    f22_logistic_beta_overflow()
