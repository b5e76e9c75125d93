"""NumPy restatement of the per-datapoint (beta-)likelihood formulas and the
weighted Gaussian posterior (oracle; tests only).

Expression order is kept identical to the reference so that, given the same
BLAS, results are bit-identical:
  examples/common/model_linreg.py:4-10,25-34      linreg log-lik, weighted_post
  examples/common/model_neurlinr.py:90-97,102-110 same log-lik, beta-likelihood
  examples/common/model_lr.py:72-86               logistic log-lik, beta-likelihood
  examples/common/model_lr.py:88-153              log-prior / log-joint and its theta-gradient, Hessian, diagonal Hessian
  bayesiancoresets/util/opt.py:9-33               get_laplace (== examples/zellner_logreg/main.py:86-111), the logistic drivers' sampler
  examples/common/gaussian.py:7-15,28-62          Gaussian-location model
"""
import numpy as np
import scipy.linalg as sl


def _split_xy(z):
    z = np.atleast_2d(z)
    return z[:, :-1], z[:, -1]


def linreg_loglik(z, th, sigsq):
    """model_linreg.py:4-10 == model_neurlinr.py:90-97 (expanded square kept)."""
    x, y = _split_xy(z)
    th = np.atleast_2d(th)
    p = x.dot(th.T)
    yc = y[:, np.newaxis]
    return -1. / 2. * np.log(2. * np.pi * sigsq) - 1. / (2. * sigsq) * (yc ** 2 - 2 * p * yc + p ** 2)


def linreg_beta_lik(z, th, beta, sigsq):
    """model_neurlinr.py:102-110 (loss-signed)."""
    x, y = _split_xy(z)
    th = np.atleast_2d(th)
    p = x.dot(th.T)
    yc = y[:, np.newaxis]
    return 1. / (2 * np.pi * sigsq) ** (beta / 2.) * (
        -(beta + 1.) / beta * np.exp(-beta / (2. * sigsq) * (yc ** 2 - 2 * p * yc + p ** 2))
        + 1. / np.sqrt(1. + beta))


def logistic_loglik(z, th):
    """model_lr.py:72-79 (branch at m < 100 kept)."""
    z = np.atleast_2d(z)
    th = np.atleast_2d(th)
    m = -z.dot(th.T)
    small = m < 100
    m[small] = -np.log1p(np.exp(m[small]))
    m[np.logical_not(small)] = -m[np.logical_not(small)]
    return m


def logistic_beta_lik(z, th, beta):
    """model_lr.py:81-86 (relies on IEEE inf -> 0 limits)."""
    z = np.atleast_2d(z)
    th = np.atleast_2d(th)
    m = -z.dot(th.T)
    with np.errstate(over='ignore'):
        m = -(((beta + 1.) / beta) * (1 + np.exp(m)) ** (-beta)
              - ((1 + np.exp(m)) ** (-beta - 1.) + (1 + np.exp(-m)) ** (-beta - 1.)))
    return m


def _gauss_quadratic(x, th, Siginv):
    x = np.atleast_2d(x)
    th = np.atleast_2d(th)
    xSx = (x * (x.dot(Siginv))).sum(axis=1)
    tSt = (th * (th.dot(Siginv))).sum(axis=1)
    xSt = x.dot(Siginv.dot(th.T))
    return x, th, xSx, tSt, xSt


def gauss_loglik(x, th, Siginv, logdetSig):
    """gaussian.py:7-15 (without the stray print)."""
    x, th, xSx, tSt, xSt = _gauss_quadratic(x, th, Siginv)
    return -x.shape[1] / 2 * np.log(2 * np.pi) - 1. / 2. * logdetSig - 1. / 2. * (xSx[:, np.newaxis] + tSt - 2 * xSt)


def gauss_beta_lik(x, th, beta, Siginv, logdetSig):
    """gaussian.py:34-44 (gain-signed; the unused normaliser is not formed)."""
    x, th, xSx, tSt, xSt = _gauss_quadratic(x, th, Siginv)
    d = float(x.shape[1])
    t1 = (1. / beta) * np.exp(-.5 * beta * (xSx[:, np.newaxis] + tSt - 2 * xSt))
    t2 = (1 + beta) ** (-.5 * d - 1)
    return t1 - t2


def gauss_beta_grad(x, th, beta, Siginv, logdetSig):
    """gaussian.py:46-62 (d/d beta of the beta-likelihood)."""
    x, th, xSx, tSt, xSt = _gauss_quadratic(x, th, Siginv)
    d = float(x.shape[1])
    logc = np.log((2 * np.pi) ** (-.5 * d) * (np.exp(logdetSig) ** (-.5)))
    q = xSx[:, np.newaxis] + tSt - 2 * xSt
    gq = np.exp(-.5 * beta * q)
    t11 = (1. / beta) * gq
    t12 = (1 + beta) ** (-.5 * d - 1.)
    t1 = logc * (t11 - t12)
    t2 = 1. / (beta) ** 2 * gq
    t3 = 1. / (2. * beta) * q * gq
    t4 = (1 + beta) ** (-.5 * d - 1.) * np.log(1. + beta)
    return t1 - t2 - t3 - t4


def linreg_weighted_post(th0, Sig0inv, sigsq, z, w):
    """model_linreg.py:25-34 == model_neurlinr.py:115-122.

    NOTE: returns LSigp @ LSigp.T @ (...), i.e. C^-1 C^-T rather
    than the true C^-T C^-1 -- reproduced on purpose, parity target is the
    reference's output."""
    X, Y = _split_xy(z)
    C = np.linalg.cholesky(Sig0inv + (w[:, np.newaxis] * X).T.dot(X) / sigsq)
    Ci = sl.solve_triangular(C, np.eye(C.shape[0]), lower=True, overwrite_b=True, check_finite=False)
    mu = np.dot(Ci.dot(Ci.T), np.dot(Sig0inv, th0) + (w[:, np.newaxis] * Y[:, np.newaxis] * X).sum(axis=0) / sigsq)
    return mu, Ci, C


def gauss_weighted_post(th0, Sig0inv, Siginv, x, w):
    """gaussian.py:28-32 (closed form, isotropic-safe)."""
    C = np.linalg.cholesky(Sig0inv + w.sum() * Siginv)
    Ci = sl.solve_triangular(C, np.eye(C.shape[0]), lower=True, overwrite_b=True, check_finite=False)
    mu = np.dot(Ci.dot(Ci.T), np.dot(Sig0inv, th0) + np.dot(Siginv, (w[:, np.newaxis] * x).sum(axis=0)))
    return mu, Ci, C


def linreg_xtwx(z, w):
    """The two row-reductions inside weighted_post (model_linreg.py:29,31):
    X^T diag(w) X  and  X^T (w*y).  Kernel K4's outputs."""
    X, Y = _split_xy(z)
    return (w[:, np.newaxis] * X).T.dot(X), (w[:, np.newaxis] * Y[:, np.newaxis] * X).sum(axis=0)


def gaussian_KL(mu0, Sig0, mu1, Sig1inv):
    """gaussian.py:22-26: KL( N(mu0, Sig0) || N(mu1, Sig1) )."""
    t1 = np.dot(Sig1inv, Sig0).trace()
    t2 = np.dot((mu1 - mu0), np.dot(Sig1inv, mu1 - mu0))
    t3 = -np.linalg.slogdet(Sig1inv)[1] - np.linalg.slogdet(Sig0)[1]
    return 0.5 * (t1 + t2 + t3 - mu0.shape[0])


# ---- x-gradients of the log-likelihood (BatchPSVICoreset: bpsvi.py:39-40 through projector.py:27-32)
def linreg_grad_x_loglik(z, th, sigsq):
    """model_linreg.py:12-17 (== model_neurlinr.py:99-100): (M, S, D+1), the last column is d/dy's stand-in 1"""
    x, y = _split_xy(z)
    th = np.atleast_2d(th)
    return 1. / sigsq * (y[:, np.newaxis] - x.dot(th.T))[:, :, np.newaxis] \
        * np.hstack((th, np.ones(th.shape[0])[:, np.newaxis]))[np.newaxis, :, :]


def logistic_grad_z_loglik(z, th):
    """model_lr.py:107-114: (M, S, D)"""
    z = np.atleast_2d(z)
    th = np.atleast_2d(th)
    m = -z.dot(th.T)
    idcs = m < 100
    m[idcs] = np.exp(m[idcs]) / (1. + np.exp(m[idcs]))
    m[np.logical_not(idcs)] = 1.
    return m[:, :, np.newaxis] * th[np.newaxis, :, :]


def gauss_grad_x_loglik(x, th, Siginv):
    """gaussian.py:17-20: (M, S, d)"""
    x = np.atleast_2d(x)
    th = np.atleast_2d(th)
    return th.dot(Siginv)[np.newaxis, :, :] - x.dot(Siginv)[:, np.newaxis, :]



# ---- the Laplace sampler of the logistic drivers (<= M coreset rows; host arithmetic in the reference too)
def logistic_log_prior(th):
    """model_lr.py:88-90"""
    th = np.atleast_2d(th)
    return -0.5 * th.shape[1] * np.log(2. * np.pi) - 0.5 * (th ** 2).sum(axis=1)


def logistic_log_joint(z, th, wts):
    """model_lr.py:92-93"""
    return (wts[:, np.newaxis] * logistic_loglik(z, th)).sum(axis=0) + logistic_log_prior(th)


def _logistic_sigmoid_m(z, th):
    """the shared head of model_lr.py:98-105 / :127-134: m = -z.th^T, then e^m/(1+e^m) below the branch at 100"""
    z = np.atleast_2d(z)
    th = np.atleast_2d(th)
    m = -z.dot(th.T)
    return z, th, m, m < 100


def logistic_grad_th_loglik(z, th):
    """model_lr.py:98-105: (M, S, D)"""
    z, th, m, idcs = _logistic_sigmoid_m(z, th)
    m[idcs] = np.exp(m[idcs]) / (1. + np.exp(m[idcs]))
    m[np.logical_not(idcs)] = 1.
    return m[:, :, np.newaxis] * z[:, np.newaxis, :]


def logistic_grad_th_log_joint(z, th, wts):
    """model_lr.py:116-121 (the prior's gradient is -th)"""
    return -np.atleast_2d(th) + (wts[:, np.newaxis, np.newaxis] * logistic_grad_th_loglik(z, th)).sum(axis=0)


def logistic_hess_th_log_joint(z, th, wts):
    """model_lr.py:123-137: (S, D, D)"""
    z, th, m, idcs = _logistic_sigmoid_m(z, th)
    m[idcs] = np.exp(m[idcs]) / (1. + np.exp(m[idcs])) ** 2
    m[np.logical_not(idcs)] = 0.
    hl = -m[:, :, np.newaxis, np.newaxis] * z[:, np.newaxis, :, np.newaxis] * z[:, np.newaxis, np.newaxis, :]
    return np.tile(-np.eye(th.shape[1]), (th.shape[0], 1, 1)) + (wts[:, np.newaxis, np.newaxis, np.newaxis] * hl).sum(axis=0)


def logistic_diag_hess_th_log_joint(z, th, wts):
    """model_lr.py:139-153: (S, D)"""
    z, th, m, idcs = _logistic_sigmoid_m(z, th)
    m[idcs] = np.exp(m[idcs]) / (1. + np.exp(m[idcs])) ** 2
    m[np.logical_not(idcs)] = 0.
    dl = -m[:, :, np.newaxis] * z[:, np.newaxis, :] ** 2
    return np.tile(-np.ones(th.shape[1]), (th.shape[0], 1)) + (wts[:, np.newaxis, np.newaxis] * dl).sum(axis=0)


def logistic_laplace(wts, Z, mu0, diag=False):
    """examples/zellner_logreg/main.py:86-111 (== bayesiancoresets/util/opt.py:9-33 for diag=False; the driver's copy
    returns matrices for diag=True, which is what its sampler_w multiplies with): N(mu, LSig LSig^T) around the mode of
    the weighted log-joint, found with scipy.optimize.minimize's default (BFGS) from mu0 -- third-party, shared."""
    from scipy.optimize import minimize
    trials = 10
    Zw = Z[wts > 0, :]
    ww = wts[wts > 0]
    while True:
        try:
            res = minimize(lambda mu: -logistic_log_joint(Zw, mu, ww)[0], mu0,
                           jac=lambda mu: -logistic_grad_th_log_joint(Zw, mu, ww)[0, :])
        except Exception:
            mu0 = mu0.copy()
            mu0 += np.sqrt((mu0 ** 2).sum()) * 0.1 * np.random.randn(mu0.shape[0])
            trials -= 1
            if trials <= 0:
                break
            continue
        break
    mu = res.x
    if diag:
        sq = np.sqrt(-logistic_diag_hess_th_log_joint(Zw, mu, ww)[0, :])
        LSigInv, LSig = np.diag(sq), np.diag(1. / sq)
    else:
        LSigInv = np.linalg.cholesky(-logistic_hess_th_log_joint(Zw, mu, ww)[0, :, :])
        LSig = sl.solve_triangular(LSigInv, np.eye(LSigInv.shape[0]), lower=True, overwrite_b=True, check_finite=False)
    return mu, LSig, LSigInv
