"""Posterior samplers for the greedy-VI coresets: callables `sampler(S, wts, pts) -> S x D` (projector.py:37,66).

The reference's drivers define these as closures over `weighted_post`
(examples/zellner_neural_linear/main.py:119-124, examples/zellner_gaussian/main.py:90-95):

    muw, LSigw, _ = weighted_post(mu0, Sig0inv, sigsq, pts, wts)
    return muw + np.random.randn(n, muw.shape[0]).dot(LSigw.T)

The classes here compute exactly that (K4 on the device for the Gram of the <= M coreset rows, LAPACK on the host for the
D x D Cholesky, the normals from NumPy's legacy stream) and add ONE thing the closures cannot offer: `prefetch()` draws the
normals of the NEXT call ahead of time.  The fused gradient of BetaCoreset / SparseVI (`bc_vi_gradient`) calls it between
enqueueing a gradient on the GPU and waiting for it -- the ~60-120 us of `randn(S, D)` then run beside K1 instead of between
two launches.  The draws come from the same stream in the same order, and a prefetched matrix is always consumed by the very
next call, so the results and the RNG position after a build are the reference's (tests/test_gpu_storefree.py).
"""
import numpy as np

import scipy.linalg as sl
from scipy.optimize import minimize

from .posterior import gaussian_weighted_post, small_lapack_scope, weighted_post


class _PosteriorSampler:
    def __init__(self, rng=None):
        self._rng = rng                # None: the global np.random stream, like the reference's closures
        self._ahead = None

    def _randn(self, n, d):
        return np.random.randn(n, d) if self._rng is None else self._rng.randn(n, d)

    def scope(self):
        """Context for a run of calls (the optimisation loop of BetaCoreset / SparseVI opens it once): the single-thread
        BLAS limit the D x D solves want is entered once instead of per call."""
        return small_lapack_scope(self._dim())

    def prefetch(self):
        """Draw the next call's normals now (no-op if they are already waiting).  Only legal when the very next draw
        from this sampler's stream is this sampler's next call with the shape of its previous one: the block is taken
        from the stream HERE, so anything else drawn in between would see the stream one block further than the
        reference's (GreedyVICoreset never prefetches after the last gradient of a loop for that reason)."""
        if self._ahead is None and self._shape is not None:
            self._ahead = self._randn(*self._shape)

    def _normals(self, n, d):
        e, self._ahead = self._ahead, None
        self._shape = (n, d)
        if e is None:
            return self._randn(n, d)
        if e.shape != (n, d):
            # the block was already drawn from the stream: dropping it silently would leave the stream one S x D block
            # ahead of the reference's from here on
            raise RuntimeError('sampler called for %r normals while a prefetched block of shape %r is pending; '
                               'prefetch() may only precede a call of the same shape' % ((n, d), e.shape))
        return e


class LinregPosteriorSampler(_PosteriorSampler):
    """theta ~ N(mu_w, Sigma_w) of Bayesian linear regression on the weighted coreset (model_linreg.py:25-34; the
    `sampler_w` of zellner_neural_linear/main.py:119-124).  Rows pts = [x (D), y]."""

    def __init__(self, th0, Sig0inv, sigsq, rng=None, ctx=None):
        super().__init__(rng)
        self.th0, self.Sig0inv, self.sigsq = np.asarray(th0, dtype=np.float64), np.asarray(Sig0inv, dtype=np.float64), float(sigsq)
        self.ctx = ctx
        self._shape = None

    def _dim(self):
        return self.th0.shape[0]

    def __call__(self, n, wts, pts):
        d = self.th0.shape[0]
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, d + 1))
        muw, LSigw, _ = weighted_post(self.th0, self.Sig0inv, self.sigsq, pts, wts, ctx=self.ctx)
        return muw + self._normals(n, d).dot(LSigw.T)


class GaussianPosteriorSampler(_PosteriorSampler):
    """The Gaussian location model's `sampler_w` (zellner_gaussian/main.py:90-95, gaussian.py:28-32)."""

    def __init__(self, mu0, Sig0inv, Siginv, rng=None, ctx=None):
        super().__init__(rng)
        self.mu0, self.Sig0inv, self.Siginv = np.asarray(mu0, dtype=np.float64), np.asarray(Sig0inv, dtype=np.float64), np.asarray(Siginv, dtype=np.float64)
        self.ctx = ctx
        self._shape = None

    def _dim(self):
        return self.mu0.shape[0]

    def __call__(self, n, wts, pts):
        d = self.mu0.shape[0]
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, d))
        muw, LSigw, _ = gaussian_weighted_post(self.mu0, self.Sig0inv, self.Siginv, pts, wts, ctx=self.ctx)
        return muw + self._normals(n, d).dot(LSigw.T)


# ---- logistic regression: Laplace approximation around the mode of the weighted log-joint (<= M coreset rows: host work
# in the reference too; the formulas keep model_lr.py's expression order so that, on one host, the
# optimiser walks the same path as the reference's)
def _lr_m(z, th):
    z = np.atleast_2d(z)
    th = np.atleast_2d(th)
    m = -z.dot(th.T)
    return z, th, m, m < 100


def _lr_log_joint(z, th, wts):
    """model_lr.py:72-79, 88-93"""
    z, th, m, small = _lr_m(z, th)
    m[small] = -np.log1p(np.exp(m[small]))
    m[np.logical_not(small)] = -m[np.logical_not(small)]
    return (wts[:, np.newaxis] * m).sum(axis=0) + (-0.5 * th.shape[1] * np.log(2. * np.pi) - 0.5 * (th ** 2).sum(axis=1))


def _lr_grad_log_joint(z, th, wts):
    """model_lr.py:98-105, 116-121"""
    z, th, m, small = _lr_m(z, th)
    m[small] = np.exp(m[small]) / (1. + np.exp(m[small]))
    m[np.logical_not(small)] = 1.
    return -th + (wts[:, np.newaxis, np.newaxis] * (m[:, :, np.newaxis] * z[:, np.newaxis, :])).sum(axis=0)


def _lr_hess_log_joint(z, th, wts, diag):
    """model_lr.py:123-137 (full) / 139-153 (diagonal)"""
    z, th, m, small = _lr_m(z, th)
    m[small] = np.exp(m[small]) / (1. + np.exp(m[small])) ** 2
    m[np.logical_not(small)] = 0.
    if diag:
        hl = -m[:, :, np.newaxis] * z[:, np.newaxis, :] ** 2
        return np.tile(-np.ones(th.shape[1]), (th.shape[0], 1)) + (wts[:, np.newaxis, np.newaxis] * hl).sum(axis=0)
    hl = -m[:, :, np.newaxis, np.newaxis] * z[:, np.newaxis, :, np.newaxis] * z[:, np.newaxis, np.newaxis, :]
    return np.tile(-np.eye(th.shape[1]), (th.shape[0], 1, 1)) + (wts[:, np.newaxis, np.newaxis, np.newaxis] * hl).sum(axis=0)


def _lr_mode_newton(Zw, ww, mu0, max_iter=200):
    """The mode of the weighted log-joint by damped Newton steps.  The log-joint is strictly concave (the N(0, I) prior puts
    its Hessian below -I), so the maximiser is unique: it is the point scipy's BFGS converges to, reached here in ~10
    iterations of D x D linear algebra instead of hundreds of rank-two updates (410 -> ~1 ms per sampler call at M = 100,
    D = 128 with weights N/M: the BFGS iteration count grows with the posterior's condition number)."""
    d = mu0.shape[0]
    mu = np.array(mu0, dtype=np.float64)

    def value(th):
        m = -Zw.dot(th)
        return -(ww * (np.maximum(m, 0.) + np.log1p(np.exp(-np.fabs(m))))).sum() - 0.5 * d * np.log(2. * np.pi) - 0.5 * th.dot(th)
    f = value(mu)
    eye = np.eye(d)
    for _ in range(max_iter):
        m = -Zw.dot(mu)
        p = 0.5 * (1. + np.tanh(0.5 * m))                     # e^m / (1 + e^m), overflow-free
        g = -mu + Zw.T.dot(ww * p)
        H = eye + (Zw * (ww * p * (1. - p))[:, np.newaxis]).T.dot(Zw)      # minus the Hessian: positive definite
        step = sl.cho_solve(sl.cho_factor(H, lower=True, check_finite=False), g, check_finite=False)
        t, dec = 1., g.dot(step)
        while True:                                           # backtracking: Newton's full step once near the mode
            cand = mu + t * step
            fc = value(cand)
            if fc >= f + 1e-4 * t * dec or t < 1e-10:
                break
            t *= 0.5
        mu, f = cand, fc
        # stop after a FULL Newton step whose decrement g.H^-1.g (twice the distance of f from its maximum) was already at
        # the rounding floor of f: the point before it was within sqrt(2 dec) <= 2e-6 of the mode, the step squares that.
        # (The gradient cannot be driven below ~ max(w) * eps -- with weights N/M a criterion on |step| alone would spin in
        # rounding noise for the remaining iterations.)
        if (t == 1. and dec <= 64. * np.finfo(np.float64).eps * (1. + abs(f))) or np.fabs(t * step).max() <= 1e-13 * (1. + np.fabs(mu).max()):
            break
    return mu


def logistic_laplace(wts, Z, mu0, diag=False, rng=None, solver='bfgs', newton_start=None):
    """`get_laplace` (examples/zellner_logreg/main.py:86-111 == bayesiancoresets/util/opt.py:9-33): (mu, LSig, LSigInv) of
    the Laplace approximation N(mu, LSig LSig^T) to the posterior of the rows Z with weights wts; the mode comes from
    scipy.optimize.minimize's default method started at mu0 (third-party arithmetic, shared with the reference, not
    restated), a failing optimisation restarts from a perturbed mu0 up to ten times.  diag=True returns the driver's
    diagonal MATRICES (main.py:105-108; util/opt.py:27-29 has vectors, which its own caller cannot multiply with).
    solver='newton' (not the reference's call, same unique mode): see _lr_mode_newton; BFGS stops at a gradient norm of
    1e-5, i.e. ~1e-6 from the mode in theta, which is how far the two answers are apart."""
    trials = 10
    Zw = Z[wts > 0, :]
    ww = wts[wts > 0]
    if solver == 'newton':
        # (newton_start: where to begin -- the maximiser is unique, so the start only decides how many steps it takes)
        mu = _lr_mode_newton(Zw, ww, mu0 if newton_start is None else newton_start)
    elif solver != 'bfgs':
        raise ValueError("solver must be 'bfgs' (the reference's scipy.optimize.minimize call) or 'newton'")
    while solver == 'bfgs':
        try:
            res = minimize(lambda mu: -_lr_log_joint(Zw, mu, ww)[0], mu0, jac=lambda mu: -_lr_grad_log_joint(Zw, mu, ww)[0, :])
        except Exception:
            mu0 = mu0.copy()
            mu0 += np.sqrt((mu0 ** 2).sum()) * 0.1 * (np.random.randn(mu0.shape[0]) if rng is None else rng.randn(mu0.shape[0]))
            trials -= 1
            if trials <= 0:
                raise RuntimeError('logistic_laplace: the mode search failed ten times')     # (the reference dies on `res` here)
            continue
        mu = res.x
        break
    if diag:
        sq = np.sqrt(-_lr_hess_log_joint(Zw, mu, ww, True)[0, :])
        return mu, np.diag(1. / sq), np.diag(sq)
    if solver == 'newton':
        # (the reference's Hessian expression builds M x D x D temporaries -- 13 MB and ~10 ms at M = 100, D = 128; the same
        # matrix as one BLAS product, equal to rounding)
        m = -Zw.dot(mu)
        p = 0.5 * (1. + np.tanh(0.5 * m))
        LSigInv = np.linalg.cholesky(np.eye(mu.shape[0]) + (Zw * (ww * p * (1. - p))[:, np.newaxis]).T.dot(Zw))
    else:
        LSigInv = np.linalg.cholesky(-_lr_hess_log_joint(Zw, mu, ww, False)[0, :, :])
    LSig = sl.solve_triangular(LSigInv, np.eye(LSigInv.shape[0]), lower=True, overwrite_b=True, check_finite=False)
    return mu, LSig, LSigInv


class LogisticLaplaceSampler(_PosteriorSampler):
    """The logistic drivers' `sampler_w` (examples/zellner_logreg/main.py:139-144): theta = mu_w + randn(S, D).LSig_w^T with
    (mu_w, LSig_w) the Laplace fit of the weighted coreset rows pts = y*x (model_lr.py:29); an empty coreset gives the
    prior N(0, I).  `mu0` is the optimiser's starting point (the drivers pass the prior mean), `diag` their `graddiag`."""

    def __init__(self, mu0, diag=False, rng=None, solver='bfgs'):
        super().__init__(rng)
        self.mu0, self.diag, self.solver = np.asarray(mu0, dtype=np.float64), bool(diag), solver
        self._shape = None
        self._mode = None             # solver='newton': the previous call's mode, the next call's starting point (consecutive
                                      # gradients move the weights a little: 2-3 Newton steps instead of 10-20 from mu0)

    def _dim(self):
        return self.mu0.shape[0]

    def prefetch(self):
        """No look-ahead with the reference's mode search: a failing `minimize` draws its restart perturbation `randn(D)`
        BEFORE the sample normals (main.py:96-101, then :144), so a block drawn ahead would come from the wrong place in the
        stream on such a call.  The Newton search draws nothing, and prefetching stays on for it."""
        if self.solver != 'bfgs':
            super().prefetch()

    def __call__(self, n, wts, pts):
        d = self.mu0.shape[0]
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, d))
        # (D x D and M x D linear algebra, thousands of tiny BLAS calls inside the mode search: one thread -- a 128-thread
        # pool spends milliseconds synchronising on each; re-entrant, the optimisation loop usually holds the scope already)
        with small_lapack_scope(d):
            muw, LSigw, _ = logistic_laplace(np.asarray(wts, dtype=np.float64), np.atleast_2d(pts), self.mu0, self.diag, rng=self._rng,
                                             solver=self.solver, newton_start=self._mode if self.solver == 'newton' else None)
            if self.solver == 'newton' and np.all(np.isfinite(muw)):
                self._mode = muw.copy()
            return muw + self._normals(n, d).dot(LSigw.T)
