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
        """Draw the next call's normals now (no-op if they are already waiting)."""
        if self._ahead is None and self._shape is not None:
            self._ahead = self._randn(*self._shape)

    def _normals(self, n, d):
        e, self._ahead = self._ahead, None
        self._shape = (n, d)
        if e is not None and e.shape == (n, d):
            return e
        return self._randn(n, d)


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
