"""Weighted Gaussian posteriors (the samplers' `weighted_post`), with the row reductions on the GPU.

  weighted_post(th0, Sig0inv, sigsq, z, w)            linear regression,
      examples/common/model_linreg.py:25-34 == model_neurlinr.py:115-122
  gaussian_weighted_post(th0, Sig0inv, Siginv, x, w)  Gaussian location model, gaussian.py:28-32

The O(N D^2) part -- X^T diag(w) X and X^T (w*y) -- is kernel K4 (bc_weighted_gram, fp64 MFMA);
the D x D Cholesky / triangular solve stay in LAPACK on the host, with the reference's exact
expressions.  NOTE: the reference forms  LSigp @ LSigp.T  (= C^-1 C^-T) where the
true posterior covariance is C^-T C^-1; that is reproduced on purpose -- parity target is the
reference's output.  `weighted_post_corrected` is the mathematically right variant.
"""
import ctypes as C

import numpy as np
import scipy.linalg as sl

import contextlib

from . import _native as N
from .device import DeviceData, _ptr, default_context

try:
    from threadpoolctl import ThreadpoolController as _Controller
except Exception:                                     # pragma: no cover
    _Controller = None
_controller = None


_limit_depth = 0


@contextlib.contextmanager
def small_lapack_scope(d):
    """The D x D Cholesky / triangular solve of a coreset posterior is microseconds of work; on a
    many-core host a multi-threaded BLAS spends milliseconds synchronising its pool on it (measured:
    4.9 ms per solve_triangular at D = 64 with 128 threads).  Same routines, one thread.  The controller is
    created once: discovering the loaded BLAS libraries costs ~0.9 ms, more than the solve it guards.
    Re-entrant: entering the limit costs ~10 us, so a caller that runs many posteriors in a row (the greedy-VI
    optimisation loop) opens the scope once and the calls inside find it open."""
    global _controller, _limit_depth
    if _Controller is None or d > 512 or _limit_depth > 0:
        yield
        return
    if _controller is None:
        _controller = _Controller()
    with _controller.limit(limits=1, user_api='blas'):
        _limit_depth += 1
        try:
            yield
        finally:
            _limit_depth -= 1


_small_lapack = small_lapack_scope


def weighted_gram(z, w=None, ctx=None, comm=None):
    """(X^T diag(w) X, X^T (w*y)) for rows z = [x, y]; `z` may be an ndarray or a DeviceData.
    With `comm` the rows are this rank's shard and the two results are summed over ranks."""
    ctx = ctx or default_context()
    if w is not None:
        w = np.ascontiguousarray(w, dtype=np.float64)
    if not isinstance(z, DeviceData):
        z = np.ascontiguousarray(np.atleast_2d(z), dtype=np.float64)
        if w is not None and w.shape != (z.shape[0],):
            raise ValueError('w must have one weight per row')
        if z.shape[0] * (z.shape[1] + 1) <= 60000 and z.shape[1] * z.shape[1] <= 60000:
            # coreset-sized call (the samplers, once per gradient): rows, weights and results in one native call
            d = z.shape[1] - 1
            G, v = np.empty((max(d, 0), max(d, 0))), np.empty(max(d, 0))
            N.call('bc_weighted_gram_host', ctx.h, _ptr(z), int(z.shape[0]), int(z.shape[1]), _ptr(w) if w is not None else None,
                   _ptr(G), _ptr(v))
            if comm is not None and comm.world > 1:
                G = comm.sum_in_rank_order(G)
                v = comm.sum_in_rank_order(v)
            return G, v
        data = DeviceData(z, ctx=ctx)
    else:
        data = z
    n, dz = data.shape
    d = dz - 1
    G = np.empty((d, d))
    v = np.empty(d)
    if w is not None and w.shape != (n,):
        raise ValueError('w must have one weight per row')
    N.call('bc_weighted_gram', data.ctx.h, data.h, _ptr(w) if w is not None else None, _ptr(G), _ptr(v))
    if comm is not None and comm.world > 1:
        G = comm.sum_in_rank_order(G)
        v = comm.sum_in_rank_order(v)
    return G, v


def weighted_post(th0, Sig0inv, sigsq, z, w, ctx=None, comm=None):
    G, v = weighted_gram(z, w, ctx=ctx, comm=comm)
    with _small_lapack(G.shape[0]):
        LSigpInv = np.linalg.cholesky(Sig0inv + G / sigsq)
        LSigp = sl.solve_triangular(LSigpInv, np.eye(LSigpInv.shape[0]), lower=True, overwrite_b=True, check_finite=False)
        mup = np.dot(LSigp.dot(LSigp.T), np.dot(Sig0inv, th0) + v / sigsq)
    return mup, LSigp, LSigpInv


def weighted_post_corrected(th0, Sig0inv, sigsq, z, w, ctx=None, comm=None):
    """Same inputs, true posterior mean Sigma_p (Sig0inv th0 + X^T W y / sigsq) with Sigma_p = C^-T C^-1."""
    G, v = weighted_gram(z, w, ctx=ctx, comm=comm)
    LSigpInv = np.linalg.cholesky(Sig0inv + G / sigsq)
    LSigp = sl.solve_triangular(LSigpInv, np.eye(LSigpInv.shape[0]), lower=True, check_finite=False)
    mup = np.dot(LSigp.T.dot(LSigp), np.dot(Sig0inv, th0) + v / sigsq)
    return mup, LSigp, LSigpInv


def gaussian_weighted_post(th0, Sig0inv, Siginv, x, w, ctx=None, comm=None):
    """gaussian.py:28-32.  sum_i w_i x_i is taken from K4 by appending a column of ones as `y`."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    w = np.ascontiguousarray(w, dtype=np.float64)
    _, xw = weighted_gram(np.hstack((x, np.ones((x.shape[0], 1)))), w, ctx=ctx, comm=comm)
    wsum = w.sum()
    if comm is not None and comm.world > 1:
        wsum = comm.sum_in_rank_order(np.array([wsum]))[0]
    with _small_lapack(Siginv.shape[0]):
        LSigpInv = np.linalg.cholesky(Sig0inv + wsum * Siginv)
        LSigp = sl.solve_triangular(LSigpInv, np.eye(LSigpInv.shape[0]), lower=True, overwrite_b=True, check_finite=False)
        mup = np.dot(LSigp.dot(LSigp.T), np.dot(Sig0inv, th0) + np.dot(Siginv, xw))
    return mup, LSigp, LSigpInv
