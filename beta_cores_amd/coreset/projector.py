"""Projectors: data rows -> row-centred (beta-)log-likelihood vectors Phi (N x S).

Two families, one protocol (bayesiancoresets/coreset/projector.py:5-66:
`project(pts, grad=False)`, `project_f(pts, beta, grad=False)`, `update(wts, pts)`):

* BlackBoxProjector / BetaBlackBoxProjector -- the reference's classes: arbitrary host
  callables produce the N x S array, which the device solvers then upload.
* DeviceProjector / DeviceBetaProjector -- K1 on the GPU: the likelihood is one of the
  models in beta_cores_amd.likelihoods, the contraction Z.Theta^T, the formula, the
  row-centring, the row norms and the column sums run in one kernel and Phi never
  leaves HBM (a DevicePhi is returned).
"""
import ctypes as C
import weakref

import numpy as np

from .. import _native as N
from ..device import DeviceData, DevicePhi, _ptr, default_context


class Projector(object):
    def project(self, pts, grad=False):
        raise NotImplementedError

    def update(self, wts, pts):
        raise NotImplementedError


class BlackBoxProjector(Projector):
    """projector.py:12-37"""

    def __init__(self, sampler, projection_dimension, loglikelihood, grad_loglikelihood=None, **kwargs):
        self.projection_dimension = projection_dimension
        self.sampler = sampler
        self.loglikelihood = loglikelihood
        self.grad_loglikelihood = grad_loglikelihood
        self.update(np.array([]), np.array([]))
        self.encoder = kwargs.get('nl', None)      # optional learned feature map

    def project(self, pts, grad=False):
        args = (pts, self.samples) + ((self.encoder,) if self.encoder else ())
        lls = self.loglikelihood(*args)
        lls -= lls.mean(axis=1)[:, np.newaxis]
        if not grad:
            return lls
        if self.grad_loglikelihood is None:
            raise ValueError('grad_loglikelihood was requested but not initialized in BlackBoxProjector.project')
        glls = self.grad_loglikelihood(pts, self.samples)
        glls -= glls.mean(axis=2)[:, :, np.newaxis]
        return lls, glls

    def update(self, wts, pts):
        self.samples = self.sampler(self.projection_dimension, wts, pts)


class BetaBlackBoxProjector(Projector):
    """projector.py:39-66"""

    def __init__(self, sampler, projection_dimension, beta_likelihood, loglikelihood, beta_gradient, **kwargs):
        self.projection_dimension = projection_dimension
        self.sampler = sampler
        self.beta_likelihood = beta_likelihood
        self.loglikelihood = loglikelihood
        self.beta_gradient = beta_gradient
        self.update(np.array([]), np.array([]))
        self.encoder = kwargs.get('nl', None)

    def project_f(self, pts, beta, grad=False):
        args = (pts, self.samples, beta) + ((self.encoder,) if self.encoder else ())
        bls = self.beta_likelihood(*args)
        bls -= bls.mean(axis=1)[:, np.newaxis]
        if not grad:
            return bls
        if self.beta_gradient is None:
            raise ValueError('grad_loglikelihood was requested but not initialized in BlackBoxProjector.project')
        glls = self.beta_gradient(pts, self.samples, beta)
        glls -= glls.mean(axis=1)[:, np.newaxis]
        return bls, glls

    def update(self, wts, pts):
        self.samples = self.sampler(self.projection_dimension, wts, pts)


_SMALL_ROWS = 4096     # below this an input is treated as transient (coreset points, sub-samples)


def _pool_shutdown(state):
    state['alive'] = False
    destroy = N.load().bc_phi_destroy
    for lst in state['free'].values():
        for _, h in lst:
            destroy(h)
    state['free'].clear()


class _PhiPool:
    """Phi buffers for small projections, recycled instead of hipMalloc'ed / hipFree'd per call.
    A handle is lent to exactly one DevicePhi wrapper; when that wrapper is garbage collected the
    handle comes back (so two live results never alias).  Handles still on loan when the pool dies
    stay valid and are destroyed by their wrapper."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.state = {'alive': True, 'free': {}}     # shared with the release callbacks
        self._fin = weakref.finalize(self, _pool_shutdown, self.state)

    def acquire(self, n_rows, s):
        lst = self.state['free'].setdefault(s, [])
        for i, (cap, h) in enumerate(lst):
            if cap >= n_rows:
                lst.pop(i)
                return cap, h
        cap = max(256, 1 << int(np.ceil(np.log2(max(n_rows, 1)))))
        h = C.c_void_p()
        N.call('bc_phi_create', self.ctx.h, int(cap), int(s), C.byref(h))
        return cap, h

    def releaser(self, cap, s):
        state = self.state

        def release(handle):
            if state['alive']:
                state['free'].setdefault(s, []).append((cap, handle))
            else:
                N.load().bc_phi_destroy(handle)
        return release


class _DeviceProjectorBase(Projector):
    def __init__(self, sampler, projection_dimension, model, ctx=None):
        self.projection_dimension = projection_dimension
        self.sampler = sampler
        self.model = model
        self.ctx = ctx or default_context()
        self._data_cache = {}      # id(ndarray) -> (weakref, DeviceData)       large, repeatedly projected arrays
        self._phi_cache = {}       # (id(DeviceData), model_id, S) -> DevicePhi  whose buffers get reused
        self._slots = {}           # dz -> DeviceData slot for small transient inputs
        self._pool = _PhiPool(self.ctx)
        self.update(np.array([]), np.array([]))

    def update(self, wts, pts):
        self.samples = self.sampler(self.projection_dimension, wts, pts)

    # -- data residency: a large array projected repeatedly is uploaded once; small ones go through a slot
    def device_data(self, pts):
        if isinstance(pts, DeviceData):
            return pts, False
        pts = np.atleast_2d(pts)
        if pts.shape[0] < _SMALL_ROWS:
            slot = self._slots.get(pts.shape[1])
            if slot is None:
                slot = self._slots[pts.shape[1]] = DeviceData.slot(pts.shape[1], cap_rows=256, ctx=self.ctx)
            return slot.update(pts), True
        key = id(pts)
        hit = self._data_cache.get(key)
        if hit is not None and hit[0]() is pts:
            return hit[1], False
        dd = DeviceData(pts, ctx=self.ctx)
        try:
            self._data_cache[key] = (weakref.ref(pts, lambda _, k=key: self._data_cache.pop(k, None)), dd)
        except TypeError:
            pass
        return dd, False

    def forget(self, pts=None):
        """Drop the cached device copy of `pts` (or of everything).  Arrays of >= 4096 rows are uploaded
        once per projector and assumed unchanged afterwards; call this after editing one in place."""
        if pts is None:
            self._data_cache.clear()
            self._phi_cache.clear()
            return
        hit = self._data_cache.pop(id(pts), None)
        if hit is not None:
            for key in [k for k in self._phi_cache if k[0] == id(hit[1])]:
                del self._phi_cache[key]

    def _run(self, pts, model_id, params):
        dd, transient = self.device_data(pts)
        theta = self.model.theta_for_device(self.samples)
        if dd.shape[1] != self.model.data_width(theta.shape[1]):
            raise ValueError('data rows have %d columns, model expects %d for %d-dimensional samples'
                             % (dd.shape[1], self.model.data_width(theta.shape[1]), theta.shape[1]))
        params = np.ascontiguousarray(params, dtype=np.float64)
        S = int(theta.shape[0])
        if transient:
            cap, h = self._pool.acquire(dd.shape[0], S)
            N.call('bc_project', self.ctx.h, dd.h, int(model_id), _ptr(theta), S, _ptr(params), int(params.shape[0]),
                   0, C.byref(h))
            return DevicePhi(h, self.ctx, release=self._pool.releaser(cap, S))
        key = (id(dd), model_id, S)
        prev = self._phi_cache.get(key)
        h = C.c_void_p(prev.h.value) if prev is not None else C.c_void_p()
        N.call('bc_project', self.ctx.h, dd.h, int(model_id), _ptr(theta), S, _ptr(params), int(params.shape[0]),
               int(dd.row_offset), C.byref(h))
        if prev is not None:
            prev.refresh()
            return prev
        phi = DevicePhi(h, self.ctx)
        phi._data = dd
        self._phi_cache[key] = phi
        return phi


class DeviceProjector(_DeviceProjectorBase):
    """GPU counterpart of BlackBoxProjector: `project(pts)` returns a DevicePhi."""

    def project(self, pts, grad=False):
        if grad:
            raise NotImplementedError('x-gradients of the log-likelihood (BatchPSVI) are out of this path\'s scope')
        return self._run(pts, self.model.model_id, self.model.params())


class DeviceBetaProjector(_DeviceProjectorBase):
    """GPU counterpart of BetaBlackBoxProjector: `project_f(pts, beta)` returns a DevicePhi.
    Also offers `project` (plain log-likelihood), which the reference's class lacks."""

    def project(self, pts, grad=False):
        if grad:
            raise NotImplementedError
        return self._run(pts, self.model.model_id, self.model.params())

    def project_f(self, pts, beta, grad=False):
        bls = self._run(pts, self.model.beta_model_id, self.model.params(beta=beta))
        if not grad:
            return bls
        if self.model.beta_grad_model_id is None:
            raise ValueError('beta-gradient was requested but this model has none')
        g = self._run(pts, self.model.beta_grad_model_id, self.model.params(beta=beta))
        return bls, g
