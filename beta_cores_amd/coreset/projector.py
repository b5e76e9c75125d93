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
from ..device import DeviceData, DevicePhi, _as_f64, _ptr, default_context
from ..util import numpy_bits


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
_PIPE_ROWS = 65536     # live host arrays from this size on are uploaded and projected in one pipelined call
_BIG_ROWS = 65536      # Phi buffers above this are not hoarded: at most one free buffer per S is kept


def _pool_shutdown(state):
    state['alive'] = False
    destroy = N.load().bc_phi_destroy
    for lst in state['free'].values():
        for _, h in lst:
            destroy(h)
    state['free'].clear()


class _PhiPool:
    """Phi buffers recycled instead of hipMalloc'ed / hipFree'd per projection.
    A handle is lent to exactly one DevicePhi wrapper; when that wrapper is garbage collected the
    handle comes back -- so two live results never alias: a solver that still holds the Phi of an
    earlier project() keeps its buffer, and the next project() gets another one (the reference
    returns a fresh array per call, projector.py:24).  Handles still on loan when the pool dies
    stay valid and are destroyed by their wrapper."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.state = {'alive': True, 'free': {}}     # shared with the release callbacks
        self._fin = weakref.finalize(self, _pool_shutdown, self.state)

    def acquire(self, n_rows, s):
        lst = self.state['free'].setdefault(s, [])
        best = None
        for i, (cap, h) in enumerate(lst):
            if cap >= n_rows and (best is None or cap < lst[best][0]):
                best = i
        if best is not None and (n_rows < _BIG_ROWS or lst[best][0] <= 2 * n_rows):
            return lst.pop(best)
        if n_rows < _BIG_ROWS:
            cap = max(256, 1 << int(np.ceil(np.log2(max(n_rows, 1)))))
        else:
            cap = ((n_rows + N.TILE_ROWS - 1) // N.TILE_ROWS) * N.TILE_ROWS      # large shards: no power-of-two slack
        h = C.c_void_p()
        N.call('bc_phi_create', self.ctx.h, int(cap), int(s), C.byref(h))
        return cap, h

    def releaser(self, cap, s):
        state = self.state

        def release(handle):
            if not state['alive']:
                N.load().bc_phi_destroy(handle)
                return
            lst = state['free'].setdefault(s, [])
            if cap >= _BIG_ROWS and any(c >= _BIG_ROWS for c, _ in lst):
                N.load().bc_phi_destroy(handle)      # one spare large buffer per S is enough for a gradient loop
                return
            lst.append((cap, handle))
        return release

    def clear(self):
        destroy = N.load().bc_phi_destroy
        for lst in self.state['free'].values():
            for _, h in lst:
                destroy(h)
        self.state['free'].clear()


# ---- host arrays pinned on the device.  While an ndarray is pinned its device copy is what gets projected, so an
# in-place edit of the array would go unseen; the array is therefore made read-only for as long as any pin holds it
# (NumPy then raises on assignment instead of the device silently serving stale rows).
_pin_guard = {}      # id(ndarray) -> entry [count, original writeable flag, weakref]; an entry is only trusted while its
                     # weakref still points at the array (ids are recycled once an array dies)


def _guard_acquire(arr):
    """Make `arr` read-only while pinned; returns the guard entry to hand to _guard_release."""
    key = id(arr)
    ent = _pin_guard.get(key)
    if ent is not None and ent[2]() is arr:
        ent[0] += 1
        return ent
    was = bool(arr.flags.writeable)
    try:
        arr.flags.writeable = False
    except ValueError:
        pass
    ent = [1, was, None]
    try:
        ent[2] = weakref.ref(arr, lambda _, k=key, e=ent: _pin_guard.pop(k, None) if _pin_guard.get(k) is e else None)
    except TypeError:
        ent[2] = lambda: arr
    _pin_guard[key] = ent
    return ent


def _guard_release(ent):
    """Drop one pin of the array this ENTRY guards (not whatever array carries the same id() today)."""
    if ent is None:
        return
    ent[0] -= 1
    if ent[0] <= 0:
        arr = ent[2]()
        if arr is not None:
            if ent[1]:
                try:
                    arr.flags.writeable = True
                except ValueError:
                    pass
            if _pin_guard.get(id(arr)) is ent:
                _pin_guard.pop(id(arr), None)


class _DeviceProjectorBase(Projector):
    def __init__(self, sampler, projection_dimension, model, ctx=None):
        self.projection_dimension = projection_dimension
        self.sampler = sampler
        self.model = model
        self.ctx = ctx or default_context()
        self._pins = {}            # id(ndarray) -> (weakref, DeviceData): arrays the caller pinned (see pin())
        self._slots = {}           # dz -> DeviceData slot for small transient inputs
        self._pool = _PhiPool(self.ctx)
        # constant rows whose value holds an np.exp: on a host whose NumPy is not the one the library restates they are
        # evaluated on the host (util/numpy_bits.py); `_key_cache`: the y's of the all-zero-feature rows of large inputs
        self._host_constants = numpy_bits.warn_if_constant_bits_differ(model) and getattr(model, 'host_constants', None) is not None
        self._key_cache = {}
        self.constant_rows_from_host = 0      # how many (y, value) pairs the last beta-projection handed to the kernel
        self.update(np.array([]), np.array([]))

    def update(self, wts, pts):
        self.samples = self.sampler(self.projection_dimension, wts, pts)

    # -- data residency.  Default: like projector.py:24, every project() reads the LIVE host array (it is uploaded
    # for that call).  A large array projected over and over (BetaCoreset's full-data gradient loop) can be pinned:
    # uploaded once, served from HBM afterwards, read-only on the host until unpinned.
    def pin(self, pts):
        """Keep a device copy of `pts` for later project() calls; returns the DeviceData.  `pts` becomes read-only
        (ndarray.flags.writeable = False) until unpin(pts) / forget(), or until the projector dies."""
        if isinstance(pts, DeviceData):
            return pts
        key = id(pts)
        hit = self._pins.get(key)
        if hit is not None and hit[0]() is pts:
            hit[3] += 1                     # pins nest: the copy goes when the last holder unpins
            return hit[1]
        if isinstance(pts, np.ndarray) and pts.base is not None:
            # a view: edits through its base array would go unseen by the device copy (the read-only flag only guards
            # the view itself), so views are not pinned -- they are uploaded per call like any live array
            raise ValueError('pin(): the array is a view of another array; pin the base array or pass a copy')
        arr = np.atleast_2d(pts)
        dd = DeviceData(arr, ctx=self.ctx)
        ent = _guard_acquire(pts)
        fin = weakref.finalize(self, _guard_release, ent)      # a dying projector lets go of its pins
        try:
            ref = weakref.ref(pts, lambda _, k=key, pins=self._pins: pins.pop(k, None))
        except TypeError:
            ref = lambda: pts
        self._pins[key] = [ref, dd, fin, 1, ent]
        return dd

    def unpin(self, pts):
        hit = self._pins.get(id(pts))
        if hit is None or hit[0]() is not pts:
            return
        hit[3] -= 1
        if hit[3] <= 0:
            self._pins.pop(id(pts), None)
            hit[2].detach()
            _guard_release(hit[4])

    def forget(self, pts=None):
        """Drop the pinned device copy of `pts` (or of everything) and the spare Phi buffers."""
        if pts is not None:
            self.unpin(pts)
            return
        for key in list(self._pins):
            hit = self._pins.pop(key)
            hit[2].detach()
            _guard_release(hit[4])
        self._pool.clear()

    def device_data(self, pts):
        """(DeviceData, transient): transient inputs live in a re-used upload slot that the next call overwrites."""
        if isinstance(pts, DeviceData):
            return pts, False
        hit = self._pins.get(id(pts))
        if hit is not None and hit[0]() is pts:
            return hit[1], False
        pts = np.atleast_2d(pts)
        if pts.shape[0] < _SMALL_ROWS:
            slot = self._slots.get(pts.shape[1])
            if slot is None:
                slot = self._slots[pts.shape[1]] = DeviceData.slot(pts.shape[1], cap_rows=256, ctx=self.ctx)
            return slot.update(pts), True
        return DeviceData(pts, ctx=self.ctx), False            # live array, uploaded for this call

    def _zero_feature_keys(self, pts, d):
        """Sorted unique y of the rows of `pts` whose d features are all zero (host array: NumPy; resident rows: one device scan)."""
        if isinstance(pts, DeviceData):
            hit = self._key_cache.get(id(pts))
            if hit is not None and hit[0]() is pts:
                return hit[1]
            cap = 65536
            out, n = np.empty(cap), C.c_int64()
            N.call('bc_data_zero_feature_keys', pts.h, int(d), cap, _ptr(out), C.byref(n))
            if n.value > cap:
                raise ValueError('%d data rows have all-zero features: more than the %d the host route for constant rows handles'
                                 % (n.value, cap))
            keys = np.unique(out[:n.value])
            try:
                self._key_cache[id(pts)] = (weakref.ref(pts, lambda _, k=id(pts), c=self._key_cache: c.pop(k, None)), keys)
            except TypeError:
                pass
            return keys
        arr = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        big = arr.shape[0] >= _SMALL_ROWS and isinstance(pts, np.ndarray)
        if big:
            hit = self._key_cache.get(id(pts))
            if hit is not None and hit[0]() is pts:
                return hit[1]
        keys = np.unique(arr[~arr[:, :d].any(axis=1), d])
        if big:
            try:
                self._key_cache[id(pts)] = (weakref.ref(pts, lambda _, k=id(pts), c=self._key_cache: c.pop(k, None)), keys)
            except TypeError:
                pass
        return keys

    def _stage_host_constants(self, pts, model_id, params, more=None):
        """Before a beta-projection on a host whose NumPy the library does not restate: the constants of the all-zero-feature
        rows of `pts` (and of `more`, a second row set projected by the same native call), evaluated here, go to the kernel
        (bc_ctx_set_constant_row_values).  No-op everywhere else."""
        if not self._host_constants or model_id != self.model.beta_model_id:
            return
        params = np.ascontiguousarray(params, dtype=np.float64)
        d = self.model.data_width(np.atleast_2d(self.samples).shape[1]) - 1
        keys = self._zero_feature_keys(pts, d)
        if more is not None:
            keys = np.union1d(keys, self._zero_feature_keys(more, d))
        vals = np.ascontiguousarray(self.model.host_constants(keys, params[1]), dtype=np.float64) if keys.size else np.zeros(0)
        self.constant_rows_from_host = int(keys.size)
        N.call('bc_ctx_set_constant_row_values', self.ctx.h, int(model_id), _ptr(params), int(params.shape[0]),
               _ptr(np.ascontiguousarray(keys)), _ptr(vals), int(keys.size))

    def _run_from_host(self, pts, model_id, params, keep=False):
        """project(ndarray) for a LARGE live host array: upload and K1 pipelined in one native call (bc_project_from_host;
        Phi, norms and column sums are the resident path's bit for bit).  Returns (DevicePhi, DeviceData of the uploaded
        rows); the latter is dropped by the caller unless it wants the rows to stay in HBM."""
        self._stage_host_constants(pts, model_id, params)
        pts = _as_f64(np.atleast_2d(pts), 'data')
        theta = self.model.theta_for_device(self.samples)
        if pts.shape[1] != self.model.data_width(theta.shape[1]):
            raise ValueError('data rows have %d columns, model expects %d for %d-dimensional samples'
                             % (pts.shape[1], self.model.data_width(theta.shape[1]), theta.shape[1]))
        params = np.ascontiguousarray(params, dtype=np.float64)
        S = int(theta.shape[0])
        cap, h = self._pool.acquire(pts.shape[0], S)
        dh = C.c_void_p()
        try:
            N.call('bc_project_from_host', self.ctx.h, _ptr(pts), int(pts.shape[0]), int(pts.shape[1]), int(model_id), _ptr(theta), S,
                   _ptr(params), int(params.shape[0]), 0, C.byref(dh), C.byref(h))
        except Exception:
            self._pool.releaser(cap, S)(h)
            raise
        dd = DeviceData._adopt(dh, pts.shape, self.ctx)
        return DevicePhi(h, self.ctx, release=self._pool.releaser(cap, S)), dd

    def _run(self, pts, model_id, params):
        if isinstance(pts, np.ndarray) and pts.ndim == 2 and pts.shape[0] >= _PIPE_ROWS and self._pins.get(id(pts)) is None:
            return self._run_from_host(pts, model_id, params)[0]
        self._stage_host_constants(pts, model_id, params)
        dd, transient = self.device_data(pts)
        theta = self.model.theta_for_device(self.samples)
        if dd.shape[1] != self.model.data_width(theta.shape[1]):
            raise ValueError('data rows have %d columns, model expects %d for %d-dimensional samples'
                             % (dd.shape[1], self.model.data_width(theta.shape[1]), theta.shape[1]))
        params = np.ascontiguousarray(params, dtype=np.float64)
        S = int(theta.shape[0])
        cap, h = self._pool.acquire(dd.shape[0], S)
        try:
            N.call('bc_project', self.ctx.h, dd.h, int(model_id), _ptr(theta), S, _ptr(params), int(params.shape[0]),
                   0 if transient else int(dd.row_offset), C.byref(h))
        except Exception:
            self._pool.releaser(cap, S)(h)
            raise
        return DevicePhi(h, self.ctx, release=self._pool.releaser(cap, S))

    # -- the store-free paths of the gradient loop (bcores.py:141-146, sparsevi.py:129-134): of the N x S projection of
    # the data rows only the S column sums are needed there, so K1 keeps its column partials and writes no Phi
    def _ids(self, beta):
        if beta is None:
            return self.model.model_id, self.model.params()
        return self.model.beta_model_id, self.model.params(beta=beta)

    def _theta_checked(self, dd):
        theta = self.model.theta_for_device(self.samples)
        if dd.shape[1] != self.model.data_width(theta.shape[1]):
            raise ValueError('data rows have %d columns, model expects %d for %d-dimensional samples'
                             % (dd.shape[1], self.model.data_width(theta.shape[1]), theta.shape[1]))
        return theta

    def colsum(self, pts, beta=None, comm=None):
        """`project(pts).sum(axis=0)` (beta None) / `project_f(pts, beta).sum(axis=0)` without materialising the
        projection: bc_project_colsum, bit-identical to the column sums of the materialised Phi.  `comm`: a native
        communicator handle (ShardComm.native_comm) -- the sum then runs over all ranks' shards.  Returns None when the
        store-free kernel does not cover the request (S > 256); the caller projects and sums instead."""
        dd, _ = self.device_data(pts)
        theta = self._theta_checked(dd)
        S = int(theta.shape[0])
        if S > 256:
            return None
        model_id, params = self._ids(beta)
        params = np.ascontiguousarray(params, dtype=np.float64)
        self._stage_host_constants(pts, model_id, params)
        out = np.empty(S)
        N.call('bc_project_colsum', self.ctx.h, dd.h, int(model_id), _ptr(theta), S, _ptr(params), int(params.shape[0]),
               comm, _ptr(out))
        return out

    def vi_gradient(self, data, core_pts, w, sum_scaling=1., beta=None, comm=None, want_resid=False, overlap=None):
        """One gradient of the greedy-VI weight optimisation in one native call (bc_vi_gradient): with the current
        samples, -corevecs.dot(sum_scaling * vecs.sum(axis=0) - w.dot(corevecs)) / S for vecs = projection of `data`
        (resident DeviceData, pinned array, or a live array that is uploaded for this call like project() does) and
        corevecs = projection of `core_pts`.  None if not covered (S > 256, no coreset rows, or a coreset too large
        for the staging area).  `overlap`: a callable run on the host between the enqueue and the wait (bc_vi_gradient_begin /
        _end), i.e. beside the GPU -- the samplers' `prefetch` (drawing the next sample matrix's normals) goes there."""
        dd, _ = self.device_data(data)
        core = np.ascontiguousarray(np.atleast_2d(core_pts), dtype=np.float64)
        m = int(core.shape[0])
        theta = self._theta_checked(dd)
        S = int(theta.shape[0])
        if S > 256 or m == 0 or m * (core.shape[1] + 1) > 60000:
            return None
        if core.shape[1] != dd.shape[1]:
            raise ValueError('coreset rows have %d columns, data rows %d' % (core.shape[1], dd.shape[1]))
        w = np.ascontiguousarray(w, dtype=np.float64)
        if w.shape != (m,):
            raise ValueError('one weight per coreset row')
        model_id, params = self._ids(beta)
        params = np.ascontiguousarray(params, dtype=np.float64)
        self._stage_host_constants(data, model_id, params, more=core)
        grad = np.empty(m)
        resid = np.empty(S) if want_resid else None
        N.call('bc_vi_gradient_begin', self.ctx.h, dd.h, _ptr(core), m, int(model_id), _ptr(theta), S, _ptr(params),
               int(params.shape[0]), _ptr(w), float(sum_scaling), comm)
        try:
            if overlap is not None:
                overlap()
        finally:
            N.call('bc_vi_gradient_end', self.ctx.h, _ptr(grad), _ptr(resid) if want_resid else None)
        return (grad, resid) if want_resid else grad


class DeviceProjector(_DeviceProjectorBase):
    """GPU counterpart of BlackBoxProjector: `project(pts)` returns a DevicePhi."""

    def project(self, pts, grad=False):
        """projector.py:23-32.  With `grad` also the x-gradient tensor of the log-likelihood at `pts` (M x S x W host
        array, centred over its last axis as the reference does it): bc_project_grad_x, meant for the coreset's
        pseudo-points (BatchPSVICoreset, bpsvi.py:39-40)."""
        lls = self._run(pts, self.model.model_id, self.model.params())
        if not grad:
            return lls
        return lls, self._grad_x(pts)

    def _grad_x(self, pts):
        if not self.model.has_grad_x:
            raise ValueError('grad_loglikelihood was requested but this model has none')
        dd, _ = self.device_data(pts)
        theta = self.model.theta_for_device(self.samples)
        params = np.ascontiguousarray(self.model.params(), dtype=np.float64)
        out = np.empty((dd.shape[0], int(theta.shape[0]), dd.shape[1]))
        N.call('bc_project_grad_x', self.ctx.h, dd.h, int(self.model.model_id), _ptr(theta), int(theta.shape[0]),
               _ptr(params), int(params.shape[0]), _ptr(out))
        return out


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
