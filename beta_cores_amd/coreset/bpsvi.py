"""BatchPSVICoreset (bayesiancoresets/coreset/bpsvi.py:6-65): a pseudo-coreset -- `sz` points initialised on data
rows and then MOVED, together with their weights, by `opt_itrs` projected-ADAM steps.

Every gradient needs (bpsvi.py:27-57)
  * the projection of the data (all rows, or `n_subsample_opt` random ones) and its column sums: K1 + K2 on the GPU
    with a DeviceProjector, only the S-vector comes back;
  * the projection of the `sz` pseudo-points and the x-gradient tensor of their log-likelihoods (sz x S x W,
    projector.py:27-32): bc_project / bc_project_grad_x on the points, which live on the host between steps because
    the optimiser (util/opt.py partial_nn_opt) updates them there.
With a BlackBoxProjector the reference's NumPy expressions run on whatever the callables return and the N-row column
sum still runs on the GPU."""
import weakref

import numpy as np

from ..device import DevicePhi
from ..util.opt import partial_nn_opt
from .coreset import Coreset


class BatchPSVICoreset(Coreset):
    def __init__(self, data, ll_projector, opt_itrs, n_subsample_opt=None, step_sched=lambda m: lambda i: 1. / (1. + i),
                 mup=None, Zmean=None, SigpInv=None, diagnostics=False, comm=None, pin_data=True, **kw):
        if comm is not None and comm.world > 1:
            raise NotImplementedError('BatchPSVICoreset initialises its points from rows of the whole data set; '
                                      'row shards are not supported')
        self.data, self.ll_projector = data, ll_projector
        self.opt_itrs, self.step_sched = opt_itrs, step_sched
        self.mup, self.SigpInv = mup, SigpInv
        n = data.shape[0]
        self.n_subsample_opt = n_subsample_opt if n_subsample_opt is None else min(n, n_subsample_opt)   # bpsvi.py:11
        self._resident = None
        wants_pin = pin_data and self.n_subsample_opt is None and hasattr(ll_projector, 'pin')
        if wants_pin and isinstance(data, np.ndarray) and data.base is None and data.ndim == 2 and n >= 4096:
            # every gradient re-projects ALL rows: keep them in HBM (read-only on the host while pinned)
            self._resident = ll_projector.pin(data)
            self._unpin = weakref.finalize(self, ll_projector.unpin, data)
        elif wants_pin and isinstance(data, np.ndarray) and data.base is not None and data.ndim == 2 and n >= 4096:
            from .greedy_vi import resident_copy_of_view
            self._resident = resident_copy_of_view(data, ll_projector)       # a view: copied once, with a warning
        super().__init__(**kw)

    # ---- bpsvi.py:17-25: a fresh random initialisation of all `sz` points, then the optimisation (itrs is unused)
    def _build(self, itrs, sz):
        n = self.data.shape[0]
        self.idcs = np.random.choice(n, size=sz, replace=False)
        self.pts = self.data[self.idcs]
        self.wts = np.full(sz, n / sz)
        self._optimize()

    # ---- bpsvi.py:27-43.  Of the data's projection the gradient only uses its column sums (bpsvi.py:52), so that
    # S-vector (K2, on the device) is what this returns, already scaled to the whole data set.
    def _data_term(self):
        m = self.n_subsample_opt
        if m is None:
            rows, scale = (self._resident if self._resident is not None else self.data), 1.
        else:
            rows, scale = self.data[np.random.randint(self.data.shape[0], size=m)], self.data.shape[0] / m
        if hasattr(self.ll_projector, 'colsum'):        # device projector: store-free K1, the bits of project().sum(axis=0)
            b = self.ll_projector.colsum(rows)
            if b is not None:
                return scale * b
        vecs = self.ll_projector.project(rows)
        if not isinstance(vecs, DevicePhi):            # black-box projector: a host array, reduced on the device
            ctx = getattr(self.ll_projector, 'ctx', None)
            vecs = DevicePhi.from_host(np.ascontiguousarray(vecs, dtype=np.float64), ctx=ctx)
        return scale * vecs.sum(axis=0)

    def _point_terms(self, p, n_samples):
        if p.size == 0:
            return np.zeros((0, n_samples)), np.zeros((0, n_samples, p.shape[1]))
        lls, glls = self.ll_projector.project(p, grad=True)
        return np.asarray(lls), np.asarray(glls)

    def _gradient(self, x, sz, d):
        """bpsvi.py:47-57: d/d(weights, points) of the squared tangent-space residual, estimated over the S samples"""
        w, p = x[:sz], x[sz:].reshape(sz, d)
        self.ll_projector.update(w, p)                  # new Theta first (bpsvi.py:29), then the RNG draws of _data_term
        target = self._data_term()
        corevecs, pgrads = self._point_terms(p, target.shape[0])
        resid = target - w.dot(corevecs)
        S = corevecs.shape[1]
        g_w = -corevecs.dot(resid) / S
        g_p = -(w[:, np.newaxis, np.newaxis] * pgrads * resid[np.newaxis, :, np.newaxis]).sum(axis=1) / S
        return np.concatenate((g_w, g_p.ravel()))

    def _optimize(self):
        sz, d = self.wts.shape[0], self.pts.shape[1]
        x0 = np.concatenate((self.wts, self.pts.ravel()))
        x = partial_nn_opt(x0, lambda v: self._gradient(v, sz, d), np.arange(sz), self.opt_itrs, step_sched=self.step_sched(sz))
        self.wts, self.pts = x[:sz].copy(), x[sz:].reshape(sz, d).copy()

    def error(self):
        return 0.          # bpsvi.py:64-65 leaves the KL estimate unimplemented
