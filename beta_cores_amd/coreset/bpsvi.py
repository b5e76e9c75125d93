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
        self.data = data
        self.ll_projector = ll_projector
        self.opt_itrs = opt_itrs
        self.n_subsample_opt = None if n_subsample_opt is None else min(data.shape[0], n_subsample_opt)   # bpsvi.py:11
        self.step_sched = step_sched
        self.mup = mup
        self.SigpInv = SigpInv
        self._dev_data = None
        if pin_data and self.n_subsample_opt is None and hasattr(ll_projector, 'pin') and isinstance(data, np.ndarray) \
                and data.ndim == 2 and data.shape[0] >= 4096:
            # every gradient re-projects ALL rows: keep them in HBM (read-only on the host while pinned)
            self._dev_data = ll_projector.pin(data)
            self._unpin = weakref.finalize(self, ll_projector.unpin, data)
        super().__init__(**kw)

    def _build(self, itrs, sz):
        """bpsvi.py:17-25: a fresh random initialisation of all `sz` points, then the optimisation (itrs is unused)"""
        init_idcs = np.random.choice(self.data.shape[0], size=sz, replace=False)
        self.pts = self.data[init_idcs]
        self.wts = self.data.shape[0] / sz * np.ones(sz)
        self.idcs = init_idcs
        self._optimize()

    def _get_projection(self, n_subsample, w, p):
        """bpsvi.py:27-43 -> (column sums of vecs, sum_scaling, sub_idcs, corevecs, pgrads).  Of `vecs` the gradient
        only uses `vecs.sum(axis=0)` (bpsvi.py:52), so that S-vector is what is returned (K2, on the device)."""
        self.ll_projector.update(w, p)
        if n_subsample is None:
            sub_idcs = None
            vecs = self.ll_projector.project(self._dev_data if self._dev_data is not None else self.data)
            sum_scaling = 1.
        else:
            sub_idcs = np.random.randint(self.data.shape[0], size=n_subsample)
            vecs = self.ll_projector.project(self.data[sub_idcs])
            sum_scaling = self.data.shape[0] / n_subsample
        if not isinstance(vecs, DevicePhi):
            vecs = DevicePhi.from_host(np.ascontiguousarray(vecs, dtype=np.float64), ctx=getattr(self.ll_projector, 'ctx', None))
        S = vecs.shape[1]
        vsum = vecs.sum(axis=0)
        if p.size > 0:
            corevecs, pgrads = self.ll_projector.project(p, grad=True)
            corevecs, pgrads = np.asarray(corevecs), np.asarray(pgrads)
        else:
            corevecs, pgrads = np.zeros((0, S)), np.zeros((0, S, p.shape[1]))
        return vsum, sum_scaling, sub_idcs, corevecs, pgrads

    def _optimize(self):
        """bpsvi.py:45-62"""
        sz = self.wts.shape[0]
        d = self.pts.shape[1]

        def grd(x):
            w = x[:sz]
            p = x[sz:].reshape((sz, d))
            vsum, sum_scaling, sub_idcs, corevecs, pgrads = self._get_projection(self.n_subsample_opt, w, p)
            resid = sum_scaling * vsum - w.dot(corevecs)
            wgrad = -corevecs.dot(resid) / corevecs.shape[1]
            ugrad = -(w[:, np.newaxis, np.newaxis] * pgrads * resid[np.newaxis, :, np.newaxis]).sum(axis=1) / corevecs.shape[1]
            return np.hstack((wgrad, ugrad.reshape(sz * d)))

        x0 = np.hstack((self.wts, self.pts.reshape(sz * d)))
        xf = partial_nn_opt(x0, grd, np.arange(sz), self.opt_itrs, step_sched=self.step_sched(sz))
        self.wts = xf[:sz].copy()
        self.pts = xf[sz:].reshape((sz, d)).copy()

    def error(self):
        return 0.          # bpsvi.py:64-65 ("TODO: implement KL estimate")
