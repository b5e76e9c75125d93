"""NumPy restatement of the projector / coreset drivers of the hot path
(oracle; tests only).

  bayesiancoresets/coreset/projector.py:12-66   row-centred (beta-)projections
  bayesiancoresets/coreset/coreset.py:7-71      Coreset state + guards
  bayesiancoresets/coreset/hilbert.py:6-43      HilbertCoreset
  bayesiancoresets/coreset/bcores.py:8-156      BetaCoreset (ungrouped, learn_beta=False)
  bayesiancoresets/coreset/sparsevi.py:8-139    SparseVI (ungrouped)
  bayesiancoresets/util/opt.py:36-77            nn_opt / partial_nn_opt
"""
import numpy as np
from .snnls_ref import RefGIGA, TOL


def project(loglik, pts, samples):
    """projector.py:23-26"""
    v = loglik(pts, samples)
    v -= v.mean(axis=1)[:, np.newaxis]
    return v


def project_f(beta_lik, pts, samples, beta):
    """projector.py:51-55"""
    v = beta_lik(pts, samples, beta)
    v -= v.mean(axis=1)[:, np.newaxis]
    return v


def nn_opt(x0, grd, opt_itrs=1000, step_sched=lambda i: 1. / (i + 1), b1=0.9, b2=0.999, eps=1e-8):
    """opt.py:36-54"""
    x = x0.copy()
    m1 = np.zeros(x.shape[0])
    m2 = np.zeros(x.shape[0])
    for i in range(opt_itrs):
        g = grd(x)
        m1 = b1 * m1 + (1. - b1) * g
        m2 = b2 * m2 + (1. - b2) * g ** 2
        upd = step_sched(i) * m1 / (1. - b1 ** (i + 1)) / (eps + np.sqrt(m2 / (1. - b2 ** (i + 1))))
        x -= upd
        x = np.maximum(x, 0.)
    return x


def partial_nn_opt(x0, grd, nn_idcs, opt_itrs=1000, step_sched=lambda i: 1. / (i + 1), b1=0.9, b2=0.999, eps=1e-8):
    """opt.py:56-77"""
    x = x0.copy()
    m1 = np.zeros(x.shape[0])
    m2 = np.zeros(x.shape[0])
    for i in range(opt_itrs):
        g = grd(x)
        m1 = b1 * m1 + (1. - b1) * g
        m2 = b2 * m2 + (1. - b2) * g ** 2
        upd = step_sched(i) * m1 / (1. - b1 ** (i + 1)) / (eps + np.sqrt(m2 / (1. - b2 ** (i + 1))))
        x -= upd
        x[nn_idcs] = np.maximum(x[nn_idcs], 0.)
    return x


class RefHilbert:
    """hilbert.py:6-43 with a fixed sample matrix (no sub-sampling RNG)."""

    def __init__(self, data, loglik, samples, solver=RefGIGA):
        vecs = project(loglik, data, samples)
        vecs = vecs[np.sqrt((vecs ** 2).sum(axis=1)) > 0., :]      # hilbert.py:16
        self.vecs = vecs
        self.solver = solver(vecs.T, vecs.sum(axis=0))             # hilbert.py:17
        self.data = data
        self.wts = np.array([])
        self.idcs = np.array([], dtype=np.int64)
        self.pts = np.array([])

    def build(self, itrs, sz):
        if self.solver.hit_limit:
            return
        if sz < (self.wts > 0).sum():
            raise ValueError('cannot shrink')
        if self.solver.size() + itrs > sz:
            raise ValueError('itrs + size > sz')
        self.solver.build(itrs)
        self._pull()

    def optimize(self):
        self.solver.optimize()
        self._pull()

    def _pull(self):
        w = self.solver.weights()
        self.wts = w[w > 0]
        self.idcs = np.where(w > 0)[0]
        self.pts = self.data[self.idcs]

    def get(self):
        return self.wts, self.pts, self.idcs

    def error(self):
        return self.solver.error()


class RefGreedyVI:
    """bcores.py:27-150 / sparsevi.py:27-136, full-data (n_subsample=None) so that no RNG
    enters besides the caller's sampler; ungrouped (bcores.py:75-90) or grouped
    (bcores.py:46-50, 91-123: per-group sums of the projection rows are scored, a whole group
    of rows joins the coreset at once).

    ``proj(pts, samples)`` is the row-centred projection (beta already bound);
    ``sampler(wts, pts)`` returns the S x D sample matrix for the current
    coreset (called once per projection, as ll_projector.update is)."""

    def __init__(self, data, proj, sampler, opt_itrs, step_sched, groups=None):
        self.data = data
        self.proj = proj
        self.sampler = sampler
        self.opt_itrs = opt_itrs
        self.step_sched = step_sched
        self.groups = groups
        self.selected_groups = []
        self.wts = np.zeros(0)
        self.idcs = np.zeros(0, dtype=np.int64)
        self.pts = np.zeros((0, data.shape[1]))
        self.sel_trace = []

    def _tangent(self, w):
        th = self.sampler(w, self.pts)                               # bcores.py:39
        if self.groups is None:
            vecs = self.proj(self.data, th)                          # bcores.py:44
        else:                                                        # bcores.py:46-50
            vecs = np.array([np.sum(self.proj(self.data[g, :], th), axis=0) for g in self.groups])
        core = self.proj(self.pts, th) if self.pts.size > 0 else np.zeros((0, vecs.shape[1]))
        return vecs, core

    def select(self):                                                # bcores.py:74-124
        vecs, core = self._tangent(self.wts)
        if self.groups is None:
            vecs = vecs[~np.all(vecs == 0., axis=1)]
        resid = 1. * vecs.sum(axis=0) - self.wts.dot(core)
        corrs = vecs.dot(resid) / np.sqrt((vecs ** 2).sum(axis=1)) / vecs.shape[1]
        ccorrs = np.fabs(core.dot(resid) / np.sqrt((core ** 2).sum(axis=1))) / core.shape[1]
        f = -1
        if ccorrs.size == 0 or corrs.max() > ccorrs.max():
            f = int(np.argmax(corrs))
            if self.groups is None:
                if f not in self.idcs:
                    self.wts = np.append(self.wts, 0.)
                    self.idcs = np.append(self.idcs, f)
                    self.pts = np.vstack((self.pts, self.data[f][None, :]))
            elif f not in self.selected_groups:
                self.selected_groups.append(f)
                g = self.groups[f]
                self.wts = np.concatenate((self.wts, np.zeros(len(g))))
                self.idcs = np.concatenate((self.idcs, np.asarray(g, dtype=np.int64)))
                self.pts = np.vstack((self.pts, self.data[g, :]))
        self.sel_trace.append(f)

    def optimize(self):                                              # bcores.py:141-150
        def grd(w):
            vecs, core = self._tangent(w)
            resid = 1. * vecs.sum(axis=0) - w.dot(core)
            return -core.dot(resid) / core.shape[1]
        self.wts = nn_opt(self.wts, grd, opt_itrs=self.opt_itrs, step_sched=self.step_sched)

    def build(self, itrs):
        for _ in range(itrs):
            self.select()
            self.optimize()

    def get(self):
        keep = self.wts > 0
        return self.wts[keep], self.pts[keep, :], self.idcs[keep]
