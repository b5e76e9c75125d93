"""NumPy restatement of the projector / coreset drivers of the hot path
(oracle; tests only).

  bayesiancoresets/coreset/projector.py:12-66   row-centred (beta-)projections
  bayesiancoresets/coreset/coreset.py:7-71      Coreset state + guards
  bayesiancoresets/coreset/hilbert.py:6-43      HilbertCoreset
  bayesiancoresets/coreset/bcores.py:8-156      BetaCoreset (all four tangent-space modes; learn_beta as pinned by F15)
  bayesiancoresets/coreset/sparsevi.py:8-139    SparseVI
  bayesiancoresets/coreset/bpsvi.py:6-65        BatchPSVICoreset (pseudo-points moved by x-gradients)
  bayesiancoresets/util/opt.py:36-77            nn_opt / partial_nn_opt
"""
import numpy as np
from .snnls_ref import RefGIGA, TOL


def project(loglik, pts, samples):
    """projector.py:23-26"""
    v = loglik(pts, samples)
    v -= v.mean(axis=1)[:, np.newaxis]
    return v


def project_grad(loglik, grad_loglik, pts, samples):
    """projector.py:23-32 with grad=True: the gradient tensor (M, S, D) is centred over its LAST axis (axis=2, the
    coordinates of x -- not the samples), exactly as the reference writes it"""
    v = project(loglik, pts, samples)
    g = grad_loglik(pts, samples)
    g -= g.mean(axis=2)[:, :, np.newaxis]
    return v, g


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
    """bcores.py:27-150 / sparsevi.py:27-136, all four tangent-space modes of _get_projection
    (bcores.py:42-61): full data, all groups, sub-sampled rows, sub-sampled groups (selection step
    only; with groups the gradient steps sub-sample ROWS, bcores.py:51-55).  Sub-sampling draws from
    the global NumPy RNG after the sampler has been called, like the reference (bcores.py:39, :53).

    All-zero rows of the tangent space are NOT dropped: the filter at bcores.py:67-68 /
    sparsevi.py:64-65 needs `select=True` with `groups is None`, and the ungrouped _select calls
    _get_projection with the default select=False (bcores.py:76, sparsevi.py:74).  Their correlation
    is 0/0 = NaN, np.argmax returns the first NaN and `corrs.max() > x` is False (pinned by F13).

    ``proj(pts, samples)`` is the row-centred projection with beta bound -- or, when ``beta`` is
    given, ``proj(pts, samples, beta)``; ``sampler(wts, pts)`` returns the S x D sample matrix for
    the current coreset (called once per projection, as ll_projector.update is).
    ``learn_beta`` (bcores.py:127-140) needs ``beta_grad(pts, samples, beta)`` (row-centred,
    projector.py:56-61); the method the reference calls there does not exist, its obvious reading
    (coreset rows' beta-gradient next to the usual tangent space) is pinned by F15."""

    def __init__(self, data, proj, sampler, opt_itrs, step_sched, groups=None, n_subsample_select=None,
                 n_subsample_opt=None, size_check_always=False, beta=None, learn_beta=False, beta_grad=None,
                 wts=None, idcs=None, pts=None):
        self.data = data
        self.proj = proj
        self.sampler = sampler
        self.opt_itrs = opt_itrs
        self.step_sched = step_sched
        self.groups = groups
        n = data.shape[0]
        self.n_subsample_select = None if n_subsample_select is None else min(n, n_subsample_select)
        self.n_subsample_opt = None if n_subsample_opt is None else min(n, n_subsample_opt)
        self.size_check_always = size_check_always
        self.beta = beta
        self.learn_beta = learn_beta
        self.beta_grad = beta_grad
        self.selected_groups = []
        self.wts = np.zeros(0) if wts is None else np.array(wts, dtype=float)
        self.idcs = np.zeros(0, dtype=np.int64) if idcs is None else np.array(idcs, dtype=np.int64)
        self.pts = np.zeros((0, data.shape[1])) if pts is None else np.array(pts, dtype=float)
        self.initialized = 0
        self.sel_trace = []

    def _p(self, pts, th, beta):
        return self.proj(pts, th) if self.beta is None else self.proj(pts, th, beta)

    def _tangent(self, n_subsample, w, beta, select=False, grad=False):
        """bcores.py:37-72; returns (vecs, sum_scaling, sub_idcs, group_idcs, corevecs[, betagrads])."""
        th = self.sampler(w, self.pts)                               # bcores.py:39
        group_idcs = None
        if n_subsample is None and self.groups is None:              # bcores.py:42-45
            sub_idcs = None
            vecs = self._p(self.data, th, beta)
            sum_scaling = 1.
        elif n_subsample is None and self.groups:                    # bcores.py:46-50
            group_idcs = list(range(len(self.groups)))
            sub_idcs = [i for g in self.groups for i in g]
            vecs = np.array([np.sum(self._p(self.data[g, :], th, beta), axis=0) for g in self.groups])
            sum_scaling = 1.
        elif n_subsample and (self.groups is None or not select):    # bcores.py:51-55
            sub_idcs = np.random.randint(self.data.shape[0], size=n_subsample)
            vecs = self._p(self.data[sub_idcs], th, beta)
            sum_scaling = self.data.shape[0] / n_subsample
        else:                                                        # bcores.py:56-61
            group_idcs = np.random.randint(len(self.groups), size=n_subsample)
            lst = [self.groups[i] for i in group_idcs]
            sub_idcs = [i for g in lst for i in g]
            vecs = np.array([np.sum(self._p(self.data[g, :], th, beta), axis=0) for g in lst])
            sum_scaling = len(self.groups) / n_subsample
        if self.pts.size > 0:                                        # bcores.py:63-66
            core = self._p(self.pts, th, beta)
        else:
            core = np.zeros((0, vecs.shape[1]))
        if not grad:
            return vecs, sum_scaling, sub_idcs, group_idcs, core
        bg = self.beta_grad(self.pts, th, beta) if self.pts.size > 0 else np.zeros((0, vecs.shape[1]))
        return vecs, sum_scaling, sub_idcs, group_idcs, core, bg

    def select(self):                                                # bcores.py:74-124
        grouped = self.groups is not None
        vecs, sum_scaling, sub_idcs, group_idcs, core = self._tangent(self.n_subsample_select, self.wts, self.beta,
                                                                      select=grouped)
        scale = 1. if (grouped and self.n_subsample_select is None) else sum_scaling
        resid = scale * vecs.sum(axis=0) - self.wts.dot(core)
        with np.errstate(invalid='ignore', divide='ignore'):
            corrs = vecs.dot(resid) / np.sqrt((vecs ** 2).sum(axis=1)) / vecs.shape[1]
            ccorrs = np.fabs(core.dot(resid) / np.sqrt((core ** 2).sum(axis=1))) / core.shape[1]
        f = -1
        if not grouped:
            if ccorrs.size == 0 or corrs.max() > ccorrs.max():
                f = int(sub_idcs[np.argmax(corrs)]) if sub_idcs is not None else int(np.argmax(corrs))
                if f not in self.idcs:
                    self.wts = np.append(self.wts, 0.)
                    self.idcs = np.append(self.idcs, f)
                    self.pts = np.vstack((self.pts, self.data[f][None, :]))
        else:
            max_core = ccorrs[self.initialized:].max() if ccorrs.shape[0] > self.initialized else -np.inf
            if ccorrs.size == 0 or corrs.max() > max_core:
                f = int(np.argmax(corrs)) if self.n_subsample_select is None else int(group_idcs[np.argmax(corrs)])
                if f not in self.selected_groups:
                    self.selected_groups.append(f)
                    g = self.groups[f]
                    self.wts = np.concatenate((self.wts, np.zeros(len(g))))
                    self.idcs = np.concatenate((self.idcs, np.asarray(g, dtype=np.int64)))
                    self.pts = np.vstack((self.pts, self.data[g, :]))
        self.sel_trace.append(f)

    def optimize(self):                                              # bcores.py:126-150
        if self.learn_beta:
            def grd(x):
                w, beta = x[:-1], x[-1]
                vecs, sum_scaling, _, _, core, bg = self._tangent(self.n_subsample_opt, w, beta, grad=True)
                resid = sum_scaling * vecs.sum(axis=0) - w.dot(core)
                wgrad = -core.dot(resid) / core.shape[1]
                betagrad = -10 ** (-5) * w.dot(bg.dot(resid)) / core.shape[1]
                return np.hstack((wgrad, betagrad))
            x0 = np.hstack((self.wts, np.asarray([self.beta])))
            xf = partial_nn_opt(x0, grd, np.arange(x0.shape[0]), self.opt_itrs, step_sched=self.step_sched)
            self.wts, self.beta = xf[:-1].copy(), xf[-1]       # the reference keeps a view here and cannot append afterwards
            return

        def grd(w):
            vecs, sum_scaling, _, _, core = self._tangent(self.n_subsample_opt, w, self.beta)
            resid = sum_scaling * vecs.sum(axis=0) - w.dot(core)
            return -core.dot(resid) / core.shape[1]
        self.wts = nn_opt(self.wts, grd, opt_itrs=self.opt_itrs, step_sched=self.step_sched)

    def build(self, itrs, sz=None):
        if sz is not None and (self.groups is None or self.size_check_always) and (self.wts > 0).sum() + itrs > sz:
            raise ValueError('itrs + size > sz')
        for _ in range(itrs):
            self.select()
            self.optimize()

    def get(self):
        keep = self.wts > 0
        return self.wts[keep], self.pts[keep, :], self.idcs[keep]


class RefBatchPSVI:
    """bpsvi.py:6-65.  `sampler(wts, pts)` -> Theta (projector.py:34-35); loglik / grad_loglik take (pts, samples)."""

    def __init__(self, data, loglik, grad_loglik, sampler, opt_itrs, n_subsample_opt=None,
                 step_sched=lambda m: lambda i: 1. / (1. + i), projector_draw=True):
        self.data = data
        self.loglik, self.grad_loglik, self.sampler = loglik, grad_loglik, sampler
        self.opt_itrs = opt_itrs
        self.n_subsample_opt = None if n_subsample_opt is None else min(data.shape[0], n_subsample_opt)   # bpsvi.py:11
        self.step_sched = step_sched
        self.wts, self.idcs, self.pts = np.array([]), np.array([], dtype=np.int64), np.zeros((0, data.shape[1]))
        # the projector draws its first Theta when IT is constructed (projector.py:17); a caller that already accounted
        # for that draw (one projector shared by several coresets, as in the driver) passes projector_draw=False
        self.samples = sampler(np.array([]), np.array([])) if projector_draw else None

    def build(self, itrs, sz):                                                                           # bpsvi.py:17-25
        init_idcs = np.random.choice(self.data.shape[0], size=sz, replace=False)
        self.pts = self.data[init_idcs]
        self.wts = self.data.shape[0] / sz * np.ones(sz)
        self.idcs = init_idcs
        self.optimize()

    def _get_projection(self, n_subsample, w, p):                                                        # bpsvi.py:27-43
        self.samples = self.sampler(w, p)
        if n_subsample is None:
            sub_idcs = None
            vecs = project(self.loglik, self.data, self.samples)
            sum_scaling = 1.
        else:
            sub_idcs = np.random.randint(self.data.shape[0], size=n_subsample)
            vecs = project(self.loglik, self.data[sub_idcs], self.samples)
            sum_scaling = self.data.shape[0] / n_subsample
        if p.size > 0:
            corevecs, pgrads = project_grad(self.loglik, self.grad_loglik, p, self.samples)
        else:
            corevecs, pgrads = np.zeros((0, vecs.shape[1])), np.zeros((0, vecs.shape[1], p.shape[1]))
        return vecs, sum_scaling, sub_idcs, corevecs, pgrads

    def optimize(self):                                                                                  # bpsvi.py:45-62
        sz = self.wts.shape[0]
        d = self.pts.shape[1]

        def grd(x):
            w = x[:sz]
            p = x[sz:].reshape((sz, d))
            vecs, sum_scaling, sub_idcs, corevecs, pgrads = self._get_projection(self.n_subsample_opt, w, p)
            resid = sum_scaling * vecs.sum(axis=0) - w.dot(corevecs)
            wgrad = -corevecs.dot(resid) / corevecs.shape[1]
            ugrad = -(w[:, np.newaxis, np.newaxis] * pgrads * resid[np.newaxis, :, np.newaxis]).sum(axis=1) / corevecs.shape[1]
            return np.hstack((wgrad, ugrad.reshape(sz * d)))

        x0 = np.hstack((self.wts, self.pts.reshape(sz * d)))
        xf = partial_nn_opt(x0, grd, np.arange(sz), self.opt_itrs, step_sched=self.step_sched(sz))
        self.wts = xf[:sz]
        self.pts = xf[sz:].reshape((sz, d))

    def get(self):                                                                                       # coreset.py:25-26
        return self.wts[self.wts > 0], self.pts[self.wts > 0, :], self.idcs[self.wts > 0]

