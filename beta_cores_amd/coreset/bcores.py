"""BetaCoreset: the beta-Cores construction (bayesiancoresets/coreset/bcores.py:8-156) on the device kernels.

The greedy recipe (tangent space -> residual -> best correlation -> projected ADAM on the weights) is shared with
SparseVI and lives in greedy_vi.GreedyVICoreset; this module holds what is specific to the beta-divergence
variant: the beta-likelihood projection, the optional learning of beta and the 4-tuple returned by get()."""
import numpy as np

from ..util.opt import partial_nn_opt
from .greedy_vi import GreedyVICoreset


class BetaCoreset(GreedyVICoreset):
    """beta-Cores: robust coreset via the beta-divergence projection (bcores.py:8-156).

    `ll_projector` must offer `project_f(pts, beta[, grad])` (BetaBlackBoxProjector or
    DeviceBetaProjector).  `learn_beta=True` (the constructor default, bcores.py:11) also optimises
    beta and needs a projector with a beta-gradient (the Gaussian-location model has one,
    gaussian.py:46-62); see _optimize for how the reference's broken branch is read."""
    _size_check_always = False

    def __init__(self, data, ll_projector, n_subsample_select=None, n_subsample_opt=None, opt_itrs=100,
                 step_sched=lambda i: 1. / (1. + i), mup=None, SigpInv=None, beta=.5, learn_beta=True, groups=None,
                 selected_groups=None, initialized=False, **kw):
        self.beta = beta
        self.learn_beta = learn_beta
        super().__init__(data, ll_projector, n_subsample_select=n_subsample_select, n_subsample_opt=n_subsample_opt,
                         opt_itrs=opt_itrs, step_sched=step_sched, mup=mup, SigpInv=SigpInv, groups=groups,
                         selected_groups=selected_groups, initialized=initialized, **kw)

    def _beta(self):
        return self.beta

    def _proj(self, pts, beta):
        return self.ll_projector.project_f(pts, beta)

    def _fused_beta(self, beta):
        return beta

    def _optimize(self):
        """bcores.py:126-150.  learn_beta=True: projected ADAM over (w, beta) jointly, the beta-gradient
        scaled by 1e-5 (bcores.py:128-140).  The method the reference calls for the tangent space,
        `_get_projection_ii`, is not defined anywhere in its tree; it is read as `_get_projection` plus the
        beta-gradient of the coreset rows from project_f(..., grad=True) -- what `betagrads.dot(resid)` being
        M-long requires -- and that reading is pinned by golden F15 (generated from the reference with exactly that
        method attached).  A second reference quirk is fenced rather than reproduced: `self.wts = xf[:-1]` leaves a
        view, so the reference's next append dies in ndarray.resize (bcores.py:85); here the weights are copied."""
        if not self.learn_beta:
            return super()._optimize()

        def grd(x):
            w, beta = x[:-1], x[-1]
            vecs, sum_scaling, _, _, corevecs, betagrads = self._tangent(self.n_subsample_opt, w, self.pts, beta, grad=True)
            resid = sum_scaling * self._colsum(vecs) - w.dot(corevecs)
            wgrad = -corevecs.dot(resid) / corevecs.shape[1]
            betagrad = -10 ** (-5) * w.dot(betagrads.dot(resid)) / corevecs.shape[1]
            return np.hstack((wgrad, betagrad))
        x0 = np.hstack((self.wts, np.asarray([self.beta])))
        xf = partial_nn_opt(x0, grd, np.arange(x0.shape[0]), self.opt_itrs, step_sched=self.step_sched)
        self.wts = xf[:-1].copy()
        self.beta = xf[-1]

    def get(self):
        keep = self.wts > 0
        return self.wts[keep], self.pts[keep, :], self.idcs[keep], self.beta
