"""SparseVICoreset (bayesiancoresets/coreset/sparsevi.py:8-139): the greedy variational coreset on the plain
log-likelihood projection; the shared recipe lives in greedy_vi.GreedyVICoreset."""
from .greedy_vi import GreedyVICoreset


class SparseVICoreset(GreedyVICoreset):
    """SparseVI (sparsevi.py:8-139): the same greedy loop with the plain log-likelihood projection."""
    _size_check_always = True

    def __init__(self, data, ll_projector, n_subsample_select=None, n_subsample_opt=None, opt_itrs=100,
                 step_sched=lambda i: 1. / (1. + i), mup=None, SigpInv=None, groups=None, selected_groups=None,
                 initialized=False, enforce_new=False, **kw):
        self.enforce_new = enforce_new
        super().__init__(data, ll_projector, n_subsample_select=n_subsample_select, n_subsample_opt=n_subsample_opt,
                         opt_itrs=opt_itrs, step_sched=step_sched, mup=mup, SigpInv=SigpInv, groups=groups,
                         selected_groups=selected_groups, initialized=initialized, **kw)

    def _beta(self):
        return None

    def _proj(self, pts, beta):
        return self.ll_projector.project(pts)
