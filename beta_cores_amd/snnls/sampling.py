import numpy as np

from .snnls import SparseNNLS


class ImportanceSampling(SparseNNLS):
    """Norm-proportional importance sampling "solver" (bayesiancoresets/snnls/sampling.py:6-32).

    One device pass for the column norms; draws come from the global NumPy RNG like the
    reference (`np.random.choice`); weights are count-based and the monotone check is
    off (sampling.py:16)."""
    _alg = 'fw'
    _fusable = False

    def __init__(self, A, b, **kw):
        kw.setdefault('allow_zero_rows', True)      # the reference only guards ps > 0 (sampling.py:12-15)
        super().__init__(A, b, **kw)
        if self.comm is not None and self.comm.world > 1:
            raise NotImplementedError('sampling solvers draw from a host RNG over all rows: single rank only')
        self.cts = np.zeros(self.n_total)
        self.ps = self._eng.phi.norms()
        if np.any(self.ps > 0):
            self.ps /= self.ps.sum()
        else:
            self.ps = np.ones(self.n_total) / float(self.n_total)
        self.check_error_monotone = False

    def reset(self):
        super().reset()
        self.cts = np.zeros(self.n_total)

    def _select(self):
        return np.random.choice(self.ps.shape[0], p=self.ps)

    def _reweight(self, f):
        self.cts[f] += 1
        self.w = (self.cts / self.cts.sum()) / self.ps


class UniformSampling(ImportanceSampling):
    """sampling.py:34-37"""

    def __init__(self, A, b, **kw):
        super().__init__(A, b, **kw)
        self.ps = np.ones(self.n_total) / float(self.n_total)
