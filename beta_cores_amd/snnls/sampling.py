"""Sampling "solvers": the importance / uniform sampling baselines behind the SparseNNLS protocol.

Behaviour of bayesiancoresets/snnls/sampling.py:6-37: draw one row index per iteration from a
fixed distribution (proportional to the row norms of Phi, or uniform), keep counts, and set
w = (counts / total) / p.  The draws use the GLOBAL NumPy RNG (`np.random.choice`) so seeded runs
reproduce the reference's sequence; the monotone-error guard is switched off.  The only O(N S)
work, the row norms, comes from the device copy of Phi.
"""
import numpy as np

from .snnls import SparseNNLS


def _normalised(p):
    total = p.sum()
    return p / total if np.any(p > 0) else np.full(p.shape[0], 1. / float(p.shape[0]))


class ImportanceSampling(SparseNNLS):
    _alg = 'fw'          # any engine: only its norms, error() and the sparse weight list are used
    _fusable = False

    def __init__(self, A, b, **kw):
        kw.setdefault('allow_zero_rows', True)      # zero-norm rows simply get probability 0 (sampling.py:12-15)
        super().__init__(A, b, **kw)
        if self.comm is not None and self.comm.world > 1:
            raise NotImplementedError('sampling solvers draw from a host RNG over all rows: single rank only')
        self.check_error_monotone = False           # sampling.py:16
        self.cts = np.zeros(self.n_total)
        self.ps = self._probabilities()

    def _probabilities(self):
        return _normalised(self._eng.phi.norms())

    def reset(self):
        super().reset()
        self.cts = np.zeros(self.n_total)

    def _select(self):
        return np.random.choice(self.ps.shape[0], p=self.ps)

    def _reweight(self, f):
        self.cts[f] += 1
        self.w = (self.cts / self.cts.sum()) / self.ps


class UniformSampling(ImportanceSampling):
    def _probabilities(self):
        return np.full(self.n_total, 1. / float(self.n_total))
