import numpy as np

from .snnls import SparseNNLS, register_hooks


class GIGA(SparseNNLS):
    """Greedy iterative geodesic ascent (bayesiancoresets/snnls/giga.py:6-64).

    _select = K3 sweep in GIGA mode (giga.py:20-38 fused: normalised two-column score
    product, validity mask, sqrt, divide, argmax with first-index tie-break);
    _reweight = the closed-form geodesic line search (giga.py:40-64) on the device.
    Zero-norm columns raise ValueError like giga.py:11-12; ||b|| == 0 raises
    NumericalPrecisionError like giga.py:16-17."""
    _alg = 'giga'
    _fusable = True

    def __init__(self, A, b, **kw):
        super().__init__(A, b, **kw)
        self.bnorm = float(np.sqrt((self.b ** 2).sum()))
        self.bn = self.b / self.bnorm

    def _select(self):
        return self._eng.select()

    def _reweight(self, f):
        self._eng.reweight(f)


register_hooks('giga', GIGA)
