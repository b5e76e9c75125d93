from .snnls import SparseNNLS, register_hooks


class FrankWolfe(SparseNNLS):
    """Frank-Wolfe on the norm-weighted polytope (bayesiancoresets/snnls/frankwolfe.py:5-40).

    _select = K3 sweep in dot mode against the residual b - A.w (frankwolfe.py:15-17);
    _reweight = first-point rule / exact line search with the precision guard
    (frankwolfe.py:19-40) on the device.  The sum of column norms is global (all-reduced
    once when the rows are sharded)."""
    _alg = 'fw'
    _fusable = True

    def _select(self):
        return self._eng.select()

    def _reweight(self, f):
        self._eng.reweight(f)


register_hooks('fw', FrankWolfe)
