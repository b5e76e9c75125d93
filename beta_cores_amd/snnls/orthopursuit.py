import numpy as np
from scipy.optimize import nnls

from .snnls import SparseNNLS, register_hooks


class OrthoPursuit(SparseNNLS):
    """Orthogonal matching pursuit (bayesiancoresets/snnls/orthopursuit.py:7-42).

    _select: K3 sweep in dot mode for the positive direction over all rows plus the
    negative direction over the active set (orthopursuit.py:17-35), both on the device.
    _reweight: w[f] = 1 then a full NNLS refit on the <= M active columns with SciPy on
    the host (orthopursuit.py:37-42) -- third-party arithmetic shared with the reference,
    O(S*M^2) work, not a data-parallel hot spot."""
    _alg = 'omp'
    _fusable = False

    def _select(self):
        return self._eng.select()

    def _reweight(self, f):
        idx, val = self._eng.sparse_weights()
        hit = np.flatnonzero(idx == f)
        if hit.size:
            val[hit[0]] = 1.
        else:
            idx = np.append(idx, np.int64(f))
            val = np.append(val, 1.)
        self._eng.set_sparse_weights(idx, val)          # device supplies the new column
        idx, val = self._eng.sparse_weights()
        cols = self._eng.columns()
        active = val > 0
        order = np.argsort(idx[active], kind='stable')    # A[:, nz_idcs] column order
        a_idx, a_cols = idx[active][order], cols[active][order]
        sol = nnls(a_cols.T, self.b, maxiter=100 * self.n_total)
        self._eng.set_sparse_weights(a_idx, sol[0], a_cols)


register_hooks('omp', OrthoPursuit)
