"""SparseNNLS: greedy sparse non-negative least squares with the state on the GPU.

Same class protocol as bayesiancoresets/snnls/snnls.py:8-106 -- ctor `Alg(A, b)`,
`build(itrs)`, `weights()`, `size()`, `error()`, `reset()`, `optimize()`, and the
subclass hooks `_select() -> int`, `_reweight(f)`, `_stabilize()` -- so a
HilbertCoreset (or user code) can plug these classes in unchanged.  What differs is
where the work happens: A (= Phi^T) lives in HBM in the tiled layout, w is kept
sparse on the device, and for GIGA / FrankWolfe `build()` runs the whole guarded
loop on the device (no host round trip per iteration).  `build_stepwise()` is the
reference's host loop over the same device primitives, kept for subclasses that
override the hooks and for OrthoPursuit (host NNLS refit).
"""
import numpy as np
from scipy.optimize import nnls

from .. import util
from ..util.errors import NumericalPrecisionError
from ..util.log import make_logger
from .engine import HipEngine, phi_from_A


class SparseNNLS(object):
    _alg = None          # 'giga' | 'fw' | 'omp' for device-backed subclasses
    _fusable = False     # True when build() may run the fused device loop

    def __init__(self, A, b, check_error_monotone=True, comm=None, ctx=None, row_offset=None,
                 allow_zero_rows=False, engine=None):
        self.alg_name, self.log = make_logger(self)
        self.A = A
        self.b = np.asarray(b, dtype=np.float64)
        self.check_error_monotone = check_error_monotone
        self.comm = comm
        self._limit = False
        if engine is not None:
            self._eng = engine          # injected by tests (protocol checks without a GPU)
        else:
            if row_offset is None:
                row_offset = 0
            phi = phi_from_A(A, ctx=ctx, row_offset=row_offset)
            self._eng = HipEngine(phi, self.b, self._alg or 'fw', allow_zero_rows=allow_zero_rows, comm=comm,
                                  tol=util.TOL)
        self.n_local = self._eng.n_local
        self.row_offset = self._eng.row_offset
        self.n_total = A.shape[1] if comm is None or comm.world == 1 else comm.total_rows(self.n_local)

    # ---- state mirrors (device is authoritative)
    @property
    def reached_numeric_limit(self):
        return self._limit

    @reached_numeric_limit.setter
    def reached_numeric_limit(self, flag):
        self._limit = bool(flag)
        self._eng.set_limit(self._limit)

    @property
    def w(self):
        """Dense weight vector over ALL rows (global indexing), materialised on demand."""
        idx, val = self._eng.sparse_weights()
        w = np.zeros(self.n_total)
        w[idx] = val
        return w

    @w.setter
    def w(self, dense):
        dense = np.asarray(dense, dtype=np.float64)
        idx = np.flatnonzero(dense)
        self._eng.set_sparse_weights(idx, dense[idx])

    def sparse_weights(self):
        """(global indices ascending, weights) of the strictly positive entries."""
        idx, val = self._eng.sparse_weights()
        keep = val > 0
        idx, val = idx[keep], val[keep]
        order = np.argsort(idx, kind='stable')
        return idx[order], val[order]

    # ---- snnls.py:18-29
    def reset(self):
        self._eng.reset()
        self._limit = False

    def size(self):
        return self._eng.size()

    def weights(self):
        return self.w

    def error(self):
        return self._eng.error()

    # ---- snnls.py:31-79
    def build(self, itrs):
        if self.reached_numeric_limit:
            self.log.warning('the numeric limit was already reached; returning. size = ' + str(self.size())
                             + ', error = ' + str(self.error()))
            return
        if self.A.size == 0:
            self.log.warning('there are no data, returning.')
            return
        if self._use_fused():
            self._limit = self._eng.build_fused(itrs)
        else:
            self.build_stepwise(itrs)
        if self.reached_numeric_limit:
            self.log.warning('the numeric limit has been reached. No more points will be added. size = '
                             + str(self.size()) + ', error = ' + str(self.error()))

    def _use_fused(self):
        if not (self._fusable and self.check_error_monotone):
            return False
        base = _HOOK_OWNER.get(self._alg)
        cls = type(self)
        return base is not None and all(getattr(cls, h) is getattr(base, h) for h in ('_select', '_reweight', '_stabilize'))

    def build_stepwise(self, itrs):
        """The reference's host loop (snnls.py:40-74) over the device primitives."""
        second_try = False
        for _ in range(itrs):
            try:
                nonempty = self.size() > 0
                guarded = self.check_error_monotone and nonempty
                if guarded:
                    err0 = self.error()
                    saved = self._eng.sparse_weights()
                f = self._select()
                self._reweight(f)
                if guarded:
                    err1 = self.error()
                    if err1 > err0:
                        self._eng.set_sparse_weights(*saved)
                        raise NumericalPrecisionError('Error not monotone: curr error = ' + str(err1)
                                                      + ' prev error = ' + str(err0))
                    second_try = False
            except NumericalPrecisionError as e:
                self.log.warning('numerical precision error: ' + str(e))
                if second_try:
                    self.log.warning('iterative step failed a second time. Assuming numeric limit reached.')
                    self.reached_numeric_limit = True
                    break
                self.log.warning('iterative step failed. Stabilizing and retrying...')
                second_try = True
                self._stabilize()

    # ---- snnls.py:82-97 (host NNLS on the <= M active columns; scipy is shared with the reference)
    def optimize(self):
        try:
            cost0 = self.error()
            saved = self._eng.sparse_weights()
            idx, val = saved
            cols = self._eng.columns()
            active = val > 0
            order = np.argsort(idx[active], kind='stable')            # column order of A[:, w > 0]
            a_idx, a_cols = idx[active][order], cols[active][order]
            sol = nnls(a_cols.T, self.b, maxiter=100 * self.n_total)
            self._eng.set_sparse_weights(a_idx, sol[0], a_cols)
            cost1 = self.error()
            if cost1 > cost0 * (1. + util.TOL):
                raise NumericalPrecisionError(
                    'self.optimize() returned a solution with increasing error. Numeric limit possibly reached: '
                    'preverr = ' + str(cost0) + ' err = ' + str(cost1) + '.')
        except NumericalPrecisionError as e:
            self.log.warning(e)
            self._eng.set_sparse_weights(saved[0], saved[1])
            self.reached_numeric_limit = True
            return

    # ---- hooks (snnls.py:99-106)
    def _stabilize(self):
        pass

    def _select(self):
        raise NotImplementedError

    def _reweight(self, f):
        raise NotImplementedError


_HOOK_OWNER = {}


def register_hooks(alg, cls):
    _HOOK_OWNER[alg] = cls
