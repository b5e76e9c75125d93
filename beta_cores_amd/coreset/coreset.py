import numpy as np

from .. import util
from ..util.errors import NumericalPrecisionError
from ..util.log import make_logger


class Coreset(object):
    """Coreset state (weights, indices, points) and the build/optimize guards.

    Protocol of bayesiancoresets/coreset/coreset.py:7-71: `build(itrs, sz)` never
    returns more than sz points and raises ValueError when asked to shrink;
    `optimize()` reverts and sets `reached_numeric_limit` when the error grows by
    more than a factor 1+TOL; `get()` returns the strictly positive entries.

    Fenced quirk: the reference shares its default `wts/idcs/pts` arrays between
    instances (mutable default arguments, coreset.py:8) and BetaCoreset grows them in
    place, so state leaks across instances.  Here every instance gets fresh arrays."""

    def __init__(self, initial_sz=10, wts=None, idcs=None, pts=None):
        self.alg_name, self.log = make_logger(self)
        self._clear()
        if wts is not None:
            self.wts = wts
        if idcs is not None:
            self.idcs = idcs
        if pts is not None:
            self.pts = pts

    def _clear(self):
        """the empty coreset: no weights, no indices, no points, numeric limit not reached"""
        self.wts, self.pts = np.array([]), np.array([])
        self.idcs = np.array([], dtype=np.int64)
        self.reached_numeric_limit = False

    def reset(self):
        self._clear()

    def size(self):
        return np.count_nonzero(self.wts > 0)

    def get(self):
        keep = self.wts > 0
        return self.wts[keep], self.pts[keep, :], self.idcs[keep]

    def error(self):
        raise NotImplementedError()

    def build(self, itrs, sz):
        if self.reached_numeric_limit:
            return
        if sz < self.size():
            raise ValueError('%s.build(): a coreset cannot shrink -- asked for sz = %d, it already holds %d points'
                             % (self.alg_name, sz, self.size()))
        self._build(itrs, sz)
        if self.reached_numeric_limit:
            self.log.warning('numeric limit reached: no further points will be added (size %d, error %g)'
                             % (self.size(), self.error()))

    def optimize(self):
        try:
            cost0 = self.error()
            saved = (self.wts.copy(), self.idcs.copy(), self.pts.copy())
            self._optimize()
            cost1 = self.error()
            if cost1 > cost0 * (1. + util.TOL):
                raise NumericalPrecisionError('optimize() made the error grow (%g -> %g): numeric limit, weights restored'
                                              % (cost0, cost1))
        except NumericalPrecisionError as e:
            self.log.warning(e)
            self.wts, self.idcs, self.pts = saved
            self.reached_numeric_limit = True
            return

    def _optimize(self):
        raise NotImplementedError

    def _build(self, itrs, sz):
        raise NotImplementedError
