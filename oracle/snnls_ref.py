"""NumPy restatement of the reference's sparse-NNLS solvers (oracle; tests only).

Follows, pass for pass:
  bayesiancoresets/snnls/snnls.py:8-106      (greedy loop, guards, optimize)
  bayesiancoresets/snnls/giga.py:6-64        (GIGA)
  bayesiancoresets/snnls/frankwolfe.py:5-40  (Frank-Wolfe)
  bayesiancoresets/snnls/orthopursuit.py:7-42(OMP)
  bayesiancoresets/snnls/sampling.py:6-37    (importance / uniform sampling)

The cost structure is deliberately the reference's: dense length-N weight
vector, a separate normalised copy of A, and five N x S sweeps per GIGA
iteration (two error() calls, one in select, one two-column score product,
one in reweight).  That makes it an honest CPU baseline as well as a checker.
Every step also appends to ``self.trace`` so tests can compare per-iteration
selections, not only final weights.
"""
import numpy as np
from scipy.optimize import nnls as _scipy_nnls

TOL = 1e-12  # bayesiancoresets/util/__init__.py:4


class RefNumericalPrecisionError(Exception):
    """bayesiancoresets/util/errors.py:1-2"""


def _l2(v):
    return np.sqrt((v ** 2).sum())


class _RefGreedy:
    """snnls.py:8-106 -- state + the guarded greedy loop."""

    monotone = True  # snnls.py:16 (check_error_monotone)

    def __init__(self, A, b):
        self.A = A            # S x N (usually the transposed view of an N x S C-array)
        self.b = b            # S
        self.n = A.shape[1]
        self.w = np.zeros(self.n)
        self.hit_limit = False
        self.trace = []       # [(f or -1, status)] per consumed iteration

    # snnls.py:18-29
    def reset(self):
        self.w = np.zeros(self.n)
        self.hit_limit = False

    def size(self):
        return int((self.w > 0).sum())

    def weights(self):
        return self.w.copy()

    def error(self):
        return _l2(self.A.dot(self.w) - self.b)

    # snnls.py:31-79
    def build(self, itrs):
        if self.hit_limit or self.A.size == 0:
            return
        second_chance_used = False
        for _ in range(itrs):
            f = -1
            try:
                had_points = self.size() > 0
                guard = self.monotone and had_points
                if guard:
                    err_before = self.error()
                    w_before = self.w.copy()
                f = self.select()
                self.reweight(f)
                if guard:
                    err_after = self.error()
                    if err_after > err_before:
                        self.w = w_before
                        raise RefNumericalPrecisionError('error not monotone')
                    second_chance_used = False
                self.trace.append((int(f), 0))
            except RefNumericalPrecisionError:
                self.trace.append((int(f), 1))
                if second_chance_used:
                    self.hit_limit = True
                    break
                second_chance_used = True
                self.stabilize()

    # snnls.py:82-97
    def optimize(self):
        cost_before = self.error()
        w_before = self.w.copy()
        active = self.w > 0
        sol = _scipy_nnls(self.A[:, active], self.b, maxiter=100 * self.n)
        self.w[active] = sol[0]
        if self.error() > cost_before * (1. + TOL):
            self.w = w_before
            self.hit_limit = True

    def stabilize(self):
        pass

    def select(self):
        raise NotImplementedError

    def reweight(self, f):
        raise NotImplementedError


def _column_norms(A):
    return np.sqrt((A ** 2).sum(axis=0))


class RefGIGA(_RefGreedy):
    """giga.py:6-64"""

    def __init__(self, A, b):
        super().__init__(A, b)
        nrm = _column_norms(self.A)
        if np.any(nrm == 0):
            raise ValueError('A must not have any 0 columns')   # giga.py:11-12
        self.An = self.A / nrm
        self.bnorm = _l2(self.b)
        if self.bnorm == 0.:
            raise RefNumericalPrecisionError('norm of b must be > 0')
        self.bn = self.b / self.bnorm

    def select(self):                                 # giga.py:20-38
        xw = self.A.dot(self.w)
        nw = _l2(xw)
        nw = 1. if nw == 0. else nw
        xw /= nw
        cdir = self.bn - self.bn.dot(xw) * xw
        cn = _l2(cdir)
        if cn < TOL:
            raise RefNumericalPrecisionError('cdirnrm < TOL')
        cdir /= cn
        sc = self.An.T.dot(np.hstack((cdir[:, None], xw[:, None])))
        ok = np.logical_and(sc[:, 1] > -1. + 1e-14, 1. - sc[:, 1] ** 2 > 0.)
        sc[ok, 1] = np.sqrt(1. - sc[ok, 1] ** 2)
        sc[np.logical_not(ok), 1] = np.inf
        return (sc[:, 0] / sc[:, 1]).argmax()

    def reweight(self, f):                            # giga.py:40-64
        xw = self.A.dot(self.w)
        nw = _l2(xw)
        nw = 1. if nw == 0. else nw
        xf = self.A[:, f]
        nf = _l2(xf)
        gA = self.bn.dot((xf / nf)) - self.bn.dot((xw / nw)) * (xw / nw).dot((xf / nf))
        gB = self.bn.dot((xw / nw)) - self.bn.dot((xf / nf)) * (xw / nw).dot((xf / nf))
        if gA <= 0. or gB < 0:
            raise RefNumericalPrecisionError('geodesic step degenerate')
        a = gB / (gA + gB) / nw
        b = gA / (gA + gB) / nf
        x = a * xw + b * xf
        nx = _l2(x)
        scale = self.bnorm / nx * (x / nx).dot(self.bn)
        alpha = a * scale
        beta = b * scale
        self.w = alpha * self.w
        self.w[f] = max(0., self.w[f] + beta)


class RefFrankWolfe(_RefGreedy):
    """frankwolfe.py:5-40"""

    def __init__(self, A, b):
        super().__init__(A, b)
        self.nrm = _column_norms(self.A)
        if np.any(self.nrm == 0):
            raise ValueError('A must not have any 0 columns')
        self.An = self.A / self.nrm

    def select(self):                                 # frankwolfe.py:15-17
        r = self.b - self.A.dot(self.w)
        return (self.An.T.dot(r)).argmax()

    def reweight(self, f):                            # frankwolfe.py:19-40
        if self.size() == 0:
            alpha = 0.
            beta = self.nrm.sum() / self.nrm[f]
        else:
            nsum = self.nrm.sum()
            nf = self.nrm[f]
            xw = self.A.dot(self.w)
            xf = self.A[:, f]
            num = (nsum / nf * xf - xw).dot(self.b - xw)
            den = ((nsum / nf * xf - xw) ** 2).sum()
            if num < 0. or den == 0. or num > den:
                raise RefNumericalPrecisionError('precision loss in line search')
            alpha = 1. - num / den
            beta = nsum / nf * num / den
        self.w = alpha * self.w
        self.w[f] = max(0., self.w[f] + beta)


class RefOrthoPursuit(_RefGreedy):
    """orthopursuit.py:7-42"""

    def __init__(self, A, b):
        super().__init__(A, b)
        nrm = _column_norms(self.A)
        if np.any(nrm == 0):
            raise ValueError('A must not have any 0 columns')
        self.An = self.A / nrm

    def select(self):                                 # orthopursuit.py:17-35
        r = self.b - self.A.dot(self.w)
        dots = self.An.T.dot(r)
        if self.size() == 0:
            return dots.argmax()
        fpos = dots.argmax()
        pos = dots[fpos]
        act = self.w > 0
        fneg = (-dots[act]).argmax()
        neg = (-dots[act])[fneg]
        if pos >= neg:
            return fpos
        return np.arange(self.n)[act][fneg]

    def reweight(self, f):                            # orthopursuit.py:37-42
        self.w[f] = 1.
        act = self.w > 0
        sol = _scipy_nnls(self.A[:, act], self.b, maxiter=100 * self.n)
        self.w[act] = sol[0]


class RefImportanceSampling(_RefGreedy):
    """sampling.py:6-32 (draws from the global NumPy RNG, like the reference)."""

    monotone = False                                  # sampling.py:16

    def __init__(self, A, b):
        super().__init__(A, b)
        self.cts = np.zeros(self.n)
        self.ps = _column_norms(self.A)
        if np.any(self.ps > 0):
            self.ps /= self.ps.sum()
        else:
            self.ps = np.ones(self.n) / float(self.n)

    def reset(self):
        super().reset()
        self.cts = np.zeros(self.n)

    def select(self):
        return np.random.choice(self.ps.shape[0], p=self.ps)

    def reweight(self, f):
        self.cts[f] += 1
        self.w = (self.cts / self.cts.sum()) / self.ps


class RefUniformSampling(RefImportanceSampling):
    """sampling.py:34-37"""

    def __init__(self, A, b):
        super().__init__(A, b)
        self.ps = np.ones(self.n) / float(self.n)
