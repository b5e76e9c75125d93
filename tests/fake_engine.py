"""A NumPy model of the HIP solver engine, for protocol tests of the sharded (multi-rank)
path on machines without a GPU.  It mirrors the CONTRACT of the device kernels -- per step one
candidate record per rank [score, global index, row norm, valid, column(S)], all-gathered,
then an identical replicated finish on every rank -- not their code.  Test infrastructure
only: the product never imports this module."""
import numpy as np

TOL = 1e-12


class NumpyShardEngine:
    def __init__(self, phi_local, b, alg, row_offset, comm, norm_sum=None):
        self.phi = np.ascontiguousarray(phi_local, dtype=np.float64)
        self.n_local, self.s = self.phi.shape
        self.row_offset = row_offset
        self.comm = comm
        self.alg = alg
        self.b = np.asarray(b, dtype=np.float64)
        self.norms = np.sqrt((self.phi ** 2).sum(axis=1))
        self.bnorm = np.sqrt((self.b ** 2).sum())
        self.bn = self.b / self.bnorm if self.bnorm > 0 else self.b * 0
        ns = np.array([self.norms.sum()])
        self.norm_sum = float(comm.sum_in_rank_order(ns)[0]) if comm is not None else float(ns[0])
        self.world = 1 if comm is None else comm.world
        self.reset()

    # -- state
    def reset(self):
        self.idx = np.zeros(0, dtype=np.int64)
        self.val = np.zeros(0)
        self.cols = np.zeros((0, self.s))
        self.limit = False
        self.tr = []
        self._records = None

    def _xw(self):
        return self.val.dot(self.cols) if self.idx.size else np.zeros(self.s)

    def error(self):
        return float(np.sqrt(((self._xw() - self.b) ** 2).sum()))

    def size(self):
        return int((self.val > 0).sum())

    def sparse_weights(self):
        return self.idx.copy(), self.val.copy()

    def columns(self):
        return self.cols.copy()

    def set_sparse_weights(self, idx, val, cols=None):
        idx = np.asarray(idx, dtype=np.int64)
        new_cols = np.zeros((idx.shape[0], self.s))
        for j, f in enumerate(idx):
            if cols is not None:
                new_cols[j] = cols[j]
                continue
            hit = np.flatnonzero(self.idx == f)
            if hit.size:
                new_cols[j] = self.cols[hit[0]]
            elif self._records is not None and any(r[3] and int(r[1]) == f for r in self._records):
                new_cols[j] = [r for r in self._records if r[3] and int(r[1]) == f][0][4:]
            elif self.row_offset <= f < self.row_offset + self.n_local:
                new_cols[j] = self.phi[f - self.row_offset]
            else:
                raise ValueError('column %d not available on this rank' % f)
        self.idx, self.val, self.cols = idx, np.asarray(val, dtype=np.float64).copy(), new_cols

    def get_limit(self):
        return self.limit

    def set_limit(self, flag):
        self.limit = bool(flag)

    def trace(self):
        f = np.array([t[0] for t in self.tr], dtype=np.int64)
        st = np.array([t[1] for t in self.tr], dtype=np.int32)
        er = np.array([t[2] for t in self.tr])
        return f, st, er

    # -- one step
    def _prep(self):
        xw = self._xw()
        if self.alg == 'giga':
            nw = np.sqrt((xw ** 2).sum())
            nw = 1. if nw == 0. else nw
            xwn = xw / nw
            cdir = self.bn - self.bn.dot(xwn) * xwn
            cn = np.sqrt((cdir ** 2).sum())
            if cn < TOL:
                return None
            return cdir / cn, xwn
        return (self.b - xw,)

    def _local_record(self, vecs):
        rec = np.zeros(4 + self.s)
        rec[1] = -1
        ok = self.norms > 0
        if vecs is None or not ok.any():
            return rec
        if self.alg == 'giga':
            s0 = self.phi.dot(vecs[0]) / np.where(ok, self.norms, 1.)
            s1 = self.phi.dot(vecs[1]) / np.where(ok, self.norms, 1.)
            good = np.logical_and(s1 > -1. + 1e-14, 1. - s1 ** 2 > 0.)
            den = np.where(good, np.sqrt(np.abs(1. - s1 ** 2)), np.inf)
            sc = s0 / den
        else:
            sc = self.phi.dot(vecs[0]) / np.where(ok, self.norms, 1.)
        sc = np.where(ok, sc, -np.inf)
        f = int(np.argmax(sc))
        rec[0], rec[1], rec[2], rec[3] = sc[f], f + self.row_offset, self.norms[f], 1.
        rec[4:] = self.phi[f]
        return rec

    def _exchange(self, rec):
        if self.comm is None or self.world == 1:
            return rec[None, :]
        return self.comm.gather_host(rec)

    def _pick(self, records):
        best = None
        for r in records:
            if not r[3]:
                continue
            if best is None or r[0] > best[0] or (r[0] == best[0] and r[1] < best[1]):
                best = r
        return best

    def _step_sizes(self, xf, nf_stored):
        xw = self._xw()
        if self.alg == 'giga':
            nw = np.sqrt((xw ** 2).sum())
            nw = 1. if nw == 0. else nw
            nf = np.sqrt((xf ** 2).sum())
            gA = self.bn.dot(xf / nf) - self.bn.dot(xw / nw) * (xw / nw).dot(xf / nf)
            gB = self.bn.dot(xw / nw) - self.bn.dot(xf / nf) * (xw / nw).dot(xf / nf)
            if gA <= 0. or gB < 0:
                return None
            a = gB / (gA + gB) / nw
            b = gA / (gA + gB) / nf
            x = a * xw + b * xf
            nx = np.sqrt((x ** 2).sum())
            scale = self.bnorm / nx * (x / nx).dot(self.bn)
            return a * scale, b * scale
        if self.size() == 0:
            return 0., self.norm_sum / nf_stored
        c = self.norm_sum / nf_stored
        num = (c * xf - xw).dot(self.b - xw)
        den = ((c * xf - xw) ** 2).sum()
        if num < 0. or den == 0. or num > den:
            return None
        return 1. - num / den, c * num / den

    def _apply(self, f, xf, alpha, beta):
        self.val = alpha * self.val
        hit = np.flatnonzero(self.idx == f)
        if hit.size:
            self.val[hit[0]] = max(0., self.val[hit[0]] + beta)
        elif beta > 0:
            self.idx = np.append(self.idx, np.int64(f))
            self.val = np.append(self.val, beta)
            self.cols = np.vstack((self.cols, xf[None, :]))

    def build_fused(self, itrs):
        retried = False
        for _ in range(itrs):
            if self.limit:
                break
            guard = self.size() > 0
            vecs = self._prep()
            records = self._exchange(self._local_record(vecs))
            self._records = records
            fail = vecs is None
            f = -1
            if not fail:
                win = self._pick(records)
                fail = win is None
            if not fail:
                f, xf = int(win[1]), win[4:].copy()
                st = self._step_sizes(xf, win[2])
                fail = st is None
            if not fail:
                saved = (self.idx.copy(), self.val.copy(), self.cols.copy())
                err0 = self.error()
                self._apply(f, xf, *st)
                if guard:
                    if self.error() > err0:
                        self.idx, self.val, self.cols = saved
                        fail = True
                    else:
                        retried = False
            if fail:
                if retried:
                    self.limit = True
                else:
                    retried = True
            self.tr.append((f, int(fail), self.error()))
        return self.limit

    # -- step-wise protocol
    def select(self):
        from beta_cores_amd.util.errors import NumericalPrecisionError
        vecs = self._prep()
        records = self._exchange(self._local_record(vecs))
        self._records = records
        if vecs is None:
            raise NumericalPrecisionError('cdirnrm < TOL')
        win = self._pick(records)
        if win is None:
            raise ValueError('no selectable row')
        return int(win[1])

    def reweight(self, f):
        from beta_cores_amd.util.errors import NumericalPrecisionError
        rec = [r for r in self._records if r[3] and int(r[1]) == f]
        if rec:
            xf, nf = rec[0][4:].copy(), rec[0][2]
        elif self.row_offset <= f < self.row_offset + self.n_local:
            xf, nf = self.phi[f - self.row_offset].copy(), self.norms[f - self.row_offset]
        else:
            raise ValueError('column not available')
        st = self._step_sizes(xf, nf)
        if st is None:
            raise NumericalPrecisionError('precision loss in the closed-form step')
        self._apply(f, xf, *st)
