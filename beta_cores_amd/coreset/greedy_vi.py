"""Shared engine of the greedy variational coresets: beta-Cores (coreset/bcores.py) and SparseVI (coreset/sparsevi.py).

Both follow one recipe per added point (bayesiancoresets/coreset/bcores.py:27-150,
sparsevi.py:27-136): re-draw Theta from the current coreset posterior, (beta-)project the
data and the coreset points, pick the row best correlated with the residual, then run
`opt_itrs` projected-ADAM steps on the weights where every gradient needs a fresh
projection.  With a Device(Beta)Projector the N-row projection (K1), its column sums
(K2) and the correlation argmax (K3) all run on the GPU and only S- and M-sized vectors
touch the host; with a black-box projector the reference's NumPy expressions are used
on whatever array the callable returns.
"""
import contextlib
import os
import warnings
import weakref

import numpy as np

from ..device import DeviceData, DevicePhi
from ..util.opt import nn_opt
from .coreset import Coreset


def _flatten(groups):
    return [i for g in groups for i in g]


def resident_copy_of_view(data, ll_projector):
    """A VIEW (Z[:n], a reshaped or memory-mapped array: `data.base is not None`) cannot be guarded against in-place edits
    the way a pinned array is -- writes through its base would go unseen -- so it is not pinned.  Left at that, every
    full-data gradient would upload all N x Dz rows again (~10 GB per gradient at the headline size).  Instead the rows are
    copied to the device ONCE into a DeviceData this coreset owns, and the caller is told: later edits of the host array
    are not seen (pin_data=False keeps the reference's read-the-live-array-every-time behaviour, bcores.py:44)."""
    warnings.warn('data is a view of another array: its %d rows were copied to the GPU once for this coreset; in-place edits of '
                  'the host array after construction are not seen (pass pin_data=False to re-read it on every projection)'
                  % data.shape[0], UserWarning, stacklevel=3)
    return DeviceData(np.ascontiguousarray(data, dtype=np.float64), ctx=getattr(ll_projector, 'ctx', None))


class GreedyVICoreset(Coreset):
    def __init__(self, data, ll_projector, n_subsample_select=None, n_subsample_opt=None, opt_itrs=100,
                 step_sched=lambda i: 1. / (1. + i), mup=None, SigpInv=None, groups=None, selected_groups=None,
                 initialized=False, comm=None, pin_data=True, fused_gradient=True, **kw):
        self.data = data
        self.ll_projector = ll_projector
        n = data.shape[0]
        self.n_subsample_select = None if n_subsample_select is None else min(n, n_subsample_select)
        self.n_subsample_opt = None if n_subsample_opt is None else min(n, n_subsample_opt)
        self.step_sched = step_sched
        self.opt_itrs = opt_itrs
        self.mup = mup
        self.SigpInv = SigpInv
        self.groups = groups
        self.selected_groups = []
        self.fused_gradient = bool(fused_gradient)     # False: every gradient materialises Phi (the general path)
        # BC_FORCE_EXCHANGE=1 keeps a 1-rank group on the collective paths (rehearsal of the RCCL code on one GPU)
        self.comm = comm if (comm is not None and (comm.world > 1 or os.environ.get('BC_FORCE_EXCHANGE') == '1')) else None
        self._dev_data = data if isinstance(data, DeviceData) else None      # rows already resident in HBM
        self._tmode, self._tpos = 'shard', None      # how the last tangent space is spread over ranks (see _tangent)
        self._n_total = n
        if self.comm is not None:
            # `data` is this rank's contiguous row shard; indices (idcs, groups, sub-samples) are GLOBAL row numbers
            off = self.comm.row_offset(n)
            self._local = (off, n)
            self._n_total = self.comm.total_rows(n)
            self.n_subsample_select = None if n_subsample_select is None else min(self._n_total, n_subsample_select)
            self.n_subsample_opt = None if n_subsample_opt is None else min(self._n_total, n_subsample_opt)
            if not isinstance(data, DeviceData):
                self._dev_data = DeviceData(data, ctx=getattr(ll_projector, 'ctx', None), row_offset=off)
            if groups is not None:        # the members of every group that live on this rank, as local row numbers, in member order
                self._local_groups = [np.asarray([i - off for i in g if off <= i < off + n], dtype=np.int64) for g in groups]
        elif pin_data and hasattr(ll_projector, 'pin') and isinstance(data, np.ndarray) and data.base is None and data.ndim == 2 \
                and data.shape[0] >= 4096 and (n_subsample_select is None or n_subsample_opt is None or groups is not None):
            # every full-data tangent space re-projects ALL rows: keep them in HBM instead of uploading per gradient
            # step.  While pinned, `data` is read-only on the host (an in-place edit raises instead of going unseen);
            # pin_data=False restores the reference's read-the-live-array-every-time behaviour (bcores.py:44).
            self._dev_data = ll_projector.pin(data)
            self._unpin = weakref.finalize(self, ll_projector.unpin, data)      # released with this coreset
        elif pin_data and hasattr(ll_projector, 'pin') and isinstance(data, np.ndarray) and data.base is not None and data.ndim == 2 \
                and data.shape[0] >= 4096 and (n_subsample_select is None or n_subsample_opt is None or groups is not None):
            self._dev_data = resident_copy_of_view(data, ll_projector)
        super().__init__(**kw)
        self.initialized = int(initialized) * len(self.wts)

    # -- which projection (plain log-likelihood or beta-likelihood)
    def _proj(self, pts, beta):
        raise NotImplementedError

    # -- pieces shared by select / gradient
    def _tangent(self, n_subsample, w, p, beta, select=False, grad=False):
        """bcores.py:37-72 / sparsevi.py:35-70: returns (vecs, sum_scaling, sub_idcs, group_idcs, corevecs)
        and, with `grad`, the row-centred beta-gradient of the coreset rows (projector.py:56-61) as a sixth item."""
        self.ll_projector.update(w, p)
        group_idcs = None
        self._tmode, self._tpos = 'shard', None
        if n_subsample is None and self.groups is None:
            sub_idcs = None
            vecs = self._proj(self._dev_data if self._dev_data is not None else self.data, beta)
            sum_scaling = 1.
        elif n_subsample is None and self.groups:
            group_idcs = list(range(len(self.groups)))
            sub_idcs = _flatten([self.groups[i] for i in group_idcs])
            vecs = self._group_vecs(group_idcs, beta)
            sum_scaling = 1.
        elif n_subsample and (self.groups is None or not select):
            # (every rank draws the same indices: the ranks share the seed of the global NumPy stream, as they must for
            # the sampler's draws already)
            sub_idcs = np.random.randint(self._n_total, size=n_subsample)
            vecs = self._proj(self.data[sub_idcs], beta) if self.comm is None else self._subsample_vecs(sub_idcs, beta)
            sum_scaling = self._n_total / n_subsample
        else:
            group_idcs = np.random.randint(len(self.groups), size=n_subsample)
            sub_idcs = _flatten([self.groups[i] for i in group_idcs])
            vecs = self._group_vecs(group_idcs, beta)
            sum_scaling = len(self.groups) / n_subsample
        betagrads = None
        if self.pts.size > 0:
            if grad:
                corevecs, betagrads = self.ll_projector.project_f(p, beta, grad=True)
                corevecs, betagrads = np.asarray(corevecs), np.asarray(betagrads)
            else:
                corevecs = np.asarray(self._proj(p, beta))
        else:
            corevecs = np.zeros((0, self.ll_projector.projection_dimension))
            betagrads = np.zeros((0, self.ll_projector.projection_dimension))
        if grad:
            return self._on_device(vecs), sum_scaling, sub_idcs, group_idcs, corevecs, betagrads
        return self._on_device(vecs), sum_scaling, sub_idcs, group_idcs, corevecs

    def _group_vecs(self, group_idcs, beta):
        """One vector per group: the sum of its rows' projections (bcores.py:46-50, 56-61).  A device projector
        projects all rows once (K1) and the groups are summed on the device (bc_phi_group_sum) -- the same rows,
        the same order of additions; a black-box projector is called group by group like in the reference.
        Row-sharded: every rank sums the members it owns (member order), the partial sums are added in rank order
        (one G x S collective) and every rank continues with the same, complete group vectors."""
        from .projector import _DeviceProjectorBase
        if self.comm is not None:
            full = self._proj(self._dev_data, beta)
            part = np.asarray(full.group_sum([self._local_groups[i] for i in group_idcs]))
            self._tmode = 'replicated'
            return self.comm.sum_in_rank_order(part)
        if isinstance(self.ll_projector, _DeviceProjectorBase):
            full = self._proj(self._dev_data if self._dev_data is not None else self.data, beta)
            if isinstance(full, DevicePhi):
                return full.group_sum([self.groups[i] for i in group_idcs])
        return np.array([np.sum(np.asarray(self._proj(self.data[self.groups[i], :], beta)), axis=0) for i in group_idcs])

    def _subsample_vecs(self, sub_idcs, beta):
        """Row-sharded sub-sample (bcores.py:51-55): this rank projects the drawn rows it owns, in draw order; `_tpos`
        keeps their positions in the draw (the reference's row numbers of `vecs`)."""
        off, n = self._local
        pos = np.flatnonzero((sub_idcs >= off) & (sub_idcs < off + n))
        self._tmode, self._tpos = 'positions', pos
        if pos.size == 0:
            return None
        return self._proj(self.data[sub_idcs[pos] - off], beta)

    def _on_device(self, vecs):
        """Black-box projectors hand back a host array; its N-row reductions still run on the GPU."""
        if isinstance(vecs, DevicePhi) or vecs is None:
            return vecs
        return DevicePhi.from_host(np.ascontiguousarray(vecs, dtype=np.float64), ctx=getattr(self.ll_projector, 'ctx', None))

    def _colsum(self, vecs):
        if self.comm is None or self._tmode == 'replicated':
            return vecs.sum(axis=0)
        if self._tmode == 'positions':          # a handful of rows per rank: host collective
            S = self.ll_projector.projection_dimension
            return self.comm.sum_in_rank_order(vecs.sum(axis=0) if vecs is not None else np.zeros(S))
        return self.comm.colsum(vecs)              # in-library all-gather + rank-order sum over RCCL (dist.py)

    def _best_correlation(self, vecs, resid):
        """`np.argmax(corrs)`, `corrs.max()` for corrs = vecs.resid / ||vecs_i|| / S (bcores.py:78-81), one K3 sweep.

        All-zero rows stay in the tangent space: the filter at bcores.py:67-68 / sparsevi.py:64-65 needs
        select=True with groups=None and the ungrouped _select passes neither (bcores.py:76).  Their correlation
        is 0/0 = NaN, so NumPy's argmax is the FIRST such row and the maximum is NaN -- every `>` against it is
        False (golden F13).  The sweep masks zero-norm rows, so that case is decided here from the norms.
        Row-sharded: ranks compare (score, row number of `vecs`) and the lowest row number wins ties, like argmax."""
        S = self.ll_projector.projection_dimension if vecs is None else vecs.shape[1]
        shared = self.comm is not None and self._tmode != 'replicated'
        if self._tmode == 'positions':
            rownum = (lambda k: int(self._tpos[k]))                     # local row -> position in the draw
        else:
            rownum = (lambda k: int(k))                                 # argmax / row_offset already speak global rows
        n_zero = vecs.norm_stats()[0] if vecs is not None else 0
        n_zero_all = int(self.comm.sum_in_rank_order(np.array([float(n_zero)]))[0]) if shared else n_zero
        if n_zero_all > 0:
            first = np.inf
            if n_zero > 0:
                k = int(np.flatnonzero(vecs.norms() == 0.)[0])
                first = float(rownum(k) if self._tmode == 'positions' else vecs.row_offset + k)
            if shared:
                first = float(self.comm.gather_host(np.array([first])).min())
            return int(first), np.nan
        if vecs is not None:
            best, score = vecs.argmax(resid, mode=1, post_div=float(S))
            if self._tmode == 'positions' and best >= 0:
                best = rownum(best - vecs.row_offset)
        else:
            best, score = -1, -np.inf
        if shared:
            cands = self.comm.gather_host(np.array([score, float(best)]))
            best, score = -1, -np.inf
            for sc, bi in cands:
                bi = int(bi)
                if bi >= 0 and (best < 0 or sc > score or (sc == score and bi < best)):
                    best, score = bi, sc
        return best, score

    def _row(self, f):
        return self._rows([f])[0]

    def _rows(self, idx):
        """data[idx] for GLOBAL row numbers: every rank contributes the rows it owns, the rest comes from the others."""
        idx = np.asarray(idx, dtype=np.int64)
        if self.comm is None:
            return self.data[idx]
        off, n = self._local
        out = np.zeros((idx.shape[0], self.data.shape[1]))
        mine = (idx >= off) & (idx < off + n)
        if mine.any():
            out[mine] = np.asarray(self.data[idx[mine] - off], dtype=np.float64)
        return self.comm.sum_in_rank_order(out)

    def _append(self, new_idcs, new_pts):
        k = len(new_idcs)
        self.wts = np.concatenate((self.wts, np.zeros(k)))
        self.idcs = np.concatenate((self.idcs, np.asarray(new_idcs, dtype=np.int64)))
        new_pts = np.atleast_2d(new_pts)
        self.pts = new_pts.copy() if self.pts.size == 0 else np.vstack((self.pts, new_pts))

    # -- bcores.py:27-35
    def _build(self, itrs, sz):
        if (self.groups is None or self._size_check_always) and self.size() + itrs > sz:
            raise ValueError(self.alg_name + '._build(): # itrs + current size cannot exceed total desired size sz. '
                             '# itr = ' + str(itrs) + ' cur sz: ' + str(self.size()) + ' desired sz: ' + str(sz))
        for _ in range(itrs):
            self._select()
            self._optimize()

    # -- bcores.py:74-124
    def _select(self):
        beta = self._beta()
        grouped = self.groups is not None
        vecs, sum_scaling, sub_idcs, group_idcs, corevecs = self._tangent(self.n_subsample_select, self.wts, self.pts,
                                                                         beta, select=True)
        scale = 1. if (grouped and self.n_subsample_select is None) else sum_scaling
        resid = scale * self._colsum(vecs) - self.wts.dot(corevecs)
        best, best_corr = self._best_correlation(vecs, resid)
        with np.errstate(invalid='ignore', divide='ignore'):      # an all-zero core row gives 0/0 = NaN, as in the reference
            corecorrs = np.fabs(corevecs.dot(resid) / np.sqrt((corevecs ** 2).sum(axis=1))) / corevecs.shape[1]
        if not grouped:
            if corecorrs.size == 0 or best_corr > corecorrs.max():
                f = sub_idcs[best] if sub_idcs is not None else best
                if f not in self.idcs:      # the sub-sample may contain coreset points
                    self._append([f], self._row(f))
            return
        if corecorrs.shape[0] > self.initialized:
            max_core = corecorrs[self.initialized:].max()
        else:
            max_core = -np.inf
        if corecorrs.size == 0 or best_corr > max_core:
            f = best if self.n_subsample_select is None else group_idcs[best]
            if not any(f == g for g in self.selected_groups):
                self.selected_groups.append(f)
                self._append(self.groups[f], self._rows(self.groups[f]) if self.comm is not None else self.data[self.groups[f], :])

    # -- bcores.py:141-150
    def _fused_gradient(self, w, beta, overlap=None):
        """The full-data, ungrouped gradient in one native call (bc_vi_gradient): the data rows go through the
        store-free K1 (only `vecs.sum(axis=0)` is needed of them, bcores.py:144-145), the coreset rows and the M x S
        algebra stay on the device, one host synchronisation.  None when this mode does not apply (black-box
        projector, sub-sampling, groups, a transport without a native communicator, no coreset rows yet)."""
        from .projector import _DeviceProjectorBase
        if not self.fused_gradient or self.n_subsample_opt is not None or self.groups is not None \
                or self.pts.size == 0 or not isinstance(self.ll_projector, _DeviceProjectorBase):
            return None
        nc = None
        if self.comm is not None:
            nc = self.comm.native_comm(self.ll_projector.ctx)
            if nc is None:
                return None
        self.ll_projector.update(w, self.pts)
        g = self.ll_projector.vi_gradient(self._dev_data if self._dev_data is not None else self.data, self.pts, w, 1.,
                                          beta=self._fused_beta(beta), comm=nc, overlap=overlap)
        if g is None:
            raise RuntimeError('fused gradient not applicable after the sampler ran')      # guarded by the checks above
        return g

    def _fused_beta(self, beta):
        return None      # SparseVI: plain log-likelihood; BetaCoreset overrides

    def _fused_ok(self):
        """The shape conditions of vi_gradient that do not depend on the sampler's output."""
        if self.pts.size == 0:
            return False
        m, dz = np.atleast_2d(self.pts).shape
        return self.ll_projector.projection_dimension <= 256 and m * (dz + 1) <= 60000

    def _optimize(self):
        beta = self._beta()
        fused = self._fused_ok() if hasattr(self.ll_projector, 'vi_gradient') else False

        # a sampler that can draw its next normals ahead of time (samplers.*PosteriorSampler.prefetch) does so while the GPU
        # works on the current gradient -- except after the LAST gradient of this call: what follows that one is not this
        # loop's next sampler call, and the stream must not be advanced on its behalf
        prefetch = getattr(getattr(self.ll_projector, 'sampler', None), 'prefetch', None) if fused else None
        calls = [0]

        def grd(w):
            if fused:
                calls[0] += 1
                g = self._fused_gradient(w, beta, overlap=prefetch if calls[0] < self.opt_itrs else None)
                if g is not None:
                    return g
            vecs, sum_scaling, _, _, corevecs = self._tangent(self.n_subsample_opt, w, self.pts, beta)
            resid = sum_scaling * self._colsum(vecs) - w.dot(corevecs)
            return -corevecs.dot(resid) / corevecs.shape[1]
        # (a sampler of ours keeps its single-thread BLAS limit open over the whole loop instead of entering it per call)
        scope = getattr(getattr(self.ll_projector, 'sampler', None), 'scope', None)
        with (scope() if scope is not None else contextlib.nullcontext()):
            self.wts = nn_opt(self.wts, grd, opt_itrs=self.opt_itrs, step_sched=self.step_sched)

    def error(self):
        return 0.   # the reference has no KL estimate either (bcores.py:152-153)
