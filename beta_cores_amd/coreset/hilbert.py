import os

import numpy as np

from ..device import DeviceData, DevicePhi
from ..snnls.giga import GIGA
from .coreset import Coreset


class HilbertCoreset(Coreset):
    """Hilbert coreset: project once, then sparse NNLS on (Phi^T, Phi^T 1).

    Same constructor and behaviour as bayesiancoresets/coreset/hilbert.py:6-43
    (`snnls=` is the solver plug-in seam; `n_subsample` sub-samples rows with
    replacement from the global NumPy RNG).  With a DeviceProjector the N x S matrix
    is produced in HBM and handed to the solver without ever visiting the host; with a
    black-box projector the host array is uploaded by the solver.

    All-zero rows: the reference drops them before building the solver
    (hilbert.py:16), after which `np.where(w>0)` indexes the FILTERED matrix while
    `pts = data[idcs]` indexes the unfiltered data (hilbert.py:32-33).  The device
    path keeps such rows in place (masked out of the sweep) and maps indices the same
    way the reference would, quirk included.

    `comm` (ShardComm): `data` is this rank's row shard; b and the column-norm sum are
    combined over ranks once, the solver exchanges one candidate record per step, and
    get() returns global row indices with `pts` filled for locally owned rows only."""

    def __init__(self, data, ll_projector, n_subsample=None, snnls=GIGA, comm=None, **kw):
        self.comm = comm
        sharded = comm is not None and (comm.world > 1 or os.environ.get('BC_FORCE_EXCHANGE') == '1')
        self._sub_rows = None
        if n_subsample is None:
            sub_idcs = None
            src = data
        elif not sharded:
            n_subsample = min(data.shape[0], n_subsample)
            sub_idcs = np.random.randint(data.shape[0], size=n_subsample)
            src = data[sub_idcs]
        else:
            # row-sharded sub-sample (hilbert.py:12-15): every rank draws the same global indices (shared seed of the global
            # stream), the drawn rows are collected once (each rank contributes the ones it owns; n_subsample x Dz doubles),
            # and the sub-sampled matrix is sharded again by POSITION in the draw -- the solver's row numbers are the
            # reference's: positions of `vecs`
            n_loc = data.shape[0]
            off, n_tot = comm.row_offset(n_loc), comm.total_rows(n_loc)
            n_subsample = min(n_tot, n_subsample)
            sub_idcs = np.random.randint(n_tot, size=n_subsample)
            rows = np.zeros((n_subsample, data.shape[1]))
            mine = (sub_idcs >= off) & (sub_idcs < off + n_loc)
            if mine.any():
                rows[mine] = np.asarray(data[sub_idcs[mine] - off], dtype=np.float64)
            rows = comm.sum_in_rank_order(rows)
            self._sub_rows = rows
            from ..dist import shard_bounds
            bounds = shard_bounds(n_subsample, comm.world)
            lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
            src = DeviceData(rows[lo:hi], ctx=getattr(ll_projector, 'ctx', None), row_offset=lo)
        if sharded and not isinstance(src, DeviceData):
            src = DeviceData(src, ctx=getattr(ll_projector, 'ctx', None), row_offset=comm.row_offset(src.shape[0]))
        vecs = ll_projector.project(src)
        self._zero_map = None
        self._zero_rows = None
        solver_kw = {}
        if isinstance(vecs, DevicePhi):
            n_zero, _ = vecs.norm_stats()
            n_zero_local = n_zero
            if sharded:
                n_zero = int(comm.sum_in_rank_order(np.array([float(n_zero)]))[0])
            if n_zero > 0:
                if sharded:
                    # all-zero rows (dropped at hilbert.py:16) shift the reference's indices (hilbert.py:32): their GLOBAL
                    # row numbers are exchanged once (count per rank, then the padded lists), every rank keeps the sorted list
                    mine = vecs.row_offset + np.flatnonzero(vecs.norms() == 0.) if n_zero_local > 0 else np.zeros(0, dtype=np.int64)
                    counts = comm.gather_host(np.array([float(mine.shape[0])]))[:, 0].astype(np.int64)
                    pad = np.full(int(counts.max()), -1.)
                    pad[:mine.shape[0]] = mine
                    allz = comm.gather_host(pad)
                    self._zero_rows = np.sort(np.concatenate([allz[r, :counts[r]] for r in range(comm.world)]).astype(np.int64))
                else:
                    self._zero_map = np.cumsum(vecs.norms() == 0.)      # rows dropped before each index
                solver_kw['allow_zero_rows'] = True
            if sharded:
                b = comm.colsum(vecs)                      # all ranks' shards, summed in rank order inside the library
                solver_kw['comm'] = comm
            else:
                b = vecs.sum(axis=0)
            self.snnls = snnls(vecs.T, b, **solver_kw)
        else:
            if sharded:
                raise NotImplementedError('sharded rows need a DeviceProjector')
            vecs = vecs[np.sqrt((vecs ** 2).sum(axis=1)) > 0., :]
            self.snnls = snnls(vecs.T, vecs.sum(axis=0))
        self.sub_idcs = sub_idcs
        self.data = data
        super().__init__(**kw)

    def reset(self):
        self.snnls.reset()
        super().reset()

    def _build(self, itrs, sz):
        if self.snnls.size() + itrs > sz:
            raise ValueError(self.alg_name + '._build(): # itrs + current size cannot exceed total desired size sz. '
                             '# itr = ' + str(itrs) + ' cur sz: ' + str(self.snnls.size()) + ' desired sz: ' + str(sz))
        self.snnls.build(itrs)
        self._pull()

    def _optimize(self):
        self.snnls.optimize()
        self._pull()

    def _pull(self):
        if hasattr(self.snnls, 'sparse_weights'):
            idx, val = self.snnls.sparse_weights()              # ascending global index == np.where(w>0) order
        else:
            w = self.snnls.weights()
            idx, val = np.where(w > 0)[0], w[w > 0]
        if self._zero_map is not None:
            idx = idx - self._zero_map[idx]                      # index into the zero-row-filtered matrix (hilbert.py:16,32)
        elif self._zero_rows is not None:
            idx = idx - np.searchsorted(self._zero_rows, idx)    # the same shift from the exchanged list of all-zero rows
        self.wts = val
        self.idcs = self.sub_idcs[idx] if self.sub_idcs is not None else idx
        if self._sub_rows is not None:
            # sharded sub-sample: the drawn rows are replicated; the reference's quirk (hilbert.py:33: `data[idcs]` with idcs
            # already mapped through sub_idcs) is reproduced from the owners of those rows
            self.pts = self._owned_rows(self.idcs)
        elif self.comm is not None and hasattr(self.snnls, 'row_offset') and getattr(self.snnls, 'comm', None) is not None:
            self.pts = self._owned_rows(self.idcs)
        else:
            self.pts = self.data[self.idcs]

    def _owned_rows(self, idcs):
        """data[idcs] for global row numbers on a row-sharded data set: rows this rank owns, NaN for the others'."""
        n_loc = self.data.shape[0]
        if not hasattr(self, '_data_off'):
            self._data_off = self.comm.row_offset(n_loc)
        off = self._data_off
        local = (idcs >= off) & (idcs < off + n_loc)
        pts = np.full((idcs.shape[0], self.data.shape[1]), np.nan)
        pts[local] = self.data[idcs[local] - off]
        return pts

    def error(self):
        return self.snnls.error()
