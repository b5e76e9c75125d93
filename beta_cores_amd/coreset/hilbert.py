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
        if n_subsample is None:
            sub_idcs = None
            src = data
        else:
            if sharded:
                raise NotImplementedError('n_subsample with sharded rows is not supported')
            n_subsample = min(data.shape[0], n_subsample)
            sub_idcs = np.random.randint(data.shape[0], size=n_subsample)
            src = data[sub_idcs]
        if sharded and not isinstance(src, DeviceData):
            src = DeviceData(src, ctx=getattr(ll_projector, 'ctx', None), row_offset=comm.row_offset(src.shape[0]))
        vecs = ll_projector.project(src)
        self._zero_map = None
        solver_kw = {}
        if isinstance(vecs, DevicePhi):
            n_zero, _ = vecs.norm_stats()
            if sharded:
                n_zero = int(comm.sum_in_rank_order(np.array([float(n_zero)]))[0])
            if n_zero > 0:
                if sharded:
                    raise NotImplementedError('all-zero projection rows with sharded data')
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
        self.wts = val
        self.idcs = self.sub_idcs[idx] if self.sub_idcs is not None else idx
        if self.comm is not None and hasattr(self.snnls, 'row_offset') and getattr(self.snnls, 'comm', None) is not None:
            off = self.snnls.row_offset
            n_loc = self.snnls.n_local
            local = (self.idcs >= off) & (self.idcs < off + n_loc)
            pts = np.full((self.idcs.shape[0], self.data.shape[1]), np.nan)
            pts[local] = self.data[self.idcs[local] - off]
            self.pts = pts
        else:
            self.pts = self.data[self.idcs]

    def error(self):
        return self.snnls.error()
