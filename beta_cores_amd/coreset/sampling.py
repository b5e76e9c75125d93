"""UniformSamplingCoreset (bayesiancoresets/coreset/sampling.py:5-52): the RAND baseline of the example drivers --
rows (or groups of rows) drawn uniformly with replacement from the global NumPy RNG, weights N * count / draws.
Host-only bookkeeping; nothing here touches the N x S matrix."""
import numpy as np

from .coreset import Coreset


class UniformSamplingCoreset(Coreset):
    def __init__(self, data, groups=None, selected_groups=None, **kw):
        super().__init__(**kw)
        self.data = data
        self.groups = groups
        self.selected_groups = []
        # a coreset handed initial points counts each of them once (sampling.py:9-11)
        seeded = 'wts' in kw
        self.ct_idcs = self.idcs.tolist() if seeded else []
        self.cts = [1] * len(self.ct_idcs)

    def reset(self):
        self.cts, self.ct_idcs = [], []
        super().reset()

    def _weights(self):
        c = np.array(self.cts)
        return self.data.shape[0] * c / c.sum()

    def _draw_rows(self, draws):                       # sampling.py:27-37
        for _ in range(draws):
            f = np.random.randint(self.data.shape[0])
            try:
                self.cts[self.ct_idcs.index(f)] += 1
            except ValueError:
                self.ct_idcs.append(f)
                self.cts.append(1)
        self.wts = self._weights()
        self.idcs = np.array(self.ct_idcs)
        self.pts = self.data[self.idcs]

    def _draw_groups(self, draws):                     # sampling.py:38-52: whole groups, each at most once
        width = self.data.shape[1]
        for _ in range(draws):
            f = np.random.randint(len(self.groups))
            if f in self.selected_groups:
                continue
            members = np.asarray(self.groups[f], dtype=np.int64)
            self.selected_groups.append(f)
            self.ct_idcs.append(self.groups[f])
            self.cts.extend([1] * members.shape[0])
            self.idcs = np.concatenate((np.asarray(self.idcs, dtype=np.int64), members))
            self.pts = np.vstack((np.asarray(self.pts).reshape(-1, width), self.data[members, :]))
            self.wts = self._weights()

    def _build(self, itrs, sz):
        if self.size() + itrs > sz:
            raise ValueError('%s._build(): %d more draws on top of the current %d points would exceed sz = %d'
                             % (self.alg_name, itrs, self.size(), sz))
        (self._draw_rows if self.groups is None else self._draw_groups)(itrs)

    def error(self):
        return 0.
