"""UniformSamplingCoreset (bayesiancoresets/coreset/sampling.py:5-52): the RAND baseline of the example drivers --
rows (or groups of rows) drawn uniformly with replacement from the global NumPy RNG, weights N * count / draws.
Host-only bookkeeping; nothing here touches the N x S matrix."""
import numpy as np

from .coreset import Coreset


class UniformSamplingCoreset(Coreset):
    def __init__(self, data, groups=None, selected_groups=None, **kw):
        super().__init__(**kw)
        self.data = data
        if 'wts' in kw:                      # initialised to a dummy random subset (sampling.py:9-11)
            self.cts = [1] * len(self.idcs.tolist())
            self.ct_idcs = self.idcs.tolist()
        else:
            self.cts = []
            self.ct_idcs = []
        self.groups = groups
        self.selected_groups = []

    def reset(self):
        self.cts = []
        self.ct_idcs = []
        super().reset()

    def _build(self, itrs, sz):
        if self.size() + itrs > sz:
            raise ValueError('%s._build(): %d more draws on top of the current %d points would exceed sz = %d'
                             % (self.alg_name, itrs, self.size(), sz))
        n = self.data.shape[0]
        if self.groups is None:              # sampling.py:27-37
            for _ in range(itrs):
                f = np.random.randint(n)
                if f in self.ct_idcs:
                    self.cts[self.ct_idcs.index(f)] += 1
                else:
                    self.ct_idcs.append(f)
                    self.cts.append(1)
            self.wts = n * np.array(self.cts) / np.array(self.cts).sum()
            self.idcs = np.array(self.ct_idcs)
            self.pts = self.data[self.idcs]
        else:                                # sampling.py:38-52: whole groups, each at most once
            for _ in range(itrs):
                f = np.random.randint(len(self.groups))
                if f not in self.selected_groups:
                    newpoints = self.data[self.groups[f], :]
                    k = newpoints.shape[0]
                    self.ct_idcs.append(self.groups[f])
                    self.cts += [1] * k
                    self.idcs = np.concatenate((np.asarray(self.idcs, dtype=np.int64), np.asarray(self.groups[f], dtype=np.int64)))
                    self.pts = np.vstack((np.asarray(self.pts).reshape(-1, self.data.shape[1]), newpoints))
                    self.wts = n * np.array(self.cts) / np.array(self.cts).sum()
                    self.selected_groups.append(f)

    def error(self):
        return 0.
