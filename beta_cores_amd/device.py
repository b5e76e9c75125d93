"""Thin RAII wrappers over the C-ABI handles: one Context per process/GPU, data rows
resident on the device (DeviceData) and the tiled Phi matrix (DevicePhi).

DevicePhi quacks enough like the N x S ndarray `vecs` of hilbert.py:11-17 /
bcores.py:44 for the drop-in classes to work without copying it to the host:
`.shape`, `.T` (what the solvers take as A), `.sum(axis=0)` (= b), `np.asarray()`.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _native as N

_default_ctx = None


def _as_f64(a, what):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if not a.flags['C_CONTIGUOUS']:
        raise ValueError(what + ' must be convertible to a C-contiguous float64 array')
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Context:
    """One GPU, one stream.  `stream`: integer hipStream_t handle to launch on (pass
    torch.cuda.current_stream().cuda_stream when collectives issued through
    torch.distributed must order with the kernels); None = library-owned stream."""

    def __init__(self, device=None, stream=None):
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0')) if os.environ.get('BC_DEVICE') is None \
                else int(os.environ['BC_DEVICE'])
        h = C.c_void_p()
        N.call('bc_ctx_create', int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        self.h = h
        self.device = int(device)
        self.stream_handle = int(stream) if stream else None    # None: library-owned stream
        self._fin = weakref.finalize(self, N.load().bc_ctx_destroy, h)

    def sync(self):
        N.call('bc_ctx_sync', self.h)

    def enable_timing(self, on=True):
        """on: False/0 = off, True/1 = time every launch of the dominant kernels, n > 1 = every n-th launch."""
        N.call('bc_ctx_enable_timing', self.h, int(on))

    def timing_classes(self, mask):
        """Which kernel classes enable_timing applies to (bit i = class i); default 0x7.  Classes 3-5 are the other stages of
        a greedy step (rescoring / local winner, candidate all-gather, step finish): for diagnostic passes."""
        N.call('bc_ctx_timing_classes', self.h, int(mask))

    def kernel_time(self, which):
        """(total_ms, launches) of kernel class `which` (0 = K3 sweep, 1 = K1 projection, 2 = K4 gram + reduce, 3 = rescoring /
        local winner, 4 = candidate all-gather, 5 = step finish)."""
        ms, n = C.c_double(), C.c_int64()
        N.call('bc_ctx_kernel_time', self.h, int(which), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def kernel_time_reset(self):
        N.call('bc_ctx_kernel_time_reset', self.h)

    VI_PHASES = ('upload', 'k1_core_rows', 'k1_data_rows', 'colsum_reduce', 'algebra_download')

    def phase_times(self, reset=True):
        """({phase: total GPU ms}, timed calls) of the bc_vi_gradient calls made while timing was on."""
        out = np.zeros(len(self.VI_PHASES))
        n = C.c_int64()
        N.call('bc_ctx_phase_times', self.h, _ptr(out), len(self.VI_PHASES), C.byref(n), 1 if reset else 0)
        return dict(zip(self.VI_PHASES, out.tolist())), n.value


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


class DeviceData:
    """Data rows Z (n x dz, row-major float64) resident in HBM."""

    def __init__(self, z=None, ctx=None, device_ptr=None, shape=None, keepalive=None, row_offset=0):
        self.ctx = ctx or default_context()
        self.row_offset = int(row_offset)     # global index of row 0 when the rows are one shard of a larger set
        h = C.c_void_p()
        if device_ptr is not None:
            n, dz = shape
            N.call('bc_data_from_device', self.ctx.h, C.c_void_p(device_ptr), int(n), int(dz), C.byref(h))
            self._keep = keepalive
        else:
            z = np.atleast_2d(z)
            z = _as_f64(z, 'data')
            n, dz = z.shape
            N.call('bc_data_from_host', self.ctx.h, _ptr(z), int(n), int(dz), C.byref(h))
        self.h = h
        self.shape = (int(n), int(dz))
        self._fin = weakref.finalize(self, N.load().bc_data_destroy, h)

    @classmethod
    def _adopt(cls, handle, shape, ctx):
        """Wrap a bc_data handle a native call allocated (bc_project_from_host); this object owns it from here on."""
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        self.row_offset = 0
        self.h = handle
        self.shape = (int(shape[0]), int(shape[1]))
        self._fin = weakref.finalize(self, N.load().bc_data_destroy, handle)
        return self

    @classmethod
    def slot(cls, dz, cap_rows=256, ctx=None):
        """A re-usable device buffer for small row sets that change every call (coreset points,
        sub-samples): allocate once, then `update(z)` in place."""
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        self.row_offset = 0
        h = C.c_void_p()
        N.call('bc_data_create', self.ctx.h, int(cap_rows), int(dz), C.byref(h))
        self.h = h
        self.shape = (0, int(dz))
        self._fin = weakref.finalize(self, N.load().bc_data_destroy, h)
        return self

    def update(self, z):
        z = _as_f64(np.atleast_2d(z), 'data')
        if z.shape[1] != self.shape[1]:
            raise ValueError('slot holds rows of %d columns, got %d' % (self.shape[1], z.shape[1]))
        N.call('bc_data_upload', self.h, _ptr(z), int(z.shape[0]))
        self.shape = (int(z.shape[0]), self.shape[1])
        return self

    def rows(self, local_idx):
        """Rows by LOCAL index, on the host (m x dz)."""
        idx = np.ascontiguousarray(local_idx, dtype=np.int64).ravel()
        out = np.empty((idx.shape[0], self.shape[1]))
        N.call('bc_data_gather_rows', self.h, _ptr(idx), int(idx.shape[0]), _ptr(out))
        return out

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.rows([idx])[0]
        return self.rows(idx)

    def __len__(self):
        return self.shape[0]

    @classmethod
    def from_torch(cls, t, ctx=None, row_offset=0):
        """Borrow a contiguous float64 CUDA tensor (kept alive by this object).

        The tensor's producer kernels run on torch's current stream; unless this library launches on
        that same stream (Context(stream=...)), wait for them here -- otherwise our first kernel could
        read rows torch has not finished writing."""
        assert t.is_cuda and t.is_contiguous() and str(t.dtype) == 'torch.float64' and t.dim() == 2
        import torch
        cur = torch.cuda.current_stream(t.device)
        c = ctx or default_context()
        if c.stream_handle is None or c.stream_handle != cur.cuda_stream:
            cur.synchronize()
        return cls(ctx=ctx, device_ptr=t.data_ptr(), shape=tuple(t.shape), keepalive=t, row_offset=row_offset)


class _PhiT:
    """`vecs.T`: the S x N view handed to the solvers as A (hilbert.py:17)."""

    def __init__(self, phi):
        self.phi = phi
        self.shape = (phi.shape[1], phi.shape[0])
        self.size = phi.shape[0] * phi.shape[1]

    @property
    def T(self):
        return self.phi

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.phi).T


class DevicePhi:
    """N x S matrix of row-centred (beta-)log-likelihoods in the tiled HBM layout."""

    def __init__(self, handle, ctx, owner=True, release=None):
        self.h = handle
        self.ctx = ctx
        self.refresh()
        if release is not None:
            self._fin = weakref.finalize(self, release, handle)      # hand the buffers back to a pool
        else:
            self._fin = weakref.finalize(self, N.load().bc_phi_destroy, handle) if owner else None

    def refresh(self):
        n, s, off = C.c_int64(), C.c_int32(), C.c_int64()
        N.call('bc_phi_shape', self.h, C.byref(n), C.byref(s), C.byref(off))
        self.shape = (n.value, s.value)
        self.row_offset = off.value
        self.size = n.value * s.value

    # -- construction
    @classmethod
    def from_host(cls, phi, ctx=None, row_offset=0):
        ctx = ctx or default_context()
        phi = np.atleast_2d(phi)
        phi = _as_f64(phi, 'Phi')
        h = C.c_void_p()
        N.call('bc_phi_from_host', ctx.h, _ptr(phi), int(phi.shape[0]), int(phi.shape[1]), int(row_offset), C.byref(h))
        return cls(h, ctx)

    # -- ndarray-like surface
    @property
    def T(self):
        return _PhiT(self)

    def sum(self, axis=None):
        if axis != 0:
            raise NotImplementedError('DevicePhi.sum supports axis=0 only (b = Phi^T 1)')
        return self.colsum()

    def __array__(self, dtype=None, copy=None):
        return self.to_host()

    def __len__(self):
        return self.shape[0]

    # -- device queries
    def colsum(self):
        out = np.empty(self.shape[1])
        N.call('bc_phi_colsum', self.h, _ptr(out))
        return out

    def colsum_all(self, native_comm):
        """Column sums over ALL ranks' shards, added in rank order on the device (bc_phi_colsum_all)."""
        out = np.empty(self.shape[1])
        N.call('bc_phi_colsum_all', self.h, native_comm, _ptr(out))
        return out

    def norms(self):
        out = np.empty(self.shape[0])
        N.call('bc_phi_norms', self.h, _ptr(out))
        return out

    def norm_stats(self):
        z, s = C.c_int64(), C.c_double()
        N.call('bc_phi_norm_stats', self.h, C.byref(z), C.byref(s))
        return z.value, s.value

    def to_host(self):
        out = np.empty(self.shape)
        N.call('bc_phi_to_host', self.h, _ptr(out))
        return out

    def rows(self, local_idx):
        idx = np.ascontiguousarray(local_idx, dtype=np.int64).ravel()
        out = np.empty((idx.shape[0], self.shape[1]))
        N.call('bc_phi_gather_rows', self.h, _ptr(idx), int(idx.shape[0]), _ptr(out))
        return out

    def group_sum(self, groups):
        """DevicePhi whose row g is the sum of this matrix's rows groups[g] (local indices), accumulated in the
        order given -- `np.array([vecs[g].sum(axis=0) for g in groups])` without leaving the device
        (grouped selection, bcores.py:46-50, 56-61)."""
        sizes = [len(g) for g in groups]
        offsets = np.zeros(len(groups) + 1, dtype=np.int64)
        np.cumsum(sizes, out=offsets[1:])
        members = (np.concatenate([np.asarray(g, dtype=np.int64).ravel() for g in groups]) if offsets[-1] > 0
                   else np.zeros(0, dtype=np.int64))
        members = np.ascontiguousarray(members, dtype=np.int64)
        h = C.c_void_p()
        N.call('bc_phi_group_sum', self.h, _ptr(members) if members.size else None, _ptr(offsets), len(groups), C.byref(h))
        return DevicePhi(h, self.ctx)

    def matvec(self, v):
        v = _as_f64(v, 'v')
        out = np.empty(self.shape[0])
        N.call('bc_phi_matvec', self.h, _ptr(v), _ptr(out))
        return out

    def argmax(self, v, mode, post_div=1.0):
        """One fused K3 sweep. mode 0: v = (S,2) [cdir, xw] -> GIGA score; mode 1: v = (S,) -> Phi.v/norm/post_div.
        Returns (global row index or -1, score)."""
        v = _as_f64(v, 'v')
        best, score = C.c_int64(), C.c_double()
        N.call('bc_phi_argmax', self.h, int(mode), _ptr(v), float(post_div), C.byref(best), C.byref(score))
        return best.value, score.value
