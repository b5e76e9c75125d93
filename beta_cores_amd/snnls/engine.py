"""HIP-backed solver engine: the device side of SparseNNLS.

One engine == one bc_snnls handle (this rank's row shard of Phi plus the replicated
sparse weight state).  `comm` (beta_cores_amd.dist.ShardComm) carries the per-step
candidate all-gather when the rows are sharded over several ranks/GPUs.
"""
import ctypes as C
import os
import weakref

import numpy as np

from .. import _native as N
from .. import util
from ..device import DevicePhi, _PhiT, _ptr, default_context

ALG_IDS = {'giga': N.ALG_GIGA, 'fw': N.ALG_FW, 'omp': N.ALG_OMP}


def phi_from_A(A, ctx=None, row_offset=0):
    """Accept what the reference's solvers accept as A (an S x N array, usually the
    transposed view `vecs.T` of a C-contiguous N x S array, hilbert.py:17) or a
    device-resident `DevicePhi.T`, and return a DevicePhi."""
    if isinstance(A, _PhiT):
        return A.phi
    if isinstance(A, DevicePhi):
        raise ValueError('pass DevicePhi.T (S x N) as A, like the reference passes vecs.T')
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError('A must be 2-dimensional (S x N)')
    return DevicePhi.from_host(A.T, ctx=ctx, row_offset=row_offset)   # A.T of vecs.T is the C-contiguous vecs: no copy


class HipEngine:
    def __init__(self, phi, b, alg, norm_sum=None, allow_zero_rows=False, comm=None, tol=1e-12):
        self.phi = phi
        self.ctx = phi.ctx
        self.comm = comm
        self.s = phi.shape[1]
        self.n_local = phi.shape[0]
        self.row_offset = phi.row_offset
        b = np.ascontiguousarray(b, dtype=np.float64)
        if b.shape != (self.s,):
            raise ValueError('b must have shape (%d,), got %s' % (self.s, b.shape))
        if norm_sum is None:
            norm_sum = phi.norm_stats()[1]
            if comm is not None and comm.world > 1:
                norm_sum = comm.sum_in_rank_order(np.array([norm_sum]))[0]
        h = C.c_void_p()
        N.call('bc_snnls_create', self.ctx.h, phi.h, _ptr(b), ALG_IDS[alg], float(norm_sum), 1 if allow_zero_rows else 0,
               C.byref(h))
        self.h = h
        self._fin = weakref.finalize(self, N.load().bc_snnls_destroy, h)
        self._tol = None
        self.set_tolerance(tol)
        on = C.c_int()
        N.call('bc_snnls_prefilter_active', h, C.byref(on))
        form = C.c_int()
        N.call('bc_snnls_prefilter_form', h, C.byref(form))
        self.prefilter_form = int(form.value)    # 0 fp64 sweeps, 1 two-pass pre-filter, 2 branch-and-bound int8 sweep, 3 two-level (4-bit, int8, fp64)
        self.prefilter = int(on.value)           # 0, or the storage precision (16 / 32) of the mirror of Phi the sweeps
                                                 # stream; candidates are rescored in fp64, selections are unchanged
        self.world = 1 if comm is None else comm.world
        # BC_FORCE_EXCHANGE=1 routes a 1-rank group through the collective too (rehearsal of the RCCL path on one GPU)
        self.exchange = self.world > 1 or (comm is not None and os.environ.get('BC_FORCE_EXCHANGE') == '1')
        self.native_exchange = False     # the C library all-gathers by itself (RCCL on its stream): no Python per step
        if self.exchange:
            nc = comm.native_comm(self.ctx) if hasattr(comm, 'native_comm') else None
            if nc is not None:
                N.call('bc_snnls_bind_comm', h, nc)
                self.native_exchange = True
                self.exchange = False    # build / select go straight to the fused entry points
            else:
                n = C.c_int32()
                N.call('bc_snnls_record_doubles', h, C.byref(n))
                self._xchg = comm.make_exchange(n.value, self.ctx)
                N.call('bc_snnls_bind_exchange', h, self.world, C.c_void_p(self._xchg.send_ptr), C.c_void_p(self._xchg.all_ptr))

    def set_tolerance(self, tol):
        tol = float(tol)
        if tol != self._tol:
            N.call('bc_snnls_set_tolerance', self.h, tol)
            self._tol = tol

    def _live_tol(self):
        """The reference reads util.TOL at every use (giga.py:28, snnls.py:92), so util.set_tolerance() after a
        solver was built changes its numeric-limit checks; the device copy follows (one scalar, sent when it changed)."""
        self.set_tolerance(util.TOL)

    # ---- fused loop (snnls.py:31-79 on the device)
    def build_fused(self, itrs):
        self._live_tol()
        lim = C.c_int()
        if not self.exchange:
            try:
                N.call('bc_snnls_build', self.h, int(itrs), C.byref(lim))
            except RuntimeError:
                # a HIP / RCCL failure in the middle of the multi-rank loop: the peers' next all-gather would wait
                # for this rank forever -- abort the communicator so they fail out of RCCL too, then re-raise
                if self.native_exchange and self.world > 1:
                    self.comm.abort()
                raise
        else:
            # the host carries the records between the two halves of a step (gloo, ranks sharing a GPU)
            done0 = C.c_int()
            N.call('bc_snnls_build_end', self.h, None, C.byref(done0), None)
            N.call('bc_snnls_build_begin', self.h, int(itrs))
            left = int(itrs)
            while left > 0:
                for _ in range(left):
                    N.call('bc_snnls_step_local', self.h)
                    self._xchg.all_gather()
                    N.call('bc_snnls_step_finish', self.h)
                done, pending = C.c_int(), C.c_int()
                N.call('bc_snnls_build_end', self.h, C.byref(lim), C.byref(done), C.byref(pending))
                if not pending.value:
                    break
                # a rank's pre-filter overflowed at that step (every rank sees it in the gathered records):
                # nothing was consumed since; redo the step with the exact sweep, then continue
                N.call('bc_snnls_step_local_exact', self.h)
                self._xchg.all_gather()
                N.call('bc_snnls_step_finish', self.h)
                left = int(itrs) - (done.value - done0.value) - 1
                if left <= 0:
                    N.call('bc_snnls_build_end', self.h, C.byref(lim), None, None)
        return bool(lim.value)

    # ---- step-wise protocol
    def select(self):
        self._live_tol()
        f = C.c_int64()
        if not self.exchange:
            N.call('bc_snnls_select', self.h, C.byref(f))
        else:
            N.call('bc_snnls_select_local', self.h)
            self._xchg.all_gather()
            rc = N.load().bc_snnls_select_pick(self.h, C.byref(f))
            if rc == N.BC_RETRY_EXACT:          # a rank's pre-filter overflowed: same step through the exact sweep
                N.call('bc_snnls_select_local_exact', self.h)
                self._xchg.all_gather()
                rc = N.load().bc_snnls_select_pick(self.h, C.byref(f))
            N.check(rc)
        return f.value

    def reweight(self, f):
        self._live_tol()
        N.call('bc_snnls_reweight', self.h, int(f))

    def error(self):
        e = C.c_double()
        N.call('bc_snnls_error', self.h, C.byref(e))
        return e.value

    def size(self):
        n = C.c_int64()
        N.call('bc_snnls_size', self.h, C.byref(n))
        return n.value

    def sparse_weights(self):
        n = C.c_int64()
        N.call('bc_snnls_weights', self.h, 0, None, None, C.byref(n))
        idx = np.empty(n.value, dtype=np.int64)
        val = np.empty(n.value)
        if n.value:
            N.call('bc_snnls_weights', self.h, n.value, _ptr(idx), _ptr(val), C.byref(n))
        return idx, val

    def columns(self):
        n = C.c_int64()
        N.call('bc_snnls_columns', self.h, 0, None, C.byref(n))
        cols = np.empty((n.value, self.s))
        if n.value:
            N.call('bc_snnls_columns', self.h, n.value, _ptr(cols), C.byref(n))
        return cols

    def set_sparse_weights(self, idx, val, cols=None):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if cols is not None:
            cols = np.ascontiguousarray(cols, dtype=np.float64)
        N.call('bc_snnls_set_weights', self.h, int(idx.shape[0]), _ptr(idx), _ptr(val), _ptr(cols) if cols is not None else None)

    def reset(self):
        N.call('bc_snnls_reset', self.h)

    def get_limit(self):
        v = C.c_int()
        N.call('bc_snnls_get_flags', self.h, C.byref(v))
        return bool(v.value)

    def set_limit(self, flag):
        N.call('bc_snnls_set_flags', self.h, 1 if flag else 0)

    def prefilter_fallbacks(self):
        """Sweeps whose pre-filter candidate list overflowed and were redone by the full fp64 sweep (diagnostic)."""
        n = C.c_int64()
        N.call('bc_snnls_prefilter_fallbacks', self.h, C.byref(n))
        return int(n.value)

    def prefilter_stats(self):
        """(sweeps, candidates handed to the exact rescoring, fp64 fallbacks) since this solver was created."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        N.call('bc_snnls_prefilter_stats', self.h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def prefilter_levels(self):
        """Two-level form (prefilter_form 3): (first-level sweeps, rows they listed, rows re-bounded from the int8 records)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        N.call('bc_snnls_prefilter_levels', self.h, C.byref(a), C.byref(b), C.byref(c))
        return int(a.value), int(b.value), int(c.value)

    def trace(self):
        n = C.c_int64()
        N.call('bc_snnls_trace', self.h, 0, None, None, None, C.byref(n))
        f = np.empty(n.value, dtype=np.int64)
        st = np.empty(n.value, dtype=np.int32)
        er = np.empty(n.value)
        if n.value:
            N.call('bc_snnls_trace', self.h, n.value, _ptr(f), _ptr(st), _ptr(er), C.byref(n))
        return f, st, er
