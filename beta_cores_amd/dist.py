"""Row sharding over ranks (one process per GPU) and the per-step candidate exchange.

The N axis shards naturally: rank r owns the contiguous rows
[bounds[r], bounds[r+1]) of Z / Phi; Theta, b, xw and the sparse weight list are
replicated.  The only data-path collective per greedy step is ONE all-gather of a
(S + 4)-double candidate record per rank (score, global index, row norm, valid flag,
the un-normalised column), after which every rank runs the identical device finish
kernel -- so the replicated state stays bit-identical and independent of the world
size.

Transports of that all-gather, in order of preference:
  * native  -- backend "nccl": the C library calls RCCL itself on its own stream (bc_comm.hip).  The
               communicator is bootstrapped here (rank 0's ncclUniqueId broadcast through
               torch.distributed) and checked with a rank-coded pattern; the whole multi-rank greedy loop
               then runs inside bc_snnls_build, no Python between steps.  BC_NATIVE_RCCL=0 disables it.
  * torch   -- backend "nccl": torch.distributed.all_gather_into_tensor on the kernels' stream, one
               Python round trip per step (fallback if the native bootstrap fails).
  * staged  -- backend "gloo" (CPU tests, or several ranks sharing one GPU): the record goes through
               pinned host tensors.
"""
import ctypes as C
import os
import warnings
import weakref

import numpy as np


def shard_bounds(n_rows, world, align=128):
    """Contiguous row ranges, sizes equal up to one tile: shard starts are multiples of
    `align` (the Phi tile height) so that tiles never straddle ranks."""
    tiles = (n_rows + align - 1) // align
    per, extra = divmod(tiles, world)
    bounds = [0]
    for r in range(world):
        bounds.append(min(n_rows, bounds[-1] + (per + (1 if r < extra else 0)) * align))
    bounds[-1] = n_rows
    return bounds


class _Exchange:
    """Send/recv buffers for the per-step record all-gather."""

    def __init__(self, comm, rec_len, ctx):
        import torch
        self.comm = comm
        self.torch = torch
        self.rec_len = rec_len
        self.ctx = ctx
        dev = torch.device('cuda', ctx.device)
        self.send = torch.zeros(rec_len, dtype=torch.float64, device=dev)
        self.all = torch.zeros(comm.world * rec_len, dtype=torch.float64, device=dev)
        self.send_ptr = self.send.data_ptr()
        self.all_ptr = self.all.data_ptr()
        self.on_device = comm.backend == 'nccl'
        if self.on_device:
            cur = torch.cuda.current_stream(dev).cuda_stream
            if ctx.stream_handle is None or ctx.stream_handle != cur:
                raise RuntimeError('RCCL exchange: create the Context on torch\'s current stream '
                                   '(Context(device, stream=torch.cuda.current_stream().cuda_stream) with a '
                                   'non-default torch stream) so collectives order with the kernels')
        if not self.on_device:
            self.h_send = torch.zeros(rec_len, dtype=torch.float64).pin_memory()
            self.h_all = torch.zeros(comm.world * rec_len, dtype=torch.float64).pin_memory()

    def all_gather(self):
        dist = self.torch.distributed
        if self.on_device:
            # RCCL: enqueued behind the sweep on the same (torch current) stream, no host sync
            dist.all_gather_into_tensor(self.all, self.send, group=self.comm.group)
        else:
            self.ctx.sync()
            self.h_send.copy_(self.send)
            dist.all_gather_into_tensor(self.h_all, self.h_send, group=self.comm.group)
            self.all.copy_(self.h_all)
            self.torch.cuda.current_stream(self.ctx.device).synchronize()


def _with_deadline(fn, what, rank):
    """Run a COLLECTIVE bootstrap call with a deadline (BC_RCCL_TIMEOUT seconds, default 180).  If a peer died or
    never arrived the call cannot return; there is no way to cancel it from here either, so the process exits
    non-zero -- the launcher (torchrun) then takes the whole job down instead of leaving every rank blocked."""
    import threading
    box = {}

    def run():
        try:
            fn()
        except BaseException as e:        # noqa: BLE001 -- handed to the caller's thread
            box['err'] = e
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(float(os.environ.get('BC_RCCL_TIMEOUT', '180')))
    if t.is_alive():
        import sys
        sys.stderr.write('beta_cores_amd: rank %d: %s did not return within BC_RCCL_TIMEOUT -- a peer is missing; '
                         'exiting so the job is torn down\n' % (rank, what))
        sys.stderr.flush()
        os._exit(70)
    if 'err' in box:
        raise box['err']


class ShardComm:
    """torch.distributed process group + the shard arithmetic the solvers need."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('ShardComm needs an initialised torch.distributed process group')
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    # -- small replicated reductions (host values, deterministic rank order)
    def gather_host(self, arr):
        """all-gather a small host float64 array -> (world, ...) ndarray, via a device tensor for nccl."""
        import torch
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        t = torch.from_numpy(arr.copy())
        if self.backend == 'nccl':
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in out])

    def sum_in_rank_order(self, arr):
        """Sum over ranks in rank order: the same bits on every rank (and on re-runs)."""
        parts = self.gather_host(arr)
        acc = parts[0].copy()
        for r in range(1, self.world):
            acc = acc + parts[r]
        return acc

    def total_rows(self, n_local):
        return int(self.sum_in_rank_order(np.array([float(n_local)]))[0])

    def row_offset(self, n_local):
        sizes = self.gather_host(np.array([float(n_local)]))[:, 0]
        return int(sizes[:self.rank].sum())

    def make_exchange(self, rec_len, ctx):
        return _Exchange(self, rec_len, ctx)

    def native_comm(self, ctx):
        """bc_comm handle (RCCL communicator owned by the C library) for `ctx`, or None when the native
        exchange is unavailable (backend is not nccl, BC_NATIVE_RCCL=0, bootstrap or self-test failed)."""
        if self.backend != 'nccl' or os.environ.get('BC_NATIVE_RCCL', '1') == '0':
            return None
        cache = self.__dict__.setdefault('_native', {})
        key = id(ctx)
        if key in cache:
            if cache[key][0] is not None and cache[key][3]:
                raise RuntimeError('the native RCCL communicator of this context was aborted (a rank failed mid-loop)')
            return cache[key][0]
        import torch
        from . import _native as N
        dev = torch.device('cuda', ctx.device)

        def agreed(flag):
            # every rank must take the same branch: one rank falling back alone would dead-lock the collectives
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
            return int(t.item()) == 1

        handle, why = None, None
        uid = (C.c_ubyte * 128)()
        try:                              # local part: find RCCL, check the device, rank 0 draws the communicator id
            lib = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            N.call('bc_comm_load', lib.encode() if os.path.exists(lib) else None)
            N.call('bc_comm_precheck', ctx.h)          # everything bc_comm_create needs locally, voted on below
            if self.rank == 0:
                N.call('bc_comm_unique_id', C.cast(uid, C.c_void_p), 128)
        except Exception as e:            # noqa: BLE001 -- any failure here means "use the torch transport"
            why = e
        if agreed(why is None):
            t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
            root = self.dist.get_global_rank(self.group, 0) if self.group is not None else 0
            self.dist.broadcast(t, src=root, group=self.group)
            uid = (C.c_ubyte * 128)(*t.cpu().tolist())
            try:                          # collective part: communicator + wiring check (rank-coded pattern)
                h = C.c_void_p()
                _with_deadline(lambda: N.call('bc_comm_create', ctx.h, C.cast(uid, C.c_void_p), self.rank, self.world,
                                              C.byref(h)), 'ncclCommInitRank', self.rank)
                handle = h
                _with_deadline(lambda: N.call('bc_comm_selftest', h), 'RCCL self-test', self.rank)
            except Exception as e:        # noqa: BLE001
                why = e
            if not agreed(why is None):
                if handle is not None:
                    N.load().bc_comm_destroy(handle)
                handle = None
        if handle is None:
            warnings.warn('native RCCL exchange unavailable (rank %d: %s); using torch.distributed per step'
                          % (self.rank, why if why is not None else 'another rank failed'))
        fin = None
        if handle is not None:
            fin = weakref.finalize(self, N.load().bc_comm_destroy, handle)
            fin.atexit = False            # at interpreter exit the process teardown reclaims it; peers may be gone
        cache[key] = [handle, fin, ctx, False]      # keeps ctx alive as long as the communicator; [3]: aborted
        return handle

    def colsum(self, vecs):
        """b = sum over all ranks' shards of the column sums of a DevicePhi (hilbert.py:17, bcores.py:77): inside the
        library over RCCL when the native exchange is up (all-gather + rank-order sum on the device, one host
        copy), else the same sum through torch.distributed on host arrays.  Bit-identical either way."""
        nc = self.native_comm(vecs.ctx) if self.world > 1 or os.environ.get('BC_FORCE_EXCHANGE') == '1' else None
        if nc is not None:
            return vecs.colsum_all(nc)
        return self.sum_in_rank_order(vecs.colsum())

    def abort(self):
        """ncclCommAbort on the native communicators: a rank that failed mid-loop calls this before it exits."""
        from . import _native as N
        for ent in self.__dict__.get('_native', {}).values():
            if ent[0] is not None:
                N.load().bc_comm_abort(ent[0])
                ent[3] = True                 # native_comm() no longer hands this handle out

    def close(self):
        """Destroy the native communicators (call before torch.distributed.destroy_process_group)."""
        for handle, fin, _, _ in self.__dict__.pop('_native', {}).values():
            if fin is not None:
                fin()

    def barrier(self):
        self.dist.barrier(group=self.group)
