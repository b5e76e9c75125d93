"""Worker for the multi-process tests (one process per rank).  argv: mode out_prefix.
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def problem(n=3000, s=40, seed=17):
    rng = np.random.RandomState(seed)
    base = rng.randn(n, 8).dot(rng.randn(8, s)) + 0.3 * rng.randn(n, s)
    return base - base.mean(axis=1)[:, None]


def linreg_problem(n=5000, d=12, s=48, seed=23):
    rng = np.random.RandomState(seed)
    X = rng.randn(n, d)
    y = X.dot(rng.randn(d)) + rng.randn(n)
    Z = np.hstack((X, y[:, None]))
    th = rng.randn(s, d) * 0.3
    return Z, th


def overflow_problem(n=6000, s=32, seed=4):
    rng = np.random.RandomState(seed)
    phi = rng.randn(n, 12).dot(rng.randn(12, s)) + 0.3 * rng.randn(n, s)
    phi -= phi.mean(axis=1)[:, None]
    phi[rng.choice(n, 200, replace=False)] = phi[17]              # 200 exact copies of one row, on both shards
    return phi


def main():
    mode, out = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if mode in ('gpu_nccl1', 'gpu_nccl1_torch', 'gpu_nccl1_overflow', 'gpu_nccl1_overflow4', 'gpu_nccl1_bcores'):
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    elif mode.startswith('gpu_ncclN'):
        import torch                        # one GPU per rank: the real multi-GPU transport (needs >= world devices)
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import beta_cores_amd as bc
    comm = bc.ShardComm()
    res = {}
    if mode == 'comm':
        v = np.arange(5, dtype=np.float64) * (rank + 1) + 0.1 * rank
        res['sum'] = comm.sum_in_rank_order(v)
        res['gather'] = comm.gather_host(v)
        res['offset'] = np.array(comm.row_offset(100 + rank))
        res['total'] = np.array(comm.total_rows(100 + rank))
    elif mode in ('fake_giga', 'fake_fw', 'fake_giga_stepwise'):
        from fake_engine import NumpyShardEngine
        phi = problem()
        b = phi.sum(axis=0)
        bounds = bc.shard_bounds(phi.shape[0], world)
        lo, hi = bounds[rank], bounds[rank + 1]
        alg = 'fw' if mode == 'fake_fw' else 'giga'
        eng = NumpyShardEngine(phi[lo:hi], b, alg, lo, comm)
        cls = bc.snnls.FrankWolfe if alg == 'fw' else bc.snnls.GIGA
        s = cls(phi[lo:hi].T, b, comm=comm, engine=eng)
        assert s.n_total == phi.shape[0] and s.row_offset == lo
        if mode.endswith('stepwise'):
            s.build_stepwise(30)
        else:
            s.build(30)
        idx, val = s.sparse_weights()
        res['idx'], res['val'], res['err'] = idx, val, np.array(s.error())
        res['w_dense'] = s.weights()
    elif mode in ('gpu_overflow', 'gpu_nccl1_overflow', 'gpu_overflow4', 'gpu_nccl1_overflow4'):
        # a pre-filter whose candidate lists overflow (200 exact copies of one row, capacity 2): the step is marked on
        # the device and redone with the exact sweep -- through the host-driven exchange (two ranks, one GPU, gloo)
        # and inside bc_snnls_build with the RCCL all-gather in the loop (one rank)
        os.environ['BC_PREFILTER'] = '4' if mode.endswith('4') else '8'      # (4: the two-level form, same mirror behind a 4-bit level)
        os.environ['BC_PREFILTER_CAP'] = '2'
        if mode.startswith('gpu_nccl1_overflow'):
            import torch
            os.environ['BC_FORCE_EXCHANGE'] = '1'
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            ctx = bc.Context(device=0, stream=stream.cuda_stream)
        else:
            ctx = bc.Context(device=0)
        bc.set_default_context(ctx)
        phi = overflow_problem()
        b = phi.sum(axis=0)
        bounds = bc.shard_bounds(phi.shape[0], world)
        lo, hi = bounds[rank], bounds[rank + 1]
        for nm, cls in (('giga', bc.snnls.GIGA), ('fw', bc.snnls.FrankWolfe)):
            s = cls(phi[lo:hi].T, b, comm=comm, row_offset=lo)
            assert s._eng.prefilter == 8
            if mode.startswith('gpu_nccl1_overflow'):
                assert s._eng.native_exchange
            if mode.endswith('4'):
                assert s._eng.prefilter_form == 3
            s.build(12)
            s.build(8)
            idx, val = s.sparse_weights()
            res[nm + '_idx'], res[nm + '_val'], res[nm + '_err'] = idx, val, np.array(s.error())
            res[nm + '_trace_f'] = s._eng.trace()[0]
            res[nm + '_fallbacks'] = np.array(s._eng.prefilter_fallbacks())
            res[nm + '_next'] = np.array(s._select())            # step-wise protocol through the same situation
    elif mode in ('gpu_nccl1_bcores', 'gpu_ncclN_bcores'):
        # BetaCoreset over the library's own RCCL communicator: every gradient is ONE bc_vi_gradient call whose column sums
        # are all-gathered and added in rank order inside the library (1 rank: rehearsal on one GPU; N ranks: one GPU each)
        import torch
        os.environ['BC_FORCE_EXCHANGE'] = '1'
        dev_id = rank if mode == 'gpu_ncclN_bcores' else 0
        stream = torch.cuda.Stream(device=dev_id)
        torch.cuda.set_stream(stream)
        ctx = bc.Context(device=dev_id, stream=stream.cuda_stream)
        bc.set_default_context(ctx)
        Z, th = linreg_problem(n=9000)
        bounds = bc.shard_bounds(Z.shape[0], world)
        lo, hi = bounds[rank], bounds[rank + 1]
        E = np.random.RandomState(3).randn(th.shape[0], Z.shape[1] - 1)
        from oracle import models_ref as M

        def sampler(sz, wts, pts):
            if pts.shape[0] == 0:
                wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
            mu, L, _ = M.linreg_weighted_post(np.zeros(Z.shape[1] - 1), np.eye(Z.shape[1] - 1), 1.0, pts, wts)
            return mu + E.dot(L.T)
        prj = bc.DeviceBetaProjector(sampler, th.shape[0], bc.likelihoods.LinearRegression(1.0), ctx=ctx)
        calls = {'n': 0}
        orig = prj.vi_gradient

        def counted(*a, **kw):
            calls['n'] += 1
            assert kw.get('comm') is not None
            return orig(*a, **kw)
        prj.vi_gradient = counted
        alg = bc.BetaCoreset(Z[lo:hi].copy(), prj, opt_itrs=5, step_sched=lambda i: 0.1 / (1. + i), beta=0.1, learn_beta=False, comm=comm)
        for m in range(6):
            alg.build(1, m + 1)
        res['idx'], res['val'], res['pts'] = alg.idcs, alg.wts, alg.pts
        res['fused_calls'] = np.array(calls['n'])
    elif mode in ('gpu_ncclN_hilbert', 'gpu_ncclN_overflow'):
        import torch
        stream = torch.cuda.Stream(device=rank)
        torch.cuda.set_stream(stream)
        ctx = bc.Context(device=rank, stream=stream.cuda_stream)
        bc.set_default_context(ctx)
        if mode == 'gpu_ncclN_hilbert':
            Z, th = linreg_problem()
            bounds = bc.shard_bounds(Z.shape[0], world)
            lo, hi = bounds[rank], bounds[rank + 1]
            prj = bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0), ctx=ctx)
            h = bc.HilbertCoreset(Z[lo:hi].copy(), prj, comm=comm)
            assert h.snnls._eng.native_exchange
            h.build(25, 25)
            wts, pts, idcs = h.get()
            res['idx'], res['val'], res['err'], res['pts'] = idcs, wts, np.array(h.error()), pts
            res['trace_f'] = h.snnls._eng.trace()[0]
            phi = h.snnls._eng.phi
            res['colsum_native'] = comm.colsum(phi)                      # RCCL all-gather + k_sum_rank_order, world > 1
            res['colsum_host'] = comm.sum_in_rank_order(phi.colsum())
        else:
            os.environ['BC_PREFILTER'] = '8'
            os.environ['BC_PREFILTER_CAP'] = '2'
            phi = overflow_problem()
            b = phi.sum(axis=0)
            bounds = bc.shard_bounds(phi.shape[0], world)
            lo, hi = bounds[rank], bounds[rank + 1]
            for nm, cls in (('giga', bc.snnls.GIGA), ('fw', bc.snnls.FrankWolfe)):
                s = cls(phi[lo:hi].T, b, comm=comm, row_offset=lo)
                assert s._eng.prefilter == 8 and s._eng.native_exchange
                s.build(12)
                s.build(8)
                idx, val = s.sparse_weights()
                res[nm + '_idx'], res[nm + '_val'], res[nm + '_err'] = idx, val, np.array(s.error())
                res[nm + '_trace_f'] = s._eng.trace()[0]
                res[nm + '_fallbacks'] = np.array(s._eng.prefilter_fallbacks())
                res[nm + '_next'] = np.array(s._select())
    elif mode in ('gpu_nccl1', 'gpu_nccl1_torch'):
        # one rank, RCCL backend, exchange forced on: the exact code path of `bench.py --gpus N`
        # (native: RCCL called by the C library inside bc_snnls_build; torch: one torch.distributed call per step)
        import torch
        os.environ['BC_FORCE_EXCHANGE'] = '1'
        os.environ['BC_NATIVE_RCCL'] = '1' if mode == 'gpu_nccl1' else '0'
        Z, th = linreg_problem()
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        ctx = bc.Context(device=0, stream=stream.cuda_stream)
        bc.set_default_context(ctx)
        prj = bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0), ctx=ctx)
        data = bc.DeviceData.from_torch(torch.from_numpy(Z).cuda(), ctx=ctx, row_offset=0)
        h = bc.HilbertCoreset(data, prj, comm=comm)
        if mode == 'gpu_nccl1':
            assert h.snnls._eng.native_exchange and not h.snnls._eng.exchange
        else:
            assert h.snnls._eng.exchange and h.snnls._eng._xchg.on_device and not h.snnls._eng.native_exchange
        h.build(25, 25)
        wts, pts, idcs = h.get()
        res['idx'], res['val'], res['err'] = idcs, wts, np.array(h.error())
        res['pts'] = pts
        res['trace_f'] = h.snnls._eng.trace()[0]
        f = h.snnls._select()                       # step-wise path through the collective as well
        res['next_f'] = np.array(f)
        if mode == 'gpu_nccl1':
            # the replicated S-vector sum inside the library (all-gather + rank-order sum on the device)
            phi = h.snnls._eng.phi
            res['colsum_native'] = comm.colsum(phi)
            res['colsum_local'] = phi.colsum()
    elif mode in ('gpu_hilbert', 'gpu_fw', 'gpu_bcores'):
        Z, th = linreg_problem()
        bounds = bc.shard_bounds(Z.shape[0], world)
        lo, hi = bounds[rank], bounds[rank + 1]
        ctx = bc.Context(device=0)                    # every rank shares GPU 0; records travel over gloo
        bc.set_default_context(ctx)
        model = bc.likelihoods.LinearRegression(1.0)
        if mode == 'gpu_bcores':
            E = np.random.RandomState(3).randn(th.shape[0], Z.shape[1] - 1)
            from oracle import models_ref as M

            def sampler(sz, wts, pts):
                if pts.shape[0] == 0:
                    wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
                mu, L, _ = M.linreg_weighted_post(np.zeros(Z.shape[1] - 1), np.eye(Z.shape[1] - 1), 1.0, pts, wts)
                return mu + E.dot(L.T)
            prj = bc.DeviceBetaProjector(sampler, th.shape[0], model, ctx=ctx)
            alg = bc.BetaCoreset(Z[lo:hi], prj, opt_itrs=5, step_sched=lambda i: 0.1 / (1. + i), beta=0.1,
                                 learn_beta=False, comm=comm)
            for m in range(6):
                alg.build(1, m + 1)
            res['idx'], res['val'] = alg.idcs, alg.wts
            res['pts'] = alg.pts
        else:
            prj = bc.DeviceProjector(lambda n, w, p: th, th.shape[0], model, ctx=ctx)
            cls = bc.snnls.FrankWolfe if mode == 'gpu_fw' else bc.snnls.GIGA
            h = bc.HilbertCoreset(Z[lo:hi], prj, snnls=cls, comm=comm)
            h.build(25, 25)
            wts, pts, idcs = h.get()
            res['idx'], res['val'], res['err'] = idcs, wts, np.array(h.error())
            res['pts'] = pts
            res['trace_f'] = h.snnls._eng.trace()[0]
            local = (idcs >= lo) & (idcs < hi)
            assert np.array_equal(pts[local], Z[idcs[local]])
            assert np.all(np.isnan(pts[~local]))
    elif mode.startswith('gpu_golden_'):
        # the reference's goldens through ROW-SHARDED coresets (two ranks, one GPU, gloo): grouped (F8), sub-sampled (F9),
        # grouped + sub-sampled (F11) tangent spaces of the greedy-VI classes; all-zero rows and sub-sampling with a sharded
        # HilbertCoreset (F12).  Every rank seeds the global NumPy stream like the golden's generator did.
        from conftest import load_golden
        from oracle import models_ref as M
        ctx = bc.Context(device=0)
        bc.set_default_context(ctx)
        what = mode[len('gpu_golden_'):]
        sched = lambda i: 0.1 / (1. + i)

        def lin_sampler(Z, E):
            D = Z.shape[1] - 1

            def sampler(sz, wts, pts):
                if pts.shape[0] == 0:
                    wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
                mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
                return mu + E.dot(L.T)
            return sampler
        if what in ('f8', 'f11'):
            g = load_golden('f8_grouped_vi' if what == 'f8' else 'f11_grouped_subsampled')
            Z, E = g['Z'], g['E']
            groups = [list(map(int, r)) for r in g['groups']]
            S, opt_itrs = E.shape[0], int(g['opt_itrs'])
            bounds = bc.shard_bounds(Z.shape[0], world)
            lo, hi = bounds[rank], bounds[rank + 1]
            model = bc.likelihoods.LinearRegression(1.0)
            for nm in ('bcores', 'svi'):
                if what == 'f11':
                    np.random.seed(110)
                kw = dict(opt_itrs=opt_itrs, step_sched=sched, groups=groups, comm=comm)
                if what == 'f11':
                    kw.update(n_subsample_select=8, n_subsample_opt=50)
                if nm == 'bcores':
                    alg = bc.BetaCoreset(Z[lo:hi].copy(), bc.DeviceBetaProjector(lin_sampler(Z, E), S, model, ctx=ctx), beta=0.1,
                                         learn_beta=False, **kw)
                else:
                    alg = bc.SparseVICoreset(Z[lo:hi].copy(), bc.DeviceProjector(lin_sampler(Z, E), S, model, ctx=ctx), **kw)
                nb = 4 if what == 'f8' else 5
                for m in range(nb):
                    alg.build(1, 12 * (m + 1))
                    res['%s_idcs_%d' % (nm, m)] = alg.idcs.copy()
                    res['%s_w_%d' % (nm, m)] = alg.wts.copy()
                    res['%s_groups_%d' % (nm, m)] = np.array([int(x) for x in alg.selected_groups])
                    res['%s_pts_%d' % (nm, m)] = alg.pts.copy()
                if what == 'f11':
                    res['%s_rng_after' % nm] = np.array(np.random.rand())
        elif what == 'f9':
            g = load_golden('f9_subsampled_gaussian')
            X, Siginv, logdet = g['X'], g['Siginv'], float(g['logdet'])
            d, S = X.shape[1], 40
            mu0, Sig0inv = np.zeros(d), np.eye(d)
            bounds = bc.shard_bounds(X.shape[0], world)
            lo, hi = bounds[rank], bounds[rank + 1]

            def sampler_w(sz, wts, pts):
                if pts.shape[0] == 0:
                    wts, pts = np.zeros(1), np.zeros((1, d))
                muw, LSigw, _ = M.gauss_weighted_post(mu0, Sig0inv, Siginv, pts, wts)
                return muw + np.random.randn(sz, muw.shape[0]).dot(LSigw.T)
            model = bc.likelihoods.GaussianLocation(Siginv, logdet)
            for nm in ('bcores', 'svi'):
                np.random.seed(90)
                kw = dict(opt_itrs=8, n_subsample_opt=60, n_subsample_select=150, step_sched=sched, comm=comm)
                if nm == 'bcores':
                    alg = bc.BetaCoreset(X[lo:hi].copy(), bc.DeviceBetaProjector(sampler_w, S, model, ctx=ctx), beta=.1, learn_beta=False, **kw)
                else:
                    alg = bc.SparseVICoreset(X[lo:hi].copy(), bc.DeviceProjector(sampler_w, S, model, ctx=ctx), **kw)
                for m in range(6):
                    alg.build(1, m + 1)
                    res['%s_idcs_%d' % (nm, m)] = alg.idcs.copy()
                    res['%s_w_%d' % (nm, m)] = alg.wts.copy()
                res['%s_rng_after' % nm] = np.array(np.random.rand())
        elif what == 'f12':
            g = load_golden('f12_constant_rows')
            for kind in ('lin', 'log'):
                for S in (100, 200):
                    tag = '%s_S%d_' % (kind, S)
                    if tag + 'Z' not in g.files:
                        continue
                    Z, th = g[tag + 'Z'], g[tag + 'th']
                    bounds = bc.shard_bounds(Z.shape[0], world)
                    lo, hi = bounds[rank], bounds[rank + 1]
                    model = bc.likelihoods.LinearRegression(1.0) if kind == 'lin' else bc.likelihoods.LogisticRegression()
                    prj = bc.DeviceProjector(lambda n, w, p, th=th: th, S, model, ctx=ctx)
                    for an, cls in (('giga', bc.snnls.GIGA), ('fw', bc.snnls.FrankWolfe)):
                        if tag + an + '_sel' not in g.files:
                            continue
                        steps = g[tag + an + '_sel'].shape[0]
                        h = bc.HilbertCoreset(Z[lo:hi].copy(), prj, snnls=cls, comm=comm)
                        h.build(steps, steps)
                        wts, pts, idcs = h.get()
                        res[tag + an + '_idcs'], res[tag + an + '_wts'], res[tag + an + '_pts'] = idcs, wts, pts
            # sub-sampled HilbertCoreset over sharded rows against the single-rank device run with the same seed
            Z, th = linreg_problem()
            bounds = bc.shard_bounds(Z.shape[0], world)
            lo, hi = bounds[rank], bounds[rank + 1]
            prj = bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0), ctx=ctx)
            np.random.seed(77)
            h = bc.HilbertCoreset(Z[lo:hi].copy(), prj, n_subsample=700, comm=comm)
            h.build(20, 20)
            wts, pts, idcs = h.get()
            res['sub_idcs'], res['sub_wts'], res['sub_pts'] = idcs, wts, pts
    np.savez(out + '.rank%d.npz' % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
