#!/usr/bin/env python3
"""bench.py -- greedy coreset iterations/s on the BASELINE.json workload.

Workload (BASELINE.json configs[3], the one `metric` is quoted on; it fits one GPU):
Zellner linear regression, N = 10M rows, D = 128, S = 100 posterior samples, GIGA via
HilbertCoreset.  With --gpus G the N rows are sharded over G ranks (strong scaling: the
total N is fixed), one process per GPU, one candidate-record all-gather (RCCL) per step.

A "step" = one greedy iteration of SparseNNLS.build (select + reweight + monotone guard)
over Phi resident in HBM.  The one-off K1 projection is timed separately and reported
as points*dims/s.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus 1 --steps 100 --warmup 10]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 100 --warmup 10
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TIMING_STRIDE = 5
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
FP64_MFMA_PEAK_TF = 78.6
CHUNK = 1 << 20                # synthetic data is generated in global chunks of 2^20 rows (= 8192 tiles)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--rows', dest='n', type=int, default=10_000_000)
    ap.add_argument('--dim', dest='d', type=int, default=128)
    ap.add_argument('--samples', dest='s', type=int, default=100)
    ap.add_argument('--alg', default='giga', choices=['giga', 'fw'])
    ap.add_argument('--cpu-sample', type=int, default=1_000_000, help='rows of the same data given to the CPU baseline')
    ap.add_argument('--cpu-iters', type=int, default=20)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--proj-reps', type=int, default=3)
    return ap.parse_args()


def gen_rows(torch, dev, lo, hi, d, thstar):
    """Rows [lo, hi) of the global synthetic design: X ~ N(0,1), y = X.th* + eps, 10% of rows
    with y ~ N(10, 0.5^2) (outliers as in model_neurlinr.py:63).  Chunk-seeded so the
    global data set does not depend on how many ranks generate it."""
    Z = torch.empty((hi - lo, d + 1), dtype=torch.float64, device=dev)
    c0, c1 = lo // CHUNK, (hi - 1) // CHUNK
    for c in range(c0, c1 + 1):
        g = torch.Generator(device=dev)
        g.manual_seed(40_000 + c)
        X = torch.randn((CHUNK, d), generator=g, dtype=torch.float64, device=dev)
        eps = torch.randn((CHUNK,), generator=g, dtype=torch.float64, device=dev)
        u = torch.rand((CHUNK,), generator=g, dtype=torch.float64, device=dev)
        yo = 10. + 0.5 * torch.randn((CHUNK,), generator=g, dtype=torch.float64, device=dev)
        y = torch.where(u < 0.1, yo, X @ thstar + eps)
        a, b = max(lo, c * CHUNK), min(hi, (c + 1) * CHUNK)
        Z[a - lo:b - lo, :d] = X[a - c * CHUNK:b - c * CHUNK]
        Z[a - lo:b - lo, d] = y[a - c * CHUNK:b - c * CHUNK]
        del X, eps, u, yo, y
    return Z


def posterior_samples(bc, data, d, s, comm):
    """Theta: S draws from the exact full-data Gaussian posterior, the 'optimal tangent space'
    (zellner_gaussian/main.py:71), via weighted_post with w = 1, prior N(0, I), sigsq = 1
    (model_linreg.py:25-34); its X^T X / X^T y reductions are kernel K4 on this rank's rows."""
    mu, L, _ = bc.weighted_post(np.zeros(d), np.eye(d), 1.0, data, None, comm=comm)
    E = np.random.default_rng(40).standard_normal((s, d))
    return mu + E.dot(L.T)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    # rehearsal knobs (not used by the driver): all ranks on one GPU + gloo transport lets a 1-GPU box run
    # the multi-rank code path end to end:  BC_BENCH_DEVICE=0 BC_BENCH_BACKEND=gloo torchrun --nproc-per-node 2 ...
    if os.environ.get('BC_BENCH_DEVICE') is not None:
        local_rank = int(os.environ['BC_BENCH_DEVICE'])
    backend = os.environ.get('BC_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    comm = None
    force_xchg = os.environ.get('BC_FORCE_EXCHANGE') == '1' and 'RANK' in os.environ   # 1-GPU rehearsal of the RCCL path
    if world > 1 or force_xchg:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    import beta_cores_amd as bc
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = bc.Context(device=local_rank, stream=stream.cuda_stream)
    bc.set_default_context(ctx)
    if world > 1 or force_xchg:
        comm = bc.ShardComm()

    N, D, S = args.n, args.d, args.s
    bounds = bc.shard_bounds(N, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_local = hi - lo

    def barrier():
        torch.cuda.synchronize(dev)
        ctx.sync()
        if world > 1:
            dist.barrier()

    # ---------------- synthetic data (untimed set-up)
    t_setup = time.time()
    g0 = torch.Generator(device=dev)
    g0.manual_seed(39)
    thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
    Z = gen_rows(torch, dev, lo, hi, D, thstar)
    torch.cuda.synchronize(dev)
    t_setup = time.time() - t_setup

    data = bc.DeviceData.from_torch(Z, ctx=ctx, row_offset=lo)
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    theta = posterior_samples(bc, data, D, S, comm)
    t_post = time.perf_counter() - t0
    k4_ms, k4_n = ctx.kernel_time(2)
    model = bc.likelihoods.LinearRegression(1.0)
    prj = bc.DeviceProjector(lambda n, w, p: theta, S, model, ctx=ctx)

    # ---------------- K1: projection, points*dims/s
    import gc
    gc.collect()
    gc.disable()          # a gen-2 collection (tens of ms with torch loaded) must not land in a timed region
    ctx.enable_timing(True)
    prj.project(data)                      # warm-up (also allocates Phi)
    barrier()
    ctx.kernel_time_reset()
    barrier()
    t0 = time.perf_counter()
    phi = None
    for i in range(args.proj_reps):
        ta = time.perf_counter()
        phi = None                         # hand the previous result's buffers back first: every rep writes the same Phi
        phi = prj.project(data)
        if os.environ.get('BC_BENCH_VERBOSE'):
            sys.stderr.write('project rep %d: %.3f ms (host call incl. sync)\n' % (i, 1e3 * (time.perf_counter() - ta)))
    barrier()
    t_proj = (time.perf_counter() - t0) / args.proj_reps
    if world > 1:
        tt = torch.tensor([t_proj], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_proj = float(tt.item())
    k1_ms, k1_n = ctx.kernel_time(1)
    k1_ms_per = k1_ms / max(k1_n, 1)
    k1_bytes = 8.0 * n_local * (D + 1) + 8.0 * n_local * S
    k1_flops = 2.0 * n_local * D * S

    # ---------------- solver construction (b, norms already fused into K1)
    cls = bc.snnls.GIGA if args.alg == 'giga' else bc.snnls.FrankWolfe
    t0 = time.perf_counter()
    alg = bc.HilbertCoreset(data, prj, snnls=cls, comm=comm)
    barrier()
    t_init = time.perf_counter() - t0
    total = args.warmup + args.steps

    # ---------------- greedy iterations: W untimed, then exactly K timed
    alg.build(args.warmup, total)
    barrier()
    # the K3 launch time for the roofline comes from HIP events around every TIMING_STRIDE-th sweep of the timed
    # region: an event pair costs ~11 us of stream time, too much to put around all of them next to a 50 us sweep
    ctx.enable_timing(TIMING_STRIDE)
    ctx.kernel_time_reset()
    t0 = time.perf_counter()
    alg.snnls.build(args.steps)
    barrier()
    t_steps = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([t_steps], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_steps = float(tt.item())
    k3_ms, k3_n = ctx.kernel_time(0)
    k3_ms_per = k3_ms / max(k3_n, 1)
    eng_ = alg.snnls._eng
    exchange_kind = ('rccl all-gather issued by the C library (native loop)' if eng_.native_exchange else
                     'torch.distributed all-gather per step (%s)' % backend if eng_.exchange else 'none (single rank)')
    pref = int(alg.snnls._eng.prefilter)      # 0, or the storage precision of the streamed mirror (16 / 32)
    if pref:
        # fp16: the sweep streams the fp16 mirror of the normalised Phi (2 B/element, planes padded to a multiple of
        # 10) and one live-mask byte per 8 rows; it writes only per-tile / per-block bounds.  fp32: 4 B/element + the
        # norms, and one fp32 upper bound written per row.  Candidates are rescored in fp64 either way.
        if pref == 8:
            # int8 mirror: one byte per element (k-groups of 4 padded to a multiple of 5) + (scale, delta) halfs per row
            k3_bytes = 4.0 * n_local * (-(-(-(-S // 4)) // 5) * 5) + 4.0 * n_local
        elif pref == 16:
            k3_bytes = 2.0 * n_local * (-(-S // 10) * 10) + n_local / 8.0
        else:
            k3_bytes = 4.0 * n_local * S + 8.0 * n_local + 4.0 * n_local
    else:
        k3_bytes = 8.0 * n_local * S + 8.0 * n_local      # one streaming read of Phi + the norms
    alg._pull()
    wts, pts, idcs = alg.get()
    err = alg.error()
    f_tr, st_tr, _ = alg.snnls._eng.trace()

    pname = 'int8' if pref == 8 else 'fp%d' % pref
    kname = 'k_sweep_i8' if pref == 8 else 'k_sweep_f%d' % pref
    out = None
    if rank == 0:
        ach = k3_bytes / (k3_ms_per * 1e-3) / 1e9 if k3_ms_per > 0 else 0.0
        # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
        # this process); only quoted when it was collected on exactly this workload shape
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
            key = ('k_sweep_i8' if pref == 8 else 'k_sweep_f%d' % pref) if pref else 'k_sweep'
            if tj['config'] == {'N': N, 'D': D, 'S': S, 'n_gpus': world} and args.alg == 'giga' and key in tj:
                traffic = tj[key]['traffic_bytes_per_launch']
                traffic_src = 'profiles/r01_pmc_traffic.json: ' + tj[key]['correction']
        except Exception:
            pass
        out = {
            'metric': 'greedy coreset iterations/sec', 'value': args.steps / t_steps, 'unit': 'iterations/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * t_steps / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'Zellner linear regression N=%d D=%d S=%d, %s via HilbertCoreset (BASELINE configs[3])'
                                   % (N, D, S, args.alg.upper()),
                       'N': N, 'D': D, 'S': S, 'M': total, 'rows_per_gpu': n_local, 'parallelism': 'rows/%d' % world,
                       'exchange': exchange_kind,
                       'sweep': '%s pre-filter + exact fp64 rescoring (bit-identical selections)' % pname if pref else 'fp64'},
            'roofline': {'kernel': ('%s<%s> (K3 %s pre-filter sweep; winners rescored in fp64, selections '
                                    'identical to the fp64 sweep)' % (kname, 'GIGA' if args.alg == 'giga' else 'dot', pname)
                                    if pref else 'k_sweep<%s> (K3 score+argmax)' % ('GIGA' if args.alg == 'giga' else 'dot')),
                         'fp64_formulation_bytes_per_launch': 8.0 * n_local * S + 8.0 * n_local,
                         'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'bytes_per_launch': k3_bytes, 'avg_launch_ms': k3_ms_per, 'launches': args.steps,
                         'launches_timed': k3_n},
            'projection': {'points_dims_per_s': N * D / t_proj, 'ms': 1e3 * t_proj, 'kernel_ms': k1_ms_per,
                           'roofline_hbm': {'achieved': k1_bytes / (k1_ms_per * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                                            'unit': 'GB/s', 'frac': k1_bytes / (k1_ms_per * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            'bytes_per_launch': k1_bytes},
                           'roofline_fp64_mfma': {'achieved': k1_flops / (k1_ms_per * 1e-3) / 1e12,
                                                  'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                                                  'frac': k1_flops / (k1_ms_per * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF}},
            'posterior_gram': {'kernel_ms': k4_ms / max(k4_n, 1), 'wall_ms': 1e3 * t_post,
                               'tflops': 2.0 * n_local * (D + 1) * (D + 1) / max(k4_ms / max(k4_n, 1), 1e-9) / 1e9,
                               'note': 'K4 X^T W X on fp64 MFMA, all local rows, w = 1 (sampler set-up, untimed)'},
            'prefilter': dict(zip(('sweeps', 'candidates_rescored', 'fp64_fallbacks'), alg.snnls._eng.prefilter_stats())),
            'solver_init_ms': 1e3 * t_init, 'setup_s': t_setup,
            'coreset': {'size': int(len(idcs)), 'error': err, 'failed_steps': int(st_tr.sum())},
        }

    # ---------------- CPU baseline (rank 0, N=1 launch only): the NumPy oracle on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import RefGIGA, RefFrankWolfe, models_ref, coreset_ref
        ns = min(args.cpu_sample, N)
        Zs = Z[:ns].cpu().numpy()
        try:
            from threadpoolctl import threadpool_info
            thr = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
        except Exception:
            thr = os.cpu_count()
        t0 = time.perf_counter()
        phi_ref = np.empty((ns, S))
        for a in range(0, ns, 100_000):        # rows are independent: chunking does not change the result
            phi_ref[a:a + 100_000] = coreset_ref.project(lambda z, t: models_ref.linreg_loglik(z, t, 1.0),
                                                         Zs[a:a + 100_000], theta)
        t_cproj = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref = (RefGIGA if args.alg == 'giga' else RefFrankWolfe)(phi_ref.T, phi_ref.sum(axis=0))
        t_cinit = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref.build(args.cpu_iters)
        t_cit = time.perf_counter() - t0
        # the same sample through the device path: selections must match the oracle exactly
        hs = bc.HilbertCoreset(Zs, bc.DeviceProjector(lambda n, w, p: theta, S, model, ctx=ctx), snnls=cls)
        hs.build(args.cpu_iters, args.cpu_iters)
        dsel = hs.snnls._eng.trace()[0]
        rsel = np.array([t[0] for t in ref.trace])
        ridx = np.where(ref.w > 0)[0]
        parity = bool(np.array_equal(dsel, rsel) and np.array_equal(hs.idcs, ridx)
                      and np.allclose(hs.wts, ref.w[ridx], rtol=1e-5))
        out['cpu_baseline'] = {
            'value': args.cpu_iters / t_cit, 'unit': 'iterations/s', 'cores': int(thr), 'kind': 'port',
            'sample': 'first %d of the %d rows (same data, same Theta): NumPy oracle projection once, GIGA init, %d '
                      'greedy iterations; cost is linear in N, so the full-size rate is ~%.3f iterations/s'
                      % (ns, N, args.cpu_iters, args.cpu_iters / t_cit * ns / N),
            'projection_points_dims_per_s': ns * D / t_cproj, 'projection_s': t_cproj, 'init_s': t_cinit,
            'host_cpus': os.cpu_count(), 'numpy': np.__version__,
            'parity_on_sample': 'ok: %d selections identical, weights within 1e-5' % len(rsel) if parity else 'MISMATCH',
        }
        if not parity:
            out['parity_failure'] = {'device': dsel.tolist(), 'oracle': rsel.tolist()}
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1 or force_xchg:
        dist.barrier()
        comm.close()                      # the library's own RCCL communicator, before torch's
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
