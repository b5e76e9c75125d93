#!/usr/bin/env python3
"""bench.py -- greedy coreset iterations/s on the BASELINE.json workload.

Workload (BASELINE.json configs[3], the one `metric` is quoted on; it fits one GPU):
Zellner linear regression, N = 10M rows, D = 128, S = 100 posterior samples, GIGA via
HilbertCoreset.  With --gpus G the N rows are sharded over G ranks (strong scaling: the
total N is fixed), one process per GPU, one candidate-record all-gather (RCCL) per step.

A "step" = one greedy iteration of SparseNNLS.build (select + reweight + monotone guard)
over Phi resident in HBM.  The one-off K1 projection is timed separately and reported
as points*dims/s.  Prints ONE JSON line on rank 0 -- kept short (~5 KB) so that the driver's
record holds all of it; the evidence sits inside the two objects that record keeps whole:
  roofline            the dominant kernel of a step: the sweep.  From 2M rows per shard that is `k_sweep_i4` (two-level pre-filter:
                      a 4-bit mirror of the normalised rows streamed at half a byte per element, the rows it cannot exclude
                      re-bounded from int8 records in the same launch, the handful left rescored in fp64 -- the row returned is the
                      fp64 sweep's); `achieved` = its algorithmic bytes (54 B per row at S = 100) / its average launch time (HIP
                      events on the launch stream), `traffic` = HBM bytes per launch from the PMC passes under profiles/;
                      `prefilter` = {form, sweeps, rows rescored exactly, fallbacks, levels: rows the 4-bit level passed on}
  roofline.kernels    one entry per timed kernel (avg ms by HIP events, GB / GF per launch, fractions of the HBM / fp64-MFMA peaks)
  roofline.loops      the beta-Cores gradient loops of configs 2, 3 (logistic + Laplace sampler) and 4, per-phase split
  roofline.from_host  ndarray -> resident rows / first iteration / M = 100 coreset (upload + K1 pipelined); never part of `value`
  roofline.fp64_formulation   SURVEY 8(d)'s kernel (fp64 Phi streamed once per step) timed in the same run: {ms, frac, it_s, GB, same_sel}
  ms_per_step_M100    steps 2..100 of the coreset built in the from_host leg (list lengths 1..100), wall clock per step
  cpu_baseline        the NumPy oracle on the same host: greedy loop on ALL rows (all BLAS threads, and `one_thread`), the host's
                      CPU model string, and the gradient loops (`loops`)
Everything verbose (full roofline objects, phase timers, solver_init parts) goes to --detail FILE, or to one stderr
line tagged BENCH_DETAIL.

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
    ap.add_argument('--cpu-sample', type=int, default=1_000_000, help='rows given to the device-vs-oracle parity leg')
    ap.add_argument('--cpu-iters', type=int, default=20, help='greedy iterations of the parity leg')
    ap.add_argument('--cpu-full-iters', type=int, default=6,
                    help='greedy iterations of the CPU baseline on ALL rows (its K1 runs row-chunked; SURVEY 8d allows a cap of 20)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the fp64-sweep leg and the other BASELINE configs')
    ap.add_argument('--no-host', action='store_true', help='skip the from-host leg (ndarray -> upload + K1 pipelined -> first iteration)')
    ap.add_argument('--detail', default=None, help='write the full (verbose) result object to this file instead of stderr')
    ap.add_argument('--proj-reps', type=int, default=5)
    ap.add_argument('--proj-warmup', type=int, default=3)
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks ourselves -- fresh processes, before
    this one has touched the GPU -- through torch.distributed.run (one rank per GPU over RCCL), relay rank 0's JSON
    line and exit with the job's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return ' '.join(line.split(':', 1)[1].split())
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def sig(x, n=4):
    """n significant digits (the JSON line is meant to be short enough to survive in the driver's record)"""
    if x is None or isinstance(x, (bool, str)):
        return x
    if isinstance(x, int):
        return x
    x = float(x)
    if x != x or x in (float('inf'), float('-inf')):
        return None                     # (strict JSON has no NaN / inf)
    if x == 0.:
        return x
    return float('%.*g' % (n, x))


def kentry(name, ms, nbytes, flops, launches):
    """One line of roofline.kernels: average launch time (HIP events on the launch stream), algorithmic bytes and flops per
    launch, and what they make of the HBM / fp64-MFMA peaks."""
    e = {'k': name, 'ms': sig(ms), 'GB': sig(nbytes / 1e9), 'GF': sig(flops / 1e9) if flops else 0, 'n': int(launches),
         'hbm': sig(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None}
    if flops:
        e['mfma'] = sig(flops / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF) if ms > 0 else None
    return e


def newest_traffic(N, D, S, world, key):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this process): the
    newest profiles/rNN_pmc_traffic.json that was collected on exactly this workload shape."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic.json')), reverse=True):
        try:
            tj = json.load(open(path))
            if tj['config'] == {'N': N, 'D': D, 'S': S, 'n_gpus': world} and key in tj:
                return tj[key]['traffic_bytes_per_launch'], 'profiles/' + os.path.basename(path)
        except Exception:
            pass
    return None, None


def fp64_sweep_leg(bc, ctx, alg, cls, barrier, n_local, S, args):
    """The same solver state advanced with BC_PREFILTER=0: every step streams the fp64 Phi once (k_sweep)."""
    phi = alg.snnls._eng.phi
    old = os.environ.get('BC_PREFILTER')
    os.environ['BC_PREFILTER'] = '0'
    try:
        ref = cls(phi.T, alg.snnls.b)
    finally:
        if old is None:
            os.environ.pop('BC_PREFILTER', None)
        else:
            os.environ['BC_PREFILTER'] = old
    steps = min(args.steps, 40)
    ref.build(5)
    barrier()
    ctx.enable_timing(TIMING_STRIDE)
    ctx.kernel_time_reset()
    t0 = time.perf_counter()
    ref.build(steps)
    barrier()
    dt = time.perf_counter() - t0
    ms, n = ctx.kernel_time(0)
    per = ms / max(n, 1)
    byt = 8.0 * n_local * S + 8.0 * n_local
    # the int8 run made the same picks for as long as both ran
    a, b = ref._eng.trace()[0], alg.snnls._eng.trace()[0]
    m = min(len(a), len(b))
    return {'kernel': 'k_sweep<GIGA> (fp64 Phi streamed once per step)' if cls.__name__ == 'GIGA' else 'k_sweep<dot>',
            'iterations_per_s': steps / dt, 'ms_per_step': 1e3 * dt / steps, 'avg_launch_ms': per, 'steps': steps,
            'bytes_per_launch': byt, 'achieved_GBps': byt / (per * 1e-3) / 1e9 if per > 0 else 0.0,
            'frac_of_hbm_peak': byt / (per * 1e-3) / 1e9 / HBM_PEAK_GBS if per > 0 else 0.0,
            'same_selections_as_prefiltered_run': bool(np.array_equal(a[:m], b[:m])), 'compared_steps': int(m)}


def k4_entry(n, d, ms):
    """K4 numbers that cannot exceed 1: the kernel computes the upper-triangular BT x BT tiles only, and of a DIAGONAL tile
    only the 16 x 16 MFMA sub-tiles on or above its diagonal (bc_gram.hip: 36 of 64 at BT = 128, 10 of 16 at BT = 64), so the
    matrix pipe executes  (off-diagonal tiles + diagonal tiles * 36/64) * 2*N*BT^2  flop, not the 2*N*(D+1)^2 of the full Gram
    matrix; `ms` covers the Gram kernel AND its two-level split-order reduction (bc_timer brackets all three launches)."""
    bt = 128 if d > 64 else 64
    nt = -(-d // bt)
    ntri = nt * (nt + 1) // 2
    ns = bt // 16
    diag_share = (ns * (ns + 1) // 2) / float(ns * ns)
    executed = 2.0 * n * bt * bt * ((ntri - nt) + nt * diag_share)
    full = 2.0 * n * (d + 1) * (d + 1)
    return {'kernel_ms (gram + reduce)': ms, 'tile': bt, 'tiles_computed': ntri, 'tiles_full_square': nt * nt,
            'diagonal_tiles': nt, 'mfma_subtiles_per_diagonal_tile': '%d of %d' % (ns * (ns + 1) // 2, ns * ns),
            'executed_tflops': executed / (ms * 1e-3) / 1e12,
            'frac_of_fp64_mfma_peak': executed / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF,
            'symmetry_factor': full / executed,
            'effective_tflops_of_full_gram': full / (ms * 1e-3) / 1e12,
            'bytes_per_launch': 8.0 * n * (d + 1), 'hbm_frac': 8.0 * n * (d + 1) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def _time_k1(ctx, call, barrier, reps=8, warm=3):
    for _ in range(warm):
        call()
    barrier()
    ctx.enable_timing(True)
    ctx.kernel_time_reset()
    r = None
    for _ in range(reps):
        r = None
        r = call()
    barrier()
    ms, n = ctx.kernel_time(1)
    return ms / max(n, 1), r


def _k1_entry(name, n, d, dz, s, ms, model):
    byt = 8.0 * n * dz + 8.0 * n * s
    fl = 2.0 * n * d * s
    return {'config': name, 'model': model, 'N': n, 'D': d, 'S': s, 'kernel_ms': ms,
            'points_dims_per_s': n * d / (ms * 1e-3),
            'roofline_hbm': {'achieved': byt / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'bytes_per_launch': byt},
            'roofline_fp64_mfma': {'achieved': fl / (ms * 1e-3) / 1e12, 'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                                   'frac': fl / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF}}


def other_configs(torch, bc, ctx, dev, barrier, no_cpu=False):
    """K1 (and K4 for config 5) on the other BASELINE configs, synthetic inputs as SURVEY 8(d) specifies, S = 100."""
    S = 100
    res = []
    g = torch.Generator(device=dev)

    def linreg_data(n, d, seed):
        g.manual_seed(seed)
        X = torch.randn((n, d), generator=g, dtype=torch.float64, device=dev)
        th = torch.randn((d,), generator=g, dtype=torch.float64, device=dev)
        y = X @ th + torch.randn((n,), generator=g, dtype=torch.float64, device=dev)
        return torch.cat((X, y[:, None]), dim=1)

    # config 2: Zellner linear regression N = 1M, D = 64
    n, d = 1_000_000, 64
    Z = linreg_data(n, d, 20)
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    mu, L, _ = bc.weighted_post(np.zeros(d), np.eye(d), 1.0, data, None)
    theta = mu + np.random.default_rng(20).standard_normal((S, d)).dot(L.T)
    prj = bc.DeviceBetaProjector(lambda k, w, p: theta, S, bc.likelihoods.LinearRegression(1.0), ctx=ctx)
    ms, _ = _time_k1(ctx, lambda: prj.project(data), barrier)
    res.append(_k1_entry('configs[1] linreg N=1M D=64', n, d, d + 1, S, ms, 'log-likelihood (model_linreg.py:4-10)'))
    ms, _ = _time_k1(ctx, lambda: prj.project_f(data, 0.1), barrier)
    res.append(_k1_entry('configs[1] linreg N=1M D=64', n, d, d + 1, S, ms, 'beta-likelihood, beta = 0.1 (model_neurlinr.py:102-110)'))
    del prj
    beta_legs = beta_coreset_leg(bc, ctx, barrier, data, 'configs[1] Zellner linreg N=1M D=64', oracle_rows=None if no_cpu else 100_000,
                                 Z_host=None if no_cpu else Z[:100_000].cpu().numpy())
    del data, Z
    # config 3: Zellner logistic regression N = 1M, D = 128
    n, d = 1_000_000, 128
    g.manual_seed(30)
    X = torch.randn((n, d), generator=g, dtype=torch.float64, device=dev)
    thstar = torch.full((d,), 1.0 / np.sqrt(d), dtype=torch.float64, device=dev)
    yl = torch.where(torch.rand((n,), generator=g, dtype=torch.float64, device=dev) < torch.sigmoid(X @ thstar), 1.0, -1.0)
    Zl = X * yl[:, None]
    del X
    data = bc.DeviceData.from_torch(Zl, ctx=ctx)
    theta = thstar.cpu().numpy() + 0.1 * np.random.default_rng(30).standard_normal((S, d))
    prj = bc.DeviceBetaProjector(lambda k, w, p: theta, S, bc.likelihoods.LogisticRegression(), ctx=ctx)
    ms, _ = _time_k1(ctx, lambda: prj.project(data), barrier)
    res.append(_k1_entry('configs[2] logistic N=1M D=128', n, d, d, S, ms, 'log-likelihood (model_lr.py:72-79)'))
    ms, _ = _time_k1(ctx, lambda: prj.project_f(data, 0.1), barrier)
    res.append(_k1_entry('configs[2] logistic N=1M D=128', n, d, d, S, ms, 'beta-likelihood, beta = 0.1 (model_lr.py:81-86)'))
    del prj
    # ... and config 3's actual driver: BetaCoreset on the beta-tempered sigmoid score with the Laplace sampler
    # (a) the reference's wiring to the letter: scipy's BFGS finds the Laplace mode -- hundreds of ms per sampler call at weights
    # N/M, so the short form; (b) the same posterior through bc.samplers' Newton solver (same unique mode, ~1e-6 apart in theta)
    beta_legs += beta_coreset_leg(bc, ctx, barrier, data, 'configs[2] Zellner logistic N=1M D=128', kind='logistic', grads=4, sizes=(100,),
                                  extras=False, oracle_rows=None if no_cpu else 50_000, Z_host=None if no_cpu else Zl[:50_000].cpu().numpy())
    beta_legs += beta_coreset_leg(bc, ctx, barrier, data, 'configs[2] Zellner logistic N=1M D=128', kind='logistic', grads=20,
                                  solver='newton')
    del data, Zl
    # config 5: neural-linear last layer, N = 2M, D = 512 random ReLU features; K4 on all rows, then K1
    n, d = 2_000_000, 512
    g.manual_seed(50)
    U = torch.randn((n, 32), generator=g, dtype=torch.float64, device=dev)
    G = torch.randn((32, d), generator=g, dtype=torch.float64, device=dev) / np.sqrt(32.)
    Z = torch.empty((n, d + 1), dtype=torch.float64, device=dev)
    Z[:, :d] = torch.relu(U @ G)
    th = torch.randn((d,), generator=g, dtype=torch.float64, device=dev)
    Z[:, d] = Z[:, :d] @ th + torch.randn((n,), generator=g, dtype=torch.float64, device=dev)
    del U, G
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    w = np.random.default_rng(50).uniform(0., 2., n)
    k4 = {}
    for nm, ww in (('w = 1', None), ('w ~ U(0, 2)', w)):
        bc.weighted_gram(data, ww)
        barrier()
        ctx.enable_timing(True)
        ctx.kernel_time_reset()
        for _ in range(3):
            bc.weighted_gram(data, ww)
        barrier()
        ms4, n4 = ctx.kernel_time(2)
        ms4 /= max(n4, 1)
        k4[nm] = k4_entry(n, d, ms4)
    mu, L, _ = bc.weighted_post(np.zeros(d), np.eye(d), 1.0, data, None)
    theta = mu + np.random.default_rng(51).standard_normal((S, d)).dot(L.T)
    prj = bc.DeviceBetaProjector(lambda k, ww_, p: theta, S, bc.likelihoods.LinearRegression(1.0), ctx=ctx)
    ms, _ = _time_k1(ctx, lambda: prj.project(data), barrier)
    e = _k1_entry('configs[4] neural-linear last layer N=2M D=512', n, d, d + 1, S, ms, 'log-likelihood (model_neurlinr.py:90-97)')
    e['posterior_gram_K4'] = k4
    res.append(e)
    del prj, data, Z
    torch.cuda.empty_cache()
    return res, beta_legs


class NativeCallTimer:
    """Wall time per C entry point while active (patches beta_cores_amd._native.call): the breakdown of solver_init."""

    def __init__(self, bc):
        import beta_cores_amd._native as N
        self.N = N
        self.acc = {}

    def __enter__(self):
        import beta_cores_amd.device as dv
        import beta_cores_amd.coreset.projector as pj
        import beta_cores_amd.snnls.engine as en
        self.mods = [m for m in (self.N, dv.N, pj.N, en.N) if m is not None]
        self.orig = self.N.call
        acc, orig = self.acc, self.orig

        def timed(name, *a):
            t0 = time.perf_counter()
            try:
                return orig(name, *a)
            finally:
                acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
        self.N.call = timed
        return self

    def __exit__(self, *exc):
        self.N.call = self.orig
        return False

    def ms(self, name):
        return 1e3 * self.acc.get(name, 0.0)

    def total_ms(self):
        return 1e3 * sum(self.acc.values())

    def other_ms(self, names):
        return 1e3 * sum(v for k, v in self.acc.items() if k not in names)


def beta_coreset_leg(bc, ctx, barrier, data, name, sizes=(10, 100), grads=50, opt_seed=7, oracle_rows=None, Z_host=None,
                     kind='linreg', solver='bfgs', extras=True):
    """The beta-Cores gradient loop (bcores.py:141-150, BetaCoreset._optimize with n_subsample_* = None, learn_beta = False,
    beta = 0.1): per gradient  sampler (weighted_post on the <= M coreset rows, model_linreg.py:25-34) -> K1 over ALL data
    rows (store-free: only vecs.sum(axis=0) is needed) -> column sums -> M x S algebra -> one ADAM step on the host.
    The coreset is pre-initialised with M rows (as the reference's drivers do with wts / idcs / pts,
    zellner_neural_linear/main.py:147-149) and `_optimize()` runs `grads` gradients."""
    n, dz = data.shape
    d = dz - 1 if kind == 'linreg' else dz
    S = 100
    th0, Sig0inv = np.zeros(d), np.eye(d)
    out = []
    # kind = 'logistic' (BASELINE configs[2]): the beta-tempered sigmoid score (model_lr.py:81-86) with the reference's own
    # sampler wiring, the Laplace approximation of the weighted coreset posterior (zellner_logreg/main.py:139-144,
    # util/opt.py:9-33 -> bc.samplers.LogisticLaplaceSampler: scipy's BFGS + a D x D Cholesky on the <= M coreset rows, host)
    model = bc.likelihoods.LinearRegression(1.0) if kind == 'linreg' else bc.likelihoods.LogisticRegression()
    for M in sizes:
        rng = np.random.RandomState(opt_seed + M)
        idcs = np.sort(rng.choice(n, M, replace=False)).astype(np.int64)
        pts = data.rows(idcs)
        srng = np.random.RandomState(1000 + M)
        t_samp = [0.0, 0]

        # zellner_neural_linear/main.py:119-124 (the weighted_post form) as bc.samplers.LinregPosteriorSampler: the same
        # arithmetic and draws; its prefetch() draws the next call's normals while the GPU works on the current gradient
        if kind == 'linreg':
            base_sampler = bc.samplers.LinregPosteriorSampler(th0, Sig0inv, 1.0, rng=srng, ctx=ctx)
        else:
            base_sampler = bc.samplers.LogisticLaplaceSampler(th0, diag=False, rng=srng, solver=solver)

        class TimedSampler:
            prefetch = staticmethod(base_sampler.prefetch)
            scope = staticmethod(base_sampler.scope)

            def __call__(self, k, wts, pts_):
                t0 = time.perf_counter()
                r = base_sampler(k, wts, pts_)
                t_samp[0] += time.perf_counter() - t0
                t_samp[1] += 1
                return r
        sampler_w = TimedSampler()
        prj = bc.DeviceBetaProjector(sampler_w, S, model, ctx=ctx)
        sched = lambda i: 0.01 / (1. + i)

        def make(fused):
            return bc.BetaCoreset(data, prj, opt_itrs=grads, step_sched=sched, beta=0.1, learn_beta=False,
                                  wts=np.full(M, float(n) / M), idcs=idcs.copy(), pts=pts.copy(), fused_gradient=fused)
        alg = make(True)
        t_call = [0.0, 0]
        orig = prj.vi_gradient

        def timed_call(*a, **kw):
            t1 = time.perf_counter()
            r = orig(*a, **kw)
            t_call[0] += time.perf_counter() - t1
            t_call[1] += 1
            return r
        prj.vi_gradient = timed_call
        ctx.enable_timing(0)
        alg.opt_itrs = 5 if extras else 2
        alg._optimize()                                    # warm-up: buffers, code objects
        alg.opt_itrs = grads
        barrier()
        t_samp[0], t_samp[1], t_call[0], t_call[1] = 0.0, 0, 0.0, 0
        if not extras:
            # short form (a sampler that takes hundreds of ms per call): the kernel timer runs inside the timed pass -- its event
            # pairs (~11 us of stream time) vanish next to such a gradient -- and the comparison passes are skipped
            ctx.enable_timing(1)
            ctx.kernel_time_reset()
        t0 = time.perf_counter()
        alg._optimize()
        barrier()
        t_fused = (time.perf_counter() - t0) / grads
        samp_ms = 1e3 * t_samp[0] / max(t_samp[1], 1)
        call_ms = 1e3 * t_call[0] / max(t_call[1], 1)
        fused_calls = t_call[1]
        ph, t_instr, t_mat = {}, float('nan'), float('nan')
        if extras:
            # instrumented pass: HIP events between the phases of the native call (they cost stream time: not the timed pass)
            ctx.enable_timing(1)
            ctx.kernel_time_reset()
            ctx.phase_times(reset=True)
            t0 = time.perf_counter()
            alg._optimize()
            barrier()
            t_instr = (time.perf_counter() - t0) / grads
            ph, ncalls = ctx.phase_times(reset=True)
            ph = {k: v / max(ncalls, 1) for k, v in ph.items()}
        k1_ms, k1_n = ctx.kernel_time(1)
        ctx.enable_timing(0)
        prj.vi_gradient = orig
        k1 = k1_ms / max(k1_n, 1)
        alg_m = None
        if extras:
            # the general path for comparison: every gradient materialises Phi (8*N*S more bytes) and reads back its column sums
            alg_m = make(False)
            alg_m.opt_itrs = 3
            alg_m._optimize()
            gm = max(10, grads // 5)
            alg_m.opt_itrs = gm
            barrier()
            t0 = time.perf_counter()
            alg_m._optimize()
            barrier()
            t_mat = (time.perf_counter() - t0) / gm
        # one whole build step (select: materialised K1 + K3 sweep, then `grads` gradients)
        alg.opt_itrs = grads
        barrier()
        t0 = time.perf_counter()
        alg.build(1, M + 1)
        barrier()
        t_build = time.perf_counter() - t0
        byt = 8.0 * n * dz
        fl = 2.0 * n * d * S
        e = {'config': name, 'N': n, 'D': d, 'S': S, 'M': M, 'beta': 0.1, 'gradients': grads,
             'ms_per_gradient': 1e3 * t_fused,
             'k1_store_free_kernel_ms': k1,
             'non_k1_fraction': max(0.0, 1.0 - k1 / (1e3 * t_fused)),
             'native_gradient_calls': fused_calls,
             'model': kind if kind == 'linreg' else '%s/%s' % (kind, solver),
             'breakdown_ms': {'sampler (host LAPACK + K4 on the M coreset rows; logistic: Laplace fit, scipy BFGS)': samp_ms,
                              'native gradient call, wall (Theta upload, K1 x2, column sums, M x S algebra, one sync)': call_ms,
                              'ADAM step + Python glue': max(0.0, 1e3 * t_fused - samp_ms - call_ms),
                              'gpu_phases_inside_the_call (HIP events, separate instrumented pass)': ph,
                              'instrumented_pass_ms_per_gradient': 1e3 * t_instr},
             'roofline_hbm': {'achieved': byt / (k1 * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': byt / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 'bytes_per_launch': byt,
                              'note': 'algorithmic bytes 8*N*Dz: Z read once, Phi never written'},
             'roofline_fp64_mfma': {'achieved': fl / (k1 * 1e-3) / 1e12, 'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                                    'frac': fl / (k1 * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF},
             'whole_gradient_vs_hbm': byt / t_fused / 1e9 / HBM_PEAK_GBS,
             'materialising_path_ms_per_gradient': 1e3 * t_mat,
             'build_step_ms (select + %d gradients)' % grads: 1e3 * t_build}
        out.append(e)
        del alg, alg_m, prj
    if oracle_rows and Z_host is not None:
        # the same loop through the NumPy oracle on the host, on the first `oracle_rows` rows (row-capped: K1 on the CPU is
        # ~10 s per million rows and gradient)
        from oracle import models_ref, coreset_ref
        Zs = Z_host[:oracle_rows]
        M = sizes[-1]
        rng = np.random.RandomState(opt_seed + M)
        idc = np.sort(rng.choice(oracle_rows, M, replace=False)).astype(np.int64)
        srng = np.random.RandomState(1000 + M)

        def samp(w, p):
            if p.shape[0] == 0:
                w, p = np.zeros(1), np.zeros((1, dz))
            if kind == 'linreg':
                mu, L, _ = models_ref.linreg_weighted_post(th0, Sig0inv, 1.0, p, w)
            else:
                mu, L, _ = models_ref.logistic_laplace(w, p, th0, False)
            return mu + srng.randn(S, d).dot(L.T)
        g_cpu = 4
        blik = (lambda z, t, b: models_ref.linreg_beta_lik(z, t, b, 1.0)) if kind == 'linreg' else models_ref.logistic_beta_lik
        ref = coreset_ref.RefGreedyVI(Zs, lambda p_, th: coreset_ref.project_f(blik, p_, th, 0.1),
                                      samp, g_cpu, lambda i: 0.01 / (1. + i), wts=np.full(M, float(oracle_rows) / M), idcs=idc, pts=Zs[idc])
        t0 = time.perf_counter()
        ref.optimize()
        t_cpu = (time.perf_counter() - t0) / g_cpu
        out.append({'cpu_baseline': {'kind': 'port', 'config': name, 'rows': int(oracle_rows), 'M': M, 'gradients': g_cpu,
                                     'ms_per_gradient': 1e3 * t_cpu,
                                     'ms_per_gradient_scaled_to_N': 1e3 * t_cpu * n / oracle_rows,
                                     'sample': 'NumPy oracle (oracle/coreset_ref.RefGreedyVI.optimize) on the first %d of %d rows, '
                                               '%d gradients; K1 on the host is linear in the rows' % (oracle_rows, n, g_cpu)}})
    return out


def from_host_leg(bc, ctx, barrier, Z_host, theta, S, model, cls, f_tr, args):
    """What a drop-in user sees: `HilbertCoreset(ndarray, projector)` (hilbert.py:11-17) from a HOST array.  Wall time from the
    ndarray to (a) the rows resident in HBM, (b) the first greedy iteration, (c) an M = 100 coreset; the upload goes through the
    pinned-staging uploader (csrc/bc_upload.hip) and K1 runs on chunk c while chunks c+1.. are on the wire
    (bc_project_from_host).  `plain`: the same array through ONE hipMemcpyAsync from pageable memory (BC_UPLOAD_THREADS=0)."""
    import gc
    n, dz = Z_host.shape
    gb = 8.0 * n * dz / 1e9
    res = {'GB': sig(gb)}

    def timed_upload(threads):
        old = os.environ.get('BC_UPLOAD_THREADS')
        if threads is not None:
            os.environ['BC_UPLOAD_THREADS'] = str(threads)
        try:
            barrier()
            t0 = time.perf_counter()
            dd = bc.DeviceData(Z_host, ctx=ctx)
            barrier()
            return time.perf_counter() - t0, dd
        finally:
            if threads is not None:
                if old is None:
                    os.environ.pop('BC_UPLOAD_THREADS', None)
                else:
                    os.environ['BC_UPLOAD_THREADS'] = old
    t_up, dd = timed_upload(None)          # first call: allocates the staging buffers, copy streams, events
    del dd
    t_up, dd = timed_upload(None)
    del dd
    gc.collect()
    # the pieces: the device allocation by itself, and the copy into an existing buffer (direct = hipMemcpyAsync from the
    # pageable array, staged = 8 host threads through pinned staging buffers)
    barrier()
    t0 = time.perf_counter()
    slot = bc.DeviceData.slot(dz, cap_rows=n, ctx=ctx)
    barrier()
    t_alloc = time.perf_counter() - t0
    t_copy = {}
    for nm, thr in (('direct', '0'), ('staged8', '8'), ('direct', '0')):
        old = os.environ.get('BC_UPLOAD_THREADS')
        os.environ['BC_UPLOAD_THREADS'] = thr
        try:
            barrier()
            t0 = time.perf_counter()
            slot.update(Z_host)
            barrier()
            t_copy[nm] = time.perf_counter() - t0
        finally:
            if old is None:
                os.environ.pop('BC_UPLOAD_THREADS', None)
            else:
                os.environ['BC_UPLOAD_THREADS'] = old
    del slot
    gc.collect()
    res.update({'upload_ms': sig(1e3 * t_up), 'upload_GBps': sig(gb / t_up), 'alloc_ms': sig(1e3 * t_alloc),
                'copy_direct_GBps': sig(gb / t_copy['direct']), 'copy_staged8_GBps': sig(gb / t_copy['staged8'])})
    prj = bc.DeviceProjector(lambda k, w, p: theta, S, model, ctx=ctx)
    M = 100
    for rep in range(2):                   # the second pass is the one reported (buffers and code objects exist)
        barrier()
        t0 = time.perf_counter()
        alg = bc.HilbertCoreset(Z_host, prj, snnls=cls)
        barrier()
        t_init = time.perf_counter() - t0
        alg.build(1, M)
        barrier()
        t_first = time.perf_counter() - t0
        alg.build(M - 1, M)
        barrier()
        t_m = time.perf_counter() - t0
        tr = alg.snnls._eng.trace()[0]
        if rep == 0:
            del alg
            gc.collect()
    k = min(len(tr), len(f_tr))
    res.update({'construct_ms': sig(1e3 * t_init), 'first_iter_ms': sig(1e3 * t_first), 'M100_ms': sig(1e3 * t_m),
                'same_trace_as_resident_run': bool(np.array_equal(tr[:k], f_tr[:k])), 'compared_steps': int(k)})
    return res


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))
    # ONE JSON line on stdout: libraries that print banners there (RCCL writes its version block at communicator creation)
    # are pointed at stderr until the line is ready
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE = %d but --gpus %d' % (world, args.gpus))
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
    # ... and warm: three more launches over the same rows (the first one above pays the scratch allocation and cold caches)
    k4_warm_ms = None
    if world == 1:
        ctx.kernel_time_reset()
        for _ in range(3):
            bc.weighted_gram(data, None)
        barrier()
        k4w, k4wn = ctx.kernel_time(2)
        k4_warm_ms = k4w / max(k4wn, 1)
        ctx.kernel_time_reset()
    model = bc.likelihoods.LinearRegression(1.0)
    prj = bc.DeviceProjector(lambda n, w, p: theta, S, model, ctx=ctx)

    # ---------------- K1: projection, points*dims/s
    import gc
    gc.collect()
    gc.disable()          # a gen-2 collection (tens of ms with torch loaded) must not land in a timed region
    ctx.enable_timing(True)
    for _ in range(max(1, args.proj_warmup)):
        prj.project(data)                  # warm-up (also allocates Phi): the first launches after the set-up kernels run
    barrier()                              # 5-15 % slower than the steady state (7.0, 7.0, 6.3, 6.1, 6.0 ms in one trace)
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
    phi = None                             # hand the 8 GB Phi of the projection runs back to the projector's pool: the coreset's
    barrier()                              # own projection reuses it instead of hipMalloc'ing a second one inside the timed init
    ctx.enable_timing(True)
    ctx.kernel_time_reset()
    with NativeCallTimer(bc) as nct:
        t0 = time.perf_counter()
        alg = bc.HilbertCoreset(data, prj, snnls=cls, comm=comm)
        barrier()
        t_init = time.perf_counter() - t0
    k1i_ms, k1i_n = ctx.kernel_time(1)
    init_parts = {'phi_alloc (bc_phi_create)': nct.ms('bc_phi_create'),
                  'K1 kernel (HIP events)': k1i_ms,
                  'bc_project host side + column-sum / norm statistics (two small kernels, one sync)': max(0.0, nct.ms('bc_project') - k1i_ms),
                  'solver create: mirror allocation + k_build_i8 + state (bc_snnls_create)': nct.ms('bc_snnls_create'),
                  'other native calls': nct.other_ms(('bc_phi_create', 'bc_project', 'bc_snnls_create')),
                  'python / collectives': max(0.0, 1e3 * t_init - nct.total_ms())}
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
        form = int(alg.snnls._eng.prefilter_form)
        if pref == 8 and form == 3:
            # two-level form (csrc/bc_prefilter_i4.h): the sweep streams the 4-bit mirror -- k-groups of 8 samples per dword, padded
            # to a multiple of the batch (S = 100: 13 dwords, no padding) -- and a 16-bit (scale, delta) code per row; the rows it
            # cannot exclude (prefilter.levels below: a fraction of a percent) are re-bounded from 128-byte int8 records
            g8 = -(-S // 8)
            sp8 = min(-(-g8 // u) * u for u in (13, 8, 7, 6, 5))
            k3_bytes = 4.0 * n_local * sp8 + 2.0 * n_local
        elif pref == 8:
            # int8 mirror: one byte per element (k-groups of 4 padded to a multiple of 5) + (scale, delta) halfs per row
            k3_bytes = 4.0 * n_local * (-(-(-(-S // 4)) // 5) * 5) + 4.0 * n_local
        elif pref == 16:
            k3_bytes = 2.0 * n_local * (-(-S // 10) * 10) + n_local / 8.0
        else:
            k3_bytes = 4.0 * n_local * S + 8.0 * n_local + 4.0 * n_local
    else:
        k3_bytes = 8.0 * n_local * S + 8.0 * n_local      # one streaming read of Phi + the norms
    # ---------------- diagnostic pass (untimed): HIP events around every stage of every step -- sweep, rescoring / local
    # winner, candidate all-gather, finish -- so that a multi-GPU run shows where its step time goes.  The event pairs cost
    # stream time (~11 us each), which is why this is a pass of its own and not part of `value`.
    diag_steps = 30
    ctx.timing_classes(0x3f)
    ctx.enable_timing(1)
    ctx.kernel_time_reset()
    t0 = time.perf_counter()
    alg.snnls.build(diag_steps)
    barrier()
    t_diag = time.perf_counter() - t0
    stage_ms = {}
    for nm, cl in (('sweep', 0), ('rescoring_or_local_winner', 3), ('all_gather', 4), ('finish', 5)):
        ms_, n_ = ctx.kernel_time(cl)
        stage_ms[nm] = ms_ / n_ if n_ else None
    ctx.enable_timing(0)
    ctx.timing_classes(0x7)
    rccl_ranks = None
    if eng_.native_exchange:
        import ctypes as _C
        from beta_cores_amd import _native as _N
        rr, ww = _C.c_int32(), _C.c_int32()
        _N.call('bc_comm_info', comm.native_comm(ctx), _C.byref(rr), _C.byref(ww))
        rccl_ranks = int(ww.value)
    step_diag = {'stage_ms (HIP events, every step of a separate %d-step pass)' % diag_steps: stage_ms,
                 'instrumented_ms_per_step': 1e3 * t_diag / diag_steps, 'rccl_ranks': rccl_ranks,
                 'transport': exchange_kind,
                 'note': 'single rank with the pre-filter: rescoring runs inside the finish launch; '
                         'multi-rank: sweep -> rescoring -> ncclAllGather of one (S+4)-double record per rank -> replicated finish'}
    alg._pull()
    wts, pts, idcs = alg.get()
    err = alg.error()
    f_tr, st_tr, _ = alg.snnls._eng.trace()

    two_level = pref == 8 and int(alg.snnls._eng.prefilter_form) == 3
    pname = ('4-bit + int8 two-level' if two_level else 'int8') if pref == 8 else 'fp%d' % pref
    kname = ('k_sweep_i4' if two_level else 'k_sweep_i8') if pref == 8 else 'k_sweep_f%d' % pref
    algn = 'GIGA' if args.alg == 'giga' else 'dot'
    out, detail = None, {}
    kernels, loops = [], []
    if rank == 0:
        ach = k3_bytes / (k3_ms_per * 1e-3) / 1e9 if k3_ms_per > 0 else 0.0
        traffic, traffic_src = (None, None)
        if args.alg == 'giga':
            traffic, traffic_src = newest_traffic(N, D, S, world, (kname if pref else 'k_sweep'))
        # ---- roofline.kernels: every kernel this run timed, one short entry each (name | avg ms | GB, GF per launch | fractions)
        kernels.append(kentry('K1 %s linreg N=%d D=%d (headline shape)' % ('k_project_r' if n_local >= 262144 else 'k_project', n_local, D),
                              k1_ms_per, k1_bytes, k1_flops, k1_n))
        k4e = k4_entry(n_local, D, max(k4_warm_ms or (k4_ms / max(k4_n, 1)), 1e-9))
        kernels.append(dict(kentry('K4 k_gram+reduce N=%d D=%d w=1 %s' % (n_local, D, 'warm' if k4_warm_ms else 'cold'),
                                   k4e['kernel_ms (gram + reduce)'], k4e['bytes_per_launch'],
                                   k4e['executed_tflops'] * 1e9 * k4e['kernel_ms (gram + reduce)'], 3 if k4_warm_ms else k4_n),
                            sym=sig(k4e['symmetry_factor'])))
        out = {
            'metric': 'greedy coreset iterations/sec', 'value': args.steps / t_steps, 'unit': 'iterations/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * t_steps / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'Zellner linreg N=%d D=%d S=%d, %s via HilbertCoreset (BASELINE configs[3])' % (N, D, S, args.alg.upper()),
                       'N': N, 'D': D, 'S': S, 'M': total, 'rows_per_gpu': n_local, 'parallelism': 'rows/%d' % world,
                       'exchange': exchange_kind,
                       'sweep': '%s pre-filter + exact fp64 rescoring (bit-identical selections)' % pname if pref else 'fp64'},
            'roofline': {'kernel': ('%s<%s>: K3 %s mirror sweep, winners rescored in fp64' % (kname, algn, pname)) if pref
                                   else 'k_sweep<%s>: K3 score+argmax' % algn,
                         'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'bytes_per_launch': k3_bytes, 'fp64_formulation_bytes_per_launch': 8.0 * n_local * S + 8.0 * n_local,
                         'avg_launch_ms': k3_ms_per, 'launches': args.steps, 'launches_timed': k3_n,
                         'kernels': kernels, 'loops': loops},
            'prefilter': dict(zip(('sweeps', 'candidates_rescored', 'fp64_fallbacks'), alg.snnls._eng.prefilter_stats()),
                              form=int(alg.snnls._eng.prefilter_form),
                              levels=dict(zip(('l1_sweeps', 'rows_passed_on', 'rows_refined_int8'), alg.snnls._eng.prefilter_levels()))),
            'step_stages': {'sweep': sig(stage_ms['sweep']), 'rescore': sig(stage_ms['rescoring_or_local_winner']),
                            'gather': sig(stage_ms['all_gather']), 'finish': sig(stage_ms['finish']),
                            'instr_ms_per_step': sig(1e3 * t_diag / diag_steps), 'rccl_ranks': rccl_ranks,
                            'transport': exchange_kind.split(' (')[0]},
            'solver_init_ms': sig(1e3 * t_init),
            'solver_init': {'phi_alloc': sig(init_parts['phi_alloc (bc_phi_create)']), 'k1': sig(init_parts['K1 kernel (HIP events)']),
                            'stats': sig(init_parts['bc_project host side + column-sum / norm statistics (two small kernels, one sync)']),
                            'solver_create': sig(init_parts['solver create: mirror allocation + k_build_i8 + state (bc_snnls_create)']),
                            'other_native': sig(init_parts['other native calls']), 'python': sig(init_parts['python / collectives'])},
            'projection': {'points_dims_per_s': sig(N * D / t_proj, 6), 'ms': sig(1e3 * t_proj), 'kernel_ms': sig(k1_ms_per)},
            'coreset': {'size': int(len(idcs)), 'error': err, 'failed_steps': int(st_tr.sum())},
        }
        detail.update({'solver_init': init_parts, 'step_stages': step_diag, 'setup_s': t_setup, 'posterior_gram_cold_wall_ms': 1e3 * t_post,
                       'posterior_gram': k4e})

    # ---------------- the host path (hilbert.py:11 takes an ndarray): upload + K1 pipelined, wall time to the first iteration.
    # (Before the legs that run NumPy / BLAS on the host: after them the runtime's pinning of the pageable 10 GB source takes
    # 0.5-1 s longer on some boxes -- 1 026 against 191 ms for the same construction, profiles/r05_notes.md.)
    Z_host = None
    if rank == 0 and world == 1 and not args.no_host:
        Z_host = Z.cpu().numpy()
        fh = from_host_leg(bc, ctx, barrier, Z_host, theta, S, model, cls, f_tr, args)
        out['roofline']['from_host'] = fh
        # the survey's M = 100 run: steps 2..100 of the coreset built from the host array (list lengths 1..100, wall clock)
        out['ms_per_step_M100'] = sig((fh['M100_ms'] - fh['first_iter_ms']) / 99.0)

    # ---------------- the SURVEY 8(d) formulation, driver-timed too: the exact fp64 sweep (8*N*S + 8*N bytes per step)
    if rank == 0 and world == 1 and not args.no_extra:
        f64 = fp64_sweep_leg(bc, ctx, alg, cls, barrier, n_local, S, args)
        kernels.append(dict(kentry('K3 k_sweep<%s> fp64 Phi streamed once per step N=%d (SURVEY 8d formulation)' % (algn, n_local),
                                   f64['avg_launch_ms'], f64['bytes_per_launch'], 0, f64['steps']),
                            it_s=sig(f64['iterations_per_s']), same_sel=f64['same_selections_as_prefiltered_run']))
        detail['fp64_sweep'] = f64
        # ... and as an object of its own: SURVEY 8(d)'s formulation (fp64 Phi streamed once per step) timed in this very run
        out['roofline']['fp64_formulation'] = {'ms': sig(f64['avg_launch_ms']), 'frac': sig(f64['frac_of_hbm_peak']),
                                               'it_s': sig(f64['iterations_per_s']), 'GB': sig(f64['bytes_per_launch'] / 1e9),
                                               'same_sel': f64['same_selections_as_prefiltered_run']}
        oc, beta2 = other_configs(torch, bc, ctx, dev, barrier, no_cpu=args.no_cpu)
        detail['other_configs'] = oc
        for e in oc:
            nm = 'K1 %s %s' % (e['config'].split(' ', 1)[1], e['model'].split(' (')[0].split(',')[0])
            kernels.append(kentry(nm, e['kernel_ms'], e['roofline_hbm']['bytes_per_launch'], 2.0 * e['N'] * e['D'] * e['S'], 8))
            for wn, k4 in (e.get('posterior_gram_K4') or {}).items():
                kernels.append(dict(kentry('K4 k_gram+reduce N=%d D=%d %s' % (e['N'], e['D'], wn), k4['kernel_ms (gram + reduce)'],
                                           k4['bytes_per_launch'], k4['executed_tflops'] * 1e9 * k4['kernel_ms (gram + reduce)'], 3),
                                    sym=sig(k4['symmetry_factor'])))
        # the beta-Cores gradient loop (BetaCoreset._optimize, bcores.py:141-150) on configs 2, 3 (logistic + Laplace sampler) and 4
        beta_all = beta2 + beta_coreset_leg(bc, ctx, barrier, data, 'configs[3] Zellner linreg N=%d D=%d' % (N, D))
        detail['beta_coreset'] = beta_all
        cpu_loops = []
        for e in beta_all:
            if 'cpu_baseline' in e:
                c = e['cpu_baseline']
                cpu_loops.append({'loop': 'bcores ' + c['config'].split(' ', 1)[0], 'rows': c['rows'], 'M': c['M'], 'ms_grad': sig(c['ms_per_gradient']),
                                  'ms_grad_scaled_to_N': sig(c['ms_per_gradient_scaled_to_N'])})
                continue
            bd = e['breakdown_ms']
            samp = [v for k, v in bd.items() if k.startswith('sampler')][0]
            call = [v for k, v in bd.items() if k.startswith('native gradient call')][0]
            loops.append({'loop': 'bcores %s %s N=%d D=%d M=%d' % (e['config'].split(' ', 1)[0], e['model'], e['N'], e['D'], e['M']),
                          'ms_grad': sig(e['ms_per_gradient']), 'k1_ms': sig(e['k1_store_free_kernel_ms']), 'samp_ms': sig(samp),
                          'call_ms': sig(call), 'non_k1': sig(e['non_k1_fraction'], 3), 'mat_ms_grad': sig(e['materialising_path_ms_per_gradient']),
                          'build_ms': sig([v for k, v in e.items() if k.startswith('build_step_ms')][0])})
            if e['M'] == 100 and not e['model'].endswith('/bfgs'):
                kernels.append(kentry('K1 store-free beta-%s N=%d D=%d' % (e['model'].split('/')[0], e['N'], e['D']), e['k1_store_free_kernel_ms'],
                                      e['roofline_hbm']['bytes_per_launch'], 2.0 * e['N'] * e['D'] * e['S'], e['gradients']))
        detail['cpu_loops'] = cpu_loops

    # ---------------- CPU baseline (rank 0, N=1 launch only): the NumPy oracle
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import RefGIGA, RefFrankWolfe, models_ref, coreset_ref
        Ref = RefGIGA if args.alg == 'giga' else RefFrankWolfe
        ll = lambda z, t: models_ref.linreg_loglik(z, t, 1.0)
        try:
            from threadpoolctl import threadpool_info
            thr = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
        except Exception:
            thr = os.cpu_count()
        # (1) parity leg: the first cpu_sample rows through both paths, selections must match exactly
        ns = min(args.cpu_sample, N)
        Zs = Z[:ns].cpu().numpy()
        phi_s = np.empty((ns, S))
        for a in range(0, ns, 100_000):        # rows are independent: chunking does not change the result
            phi_s[a:a + 100_000] = coreset_ref.project(ll, Zs[a:a + 100_000], theta)
        ref = Ref(phi_s.T, phi_s.sum(axis=0))
        ref.build(args.cpu_iters)
        hs = bc.HilbertCoreset(Zs, bc.DeviceProjector(lambda n, w, p: theta, S, model, ctx=ctx), snnls=cls)
        hs.build(args.cpu_iters, args.cpu_iters)
        dsel = hs.snnls._eng.trace()[0]
        rsel = np.array([t[0] for t in ref.trace])
        ridx = np.where(ref.w > 0)[0]
        parity = bool(np.array_equal(dsel, rsel) and np.array_equal(hs.idcs, ridx)
                      and np.allclose(hs.wts, ref.w[ridx], rtol=1e-5))
        del ref, hs, phi_s, Zs
        # (2) the baseline proper: the oracle on ALL N rows of the same data (K1 row-chunked; the greedy loop capped)
        t0 = time.perf_counter()
        phi_ref = np.empty((N, S))
        for a in range(0, N, CHUNK):
            zc = Z_host[a:a + CHUNK] if Z_host is not None else Z[a:a + CHUNK].cpu().numpy()
            for b in range(0, zc.shape[0], 100_000):
                blk = zc[b:b + 100_000]
                phi_ref[a + b:a + b + blk.shape[0]] = coreset_ref.project(ll, blk, theta)
        t_cproj = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref = Ref(phi_ref.T, phi_ref.sum(axis=0))
        t_cinit = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref.build(args.cpu_full_iters)
        t_cit = time.perf_counter() - t0
        # the device run selected the same rows over those first iterations (same Theta, same data, all N rows)
        rsel_full = np.array([t[0] for t in ref.trace])
        # ... and ONE thread: the same loop from a fresh solver state with the BLAS pool limited to one
        # thread, capped at two iterations (NumPy's element-wise passes are single-threaded either way)
        one_thread = None
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=1):
                ref1 = Ref(phi_ref.T, ref.b)
                t0 = time.perf_counter()
                ref1.build(2)
                t_1t = time.perf_counter() - t0
            one_thread = {'value': sig(2.0 / t_1t), 'unit': 'iterations/s', 'cores': 1, 'iters': 2,
                          'same_selections': bool(np.array_equal([t[0] for t in ref1.trace], rsel_full[:2]))}
            del ref1
        except Exception as e:       # no threadpoolctl: say so instead of reporting a pool-sized number as one thread
            one_thread = {'value': None, 'note': 'threadpoolctl unavailable: %s' % type(e).__name__}
        full_match = bool(np.array_equal(rsel_full, f_tr[:len(rsel_full)]))
        out['cpu_baseline'] = {
            'value': args.cpu_full_iters / t_cit, 'unit': 'iterations/s', 'cores': int(thr), 'kind': 'port',
            'sample': 'ALL %d rows, same data and Theta: oracle K1 row-chunked, GIGA init, %d greedy iterations (5 NxS passes each)'
                      % (N, args.cpu_full_iters),
            'projection_s': sig(t_cproj), 'init_s': sig(t_cinit), 'iters_s': sig(t_cit), 'projection_points_dims_per_s': sig(N * D / t_cproj),
            'first_iter_s': sig(t_cproj + t_cinit + t_cit / args.cpu_full_iters),
            'M100_s_extrapolated': sig(t_cproj + t_cinit + 100 * t_cit / args.cpu_full_iters),
            'host_cpus': os.cpu_count(), 'cpu_model': cpu_model(), 'numpy': np.__version__, 'one_thread': one_thread,
            'selections_equal_device_run': 'ok: %d of %d on all rows' % (len(rsel_full), len(rsel_full)) if full_match else 'MISMATCH',
            'parity_on_sample': ('ok: %d selections identical, weights within 1e-5 (first %d rows)' % (len(rsel), ns)) if parity else 'MISMATCH',
            'loops': detail.get('cpu_loops', []),
        }
        if not parity or not full_match:
            out['parity_failure'] = {'device_sample': dsel.tolist(), 'oracle_sample': rsel.tolist(),
                                     'device_full': f_tr[:len(rsel_full)].tolist(), 'oracle_full': rsel_full.tolist()}
        del phi_ref, ref
    sys.stdout.flush()
    os.dup2(_stdout_fd, 1)
    if rank == 0:
        print(json.dumps(out, separators=(',', ':')))
        sys.stdout.flush()
        # everything the compact line leaves out (per-phase splits, every roofline object in full): a file when asked for,
        # otherwise one tagged line on stderr -- never stdout, which carries exactly one line
        dj = json.dumps({'line': out, 'detail': detail})
        if args.detail:
            with open(args.detail, 'w') as f:
                f.write(dj + '\n')
        else:
            sys.stderr.write('BENCH_DETAIL ' + dj + '\n')
    os.dup2(2, 1)                         # teardown chatter goes to stderr again
    parity_failed = bool(out and 'parity_failure' in out)
    if world > 1 or force_xchg:
        dist.barrier()
        comm.close()                      # the library's own RCCL communicator, before torch's
        dist.destroy_process_group()
    if parity_failed:                     # a fast result that differs from the oracle's is not a result: fail the run
        sys.stderr.write('bench.py: device selections / weights differ from the CPU oracle (see parity_failure in the JSON line)\n')
        raise SystemExit(3)


if __name__ == '__main__':
    main()
