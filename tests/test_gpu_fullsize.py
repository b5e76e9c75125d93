"""BASELINE.json configurations at FULL size on one MI355X, checked through size-independent
properties (the oracle cannot run these sizes in seconds):
  configs[1] linreg   N=1M D=64     configs[2] logistic N=1M D=128     configs[4] neural-linear N=2M D=512
Properties (the reference's own invariant list, tests/test_snnls/test_deterministic.py:37-111, plus
consistency of the device reductions): nnz <= m, size() == nnz, w >= 0, error monotone, error() equals
||sum_i w_i Phi_i - sum_i Phi_i|| recomputed from gathered rows, b == Phi^T 1 via an independent device
matvec, K4 == torch fp64 GEMM, determinism (two builds give identical bits), and a 200k-row prefix of
the same data reproduces the oracle's selections exactly."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from oracle import models_ref as M
from oracle import coreset_ref as C
from oracle import RefGIGA


@pytest.fixture(scope='module')
def env():
    import torch
    import beta_cores_amd as bc
    ctx = bc.default_context()
    return bc, torch, ctx


def gen(torch, n, d, seed, kind):
    g = torch.Generator(device='cuda')
    g.manual_seed(seed)
    if kind == 'relu':                                   # config 5: random ReLU features of a 32-d input
        U = torch.randn((n, 32), generator=g, dtype=torch.float64, device='cuda')
        G = torch.randn((32, d), generator=g, dtype=torch.float64, device='cuda') / np.sqrt(32.)
        X = torch.relu(U @ G)
    else:
        X = torch.randn((n, d), generator=g, dtype=torch.float64, device='cuda')
    thstar = torch.randn((d,), generator=g, dtype=torch.float64, device='cuda') / np.sqrt(d)
    if kind == 'logistic':
        p = torch.sigmoid(X @ (torch.ones(d, dtype=torch.float64, device='cuda') / np.sqrt(d)))
        y = torch.where(torch.rand((n,), generator=g, dtype=torch.float64, device='cuda') <= p, 1., -1.)
        return (y[:, None] * X).contiguous(), thstar
    y = X @ thstar + torch.randn((n,), generator=g, dtype=torch.float64, device='cuda')
    return torch.cat((X, y[:, None]), dim=1).contiguous(), thstar


def check_invariants(bc, phi, solver, steps):
    S = phi.shape[1]
    b = solver.b
    prev = np.inf
    for m in range(1, steps + 1):
        solver.build(1)
        idx, val = solver.sparse_weights()
        assert len(idx) <= m and len(idx) == solver.size() and np.all(val > 0)
        rows = phi.rows(idx - phi.row_offset)
        err = np.sqrt(((val.dot(rows) - b) ** 2).sum())
        assert err <= prev * (1 + 1e-12) + 1e-9
        assert abs(solver.error() - err) <= 1e-9 * max(1., err)
        prev = err
    return idx, val


@pytest.mark.parametrize('cfg', ['linreg_1M_64', 'logistic_1M_128', 'neurlin_2M_512'])
def test_fullsize_properties(env, cfg):
    bc, torch, ctx = env
    n, d, kind = {'linreg_1M_64': (1_000_000, 64, 'linreg'), 'logistic_1M_128': (1_000_000, 128, 'logistic'),
                  'neurlin_2M_512': (2_000_000, 512, 'relu')}[cfg]
    S = 100
    Z, thstar = gen(torch, n, d, {'linreg': 20, 'logistic': 30, 'relu': 50}[kind], kind)
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    rng = np.random.default_rng(7)
    if kind == 'logistic':
        model = bc.likelihoods.LogisticRegression()
        theta = thstar.cpu().numpy() + 0.1 * rng.standard_normal((S, d))
        oracle_ll = M.logistic_loglik
    else:
        model = bc.likelihoods.LinearRegression(1.0)
        # K4 at full size: X^T X, X^T y against torch's fp64 GEMM
        G, v = bc.weighted_gram(data)
        Gt = (Z[:, :d].T @ Z[:, :d]).cpu().numpy()
        vt = (Z[:, :d].T @ Z[:, d]).cpu().numpy()
        assert np.abs(G - Gt).max() <= 1e-11 * np.abs(Gt).max()
        assert np.abs(v - vt).max() <= 1e-10 * np.abs(vt).max()
        mu, L, _ = bc.weighted_post(np.zeros(d), np.eye(d), 1.0, data, None)
        theta = mu + rng.standard_normal((S, d)).dot(L.T)
        oracle_ll = lambda z, t: M.linreg_loglik(z, t, 1.0)
    prj = bc.DeviceBetaProjector(lambda k, w, p: theta, S, model, ctx=ctx)
    phi = prj.project(data)
    assert phi.shape == (n, S)
    # K2: b = Phi^T 1 two independent ways (fused column partials vs per-row matvecs summed on the host)
    b = phi.colsum()
    nrm = phi.norms()
    assert np.all(np.isfinite(b)) and np.all(nrm >= 0) and phi.norm_stats()[0] == 0
    e = np.zeros(S); e[3] = 1.
    col3 = phi.matvec(e)
    assert abs(col3.sum() - b[3]) <= 1e-9 * max(1., np.abs(col3).sum())
    # rows are centred: Phi . 1 == 0 up to rounding
    assert np.abs(phi.matvec(np.ones(S))).max() <= 1e-9 * (1. + np.abs(col3).max())
    # K1 vs the oracle formula on a scattered sample of rows
    pick = rng.choice(n, 300, replace=False)
    Zs = data.rows(pick)
    ref_rows = C.project(oracle_ll, Zs, theta)
    got_rows = phi.rows(pick)
    assert np.abs(got_rows - ref_rows).max() <= 1e-10 * (1. + np.abs(ref_rows).max())
    # greedy loop invariants at full size, GIGA and FrankWolfe
    idx1, val1 = check_invariants(bc, phi, bc.snnls.GIGA(phi.T, b), 25)
    check_invariants(bc, phi, bc.snnls.FrankWolfe(phi.T, b), 10)
    # determinism: a second solver over the same Phi gives identical bits
    s2 = bc.snnls.GIGA(phi.T, b)
    s2.build(25)
    idx2, val2 = s2.sparse_weights()
    assert np.array_equal(idx1, idx2) and np.array_equal(val1, val2)
    # beta-projection at full size stays finite and centred
    phib = prj.project_f(data, 0.1)
    assert np.all(np.isfinite(phib.colsum())) and np.abs(phib.matvec(np.ones(S))).max() <= 1e-9
    # ... and equals the oracle's beta-likelihood (model_lr.py:81-86 / model_neurlinr.py:102-110) on the same scattered rows
    oracle_bl = M.logistic_beta_lik if kind == 'logistic' else (lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0))
    with np.errstate(over='ignore'):
        ref_b = C.project_f(oracle_bl, Zs, theta, 0.1)
    got_b = phib.rows(pick)
    assert np.abs(got_b - ref_b).max() <= 1e-10 * (1. + np.abs(ref_b).max())          # same bound as the log-likelihood rows above
    # a 200k-row prefix through device and oracle: identical selections
    ns = 200_000
    Zp = Z[:ns].cpu().numpy()
    ref = RefGIGA(*(lambda P: (P.T, P.sum(axis=0)))(C.project(oracle_ll, Zp, theta)))
    ref.build(15)
    h = bc.HilbertCoreset(Zp, bc.DeviceProjector(lambda k, w, p: theta, S, model, ctx=ctx))
    h.build(15, 15)
    np.testing.assert_array_equal(h.snnls._eng.trace()[0], [t[0] for t in ref.trace])
    ridx = np.where(ref.w > 0)[0]
    np.testing.assert_array_equal(h.idcs, ridx)
    np.testing.assert_allclose(h.wts, ref.w[ridx], rtol=1e-5)
    del phi, phib, data, Z
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ BASELINE configs[3]: N = 10M, D = 128
def _traces_with_prefilter(bc, phi, b, steps, cls=None):
    """The same greedy loop through the int8 pre-filter + exact rescoring and through the exact fp64 sweep."""
    import os
    cls = cls or bc.snnls.GIGA
    out = {}
    for prec in ('8', '0'):
        old = os.environ.get('BC_PREFILTER')
        os.environ['BC_PREFILTER'] = prec
        try:
            s = cls(phi.T, b)
        finally:
            if old is None:
                os.environ.pop('BC_PREFILTER', None)
            else:
                os.environ['BC_PREFILTER'] = old
        assert s._eng.prefilter == int(prec)
        s.build(steps)
        out[prec] = (s._eng.trace(), s.sparse_weights(), s.error(), s)
    return out


@pytest.mark.parametrize('shard', ['full_10M', 'rank3_of_8'])
def test_config4_headline_size(env, shard):
    """The headline workload itself (bench.py's data recipe): all 10M rows on one GPU, and the 1 250 048-row shard
    rank 3 of 8 holds (row_offset = 3 750 144 != 0, global indices).  int8-pre-filter trace == fp64-sweep trace for
    30 steps, scattered-row K1 vs the oracle, reductions cross-checked, invariants, and a 200k-row prefix through
    device and oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bc, torch, ctx = env
    N, D, S = 10_000_000, 128, 100
    dev = torch.device('cuda', ctx.device)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(39)
    thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
    bounds = bc.shard_bounds(N, 8)
    lo, hi = (0, N) if shard == 'full_10M' else (bounds[3], bounds[4])
    assert shard == 'full_10M' or (lo, hi - lo) == (3_750_144, 1_250_048)
    Z = bench.gen_rows(torch, dev, lo, hi, D, thstar)
    torch.cuda.synchronize()
    n = hi - lo
    data = bc.DeviceData.from_torch(Z, ctx=ctx, row_offset=lo)
    rng = np.random.default_rng(40)
    # Theta from this shard's own posterior (any fixed S x D matrix would do for the checks below)
    mu, L, _ = bc.weighted_post(np.zeros(D), np.eye(D), 1.0, data, None)
    theta = mu + rng.standard_normal((S, D)).dot(L.T)
    model = bc.likelihoods.LinearRegression(1.0)
    prj = bc.DeviceProjector(lambda k, w, p: theta, S, model, ctx=ctx)
    phi = prj.project(data)
    assert phi.shape == (n, S) and phi.row_offset == lo
    b = phi.colsum()
    assert phi.norm_stats()[0] == 0
    e3 = np.zeros(S); e3[3] = 1.
    col3 = phi.matvec(e3)
    assert abs(col3.sum() - b[3]) <= 1e-9 * max(1., np.abs(col3).sum())
    assert np.abs(phi.matvec(np.ones(S))).max() <= 1e-9 * (1. + np.abs(col3).max())
    # K1 vs the oracle on rows scattered over the whole shard (first, last, tile edges, random)
    pick = np.unique(np.concatenate(([0, 1, 127, 128, n - 129, n - 128, n - 1], rng.choice(n, 400, replace=False))))
    ll = lambda z, t: M.linreg_loglik(z, t, 1.0)
    ref_rows = C.project(ll, data.rows(pick), theta)
    got_rows = phi.rows(pick)
    assert np.abs(got_rows - ref_rows).max() <= 1e-10 * (1. + np.abs(ref_rows).max())
    np.testing.assert_allclose(phi.norms()[pick], np.sqrt((ref_rows ** 2).sum(axis=1)), rtol=1e-9)
    # int8 pre-filter + exact rescoring vs the exact fp64 sweep: identical traces, weights and errors
    tr = _traces_with_prefilter(bc, phi, b, 30)
    (f8, st8, e8), (i8, v8), err8, s8 = tr['8']
    (f0, st0, e0), (i0, v0), err0, _ = tr['0']
    assert np.array_equal(f8, f0) and np.array_equal(st8, st0) and np.array_equal(e8, e0)
    assert np.array_equal(i8, i0) and np.array_equal(v8, v0) and err8 == err0
    assert f8.min() >= lo and f8.max() < hi and s8._eng.prefilter_fallbacks() == 0     # global row numbers
    tr_fw = _traces_with_prefilter(bc, phi, b, 12, bc.snnls.FrankWolfe)
    assert np.array_equal(tr_fw['8'][0][0], tr_fw['0'][0][0]) and np.array_equal(tr_fw['8'][1][1], tr_fw['0'][1][1])
    # invariants of the loop at this size (error recomputed from gathered rows, monotone, nnz <= m)
    check_invariants(bc, phi, bc.snnls.GIGA(phi.T, b), 12)
    del tr, tr_fw, s8
    # a 200k-row prefix of the shard through device and oracle: identical selections
    ns = 200_000
    Zp = Z[:ns].cpu().numpy()
    ref = RefGIGA(*(lambda P: (P.T, P.sum(axis=0)))(C.project(ll, Zp, theta)))
    ref.build(15)
    h = bc.HilbertCoreset(Zp, bc.DeviceProjector(lambda k, w, p: theta, S, model, ctx=ctx))
    h.build(15, 15)
    np.testing.assert_array_equal(h.snnls._eng.trace()[0], [t[0] for t in ref.trace])
    ridx = np.where(ref.w > 0)[0]
    np.testing.assert_array_equal(h.idcs, ridx)
    np.testing.assert_allclose(h.wts, ref.w[ridx], rtol=1e-5)
    del phi, data, Z, h
    prj.forget()
    torch.cuda.empty_cache()


def test_config4_all_rows_vs_oracle(env):
    """The headline workload with NOTHING sampled: all 10M rows of bench.py's data recipe through the NumPy oracle on the
    host (its own K1 in row chunks, then the solver) and through the device path; the selections of the first greedy
    iterations are identical and the weights agree -- GIGA (giga.py:20-64), FrankWolfe (frankwolfe.py:15-40) and
    OrthoPursuit (orthopursuit.py:17-42) on the same 8 GB `phi_ref`.  (bench.py's cpu_baseline leg makes the GIGA comparison
    inside the driver's run; here a difference is a red test.)  ~60 s of host work, 8 GB of host memory for the oracle's Phi."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from oracle import RefFrankWolfe, RefOrthoPursuit
    bc, torch, ctx = env
    N, D, S = 10_000_000, 128, 100
    ITERS = {'giga': 6, 'fw': 5, 'omp': 4}
    dev = torch.device('cuda', ctx.device)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(39)
    thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
    Z = bench.gen_rows(torch, dev, 0, N, D, thstar)
    torch.cuda.synchronize()
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    mu, L, _ = bc.weighted_post(np.zeros(D), np.eye(D), 1.0, data, None)
    theta = mu + np.random.default_rng(41).standard_normal((S, D)).dot(L.T)
    model = bc.likelihoods.LinearRegression(1.0)
    dev_out = {}
    for nm, cls in (('giga', bc.snnls.GIGA), ('fw', bc.snnls.FrankWolfe), ('omp', bc.snnls.OrthoPursuit)):
        h = bc.HilbertCoreset(data, bc.DeviceProjector(lambda k, w, p: theta, S, model, ctx=ctx), snnls=cls)
        if nm == 'omp':
            sel = []
            for m in range(ITERS[nm]):            # OrthoPursuit runs the step-wise protocol (NNLS refit on the host)
                h.build(1, m + 1)
                sel.append(h.idcs.copy())
        else:
            h.build(ITERS[nm], ITERS[nm])
            sel = h.snnls._eng.trace()[0][:ITERS[nm]]
        dev_out[nm] = (sel, h.idcs.copy(), h.wts.copy(), h.error())
        del h
        torch.cuda.empty_cache()
    ll = lambda z, t: M.linreg_loglik(z, t, 1.0)
    phi_ref = np.empty((N, S))
    CH = 1_000_000
    for a in range(0, N, CH):
        zc = Z[a:a + CH].cpu().numpy()
        for b in range(0, zc.shape[0], 100_000):       # rows are independent: chunking does not change the result
            blk = zc[b:b + 100_000]
            phi_ref[a + b:a + b + blk.shape[0]] = C.project(ll, blk, theta)
    del Z, data
    torch.cuda.empty_cache()
    bsum = phi_ref.sum(axis=0)
    for nm, cls in (('giga', RefGIGA), ('fw', RefFrankWolfe), ('omp', RefOrthoPursuit)):
        dsel, didx, dwts, derr = dev_out[nm]
        ref = cls(phi_ref.T, bsum)
        if nm == 'omp':
            for m in range(ITERS[nm]):
                ref.build(1)
                np.testing.assert_array_equal(dsel[m], np.where(ref.w > 0)[0], err_msg=nm)
        else:
            ref.build(ITERS[nm])
            np.testing.assert_array_equal(dsel, np.array([t[0] for t in ref.trace]), err_msg=nm)
        ridx = np.where(ref.w > 0)[0]
        np.testing.assert_array_equal(didx, ridx, err_msg=nm)
        np.testing.assert_allclose(dwts, ref.w[ridx], rtol=1e-5, err_msg=nm)
        assert abs(derr - ref.error()) <= 1e-6 * max(1., ref.error()), nm
        del ref


# ------------------------------------------------------------------ the beta-Cores loop itself at configs[1] / configs[2] size
_BCORES_ORACLE = {}
_BCORES_CFGS = ['linreg_1M_64', 'logistic_1M_128', 'logistic_1M_128_laplace', 'svi_linreg_1M_64', 'svi_logistic_1M_128']


def _bcores_case(bc, torch, cfg):
    """Data, beta-likelihood and sampler of one stated-size BetaCoreset case.  The samplers are host closures
    (projector.py:37,66 calls them with the <= M coreset rows) and the SAME closure serves the oracle and the device run."""
    S, beta, opt_itrs = 100, 0.1, 3
    if cfg.startswith('svi_'):            # SparseVI on the plain log-likelihood (sparsevi.py:72-136): same data and samplers
        Z, S, _, opt_itrs, kind, sampler, _ = _bcores_case(bc, torch, 'linreg_1M_64' if 'linreg' in cfg else 'logistic_1M_128')
        ll = (lambda z, t: M.linreg_loglik(z, t, 1.0)) if kind == 'linreg' else M.logistic_loglik
        return Z, S, None, opt_itrs, kind, sampler, ll
    if cfg == 'linreg_1M_64':
        n, d, kind = 1_000_000, 64, 'linreg'
        Z, thstar = gen(torch, n, d, 20, kind)
        E = np.random.default_rng(8).standard_normal((S, d))

        def sampler(sz, wts, pts):
            # the drivers' sampler_w (zellner_neural_linear/main.py:119-124) with fixed normals: Theta follows the CURRENT
            # coreset, so every gradient projects all rows under a different Theta
            if pts.shape[0] == 0:
                wts, pts = np.zeros(1), np.zeros((1, d + 1))
            mu, L, _ = M.linreg_weighted_post(np.zeros(d), np.eye(d), 1.0, pts, wts)
            return mu + E.dot(L.T)
        blik = lambda z, t, b: M.linreg_beta_lik(z, t, b, 1.0)
    else:
        n, d, kind = 1_000_000, 128, 'logistic'
        Z, thstar = gen(torch, n, d, 30, kind)
        E = np.random.default_rng(9).standard_normal((S, d))
        if cfg == 'logistic_1M_128':
            theta = thstar.cpu().numpy() + 0.1 * E
            sampler = lambda sz, wts, pts: theta
        else:
            # the logistic drivers' sampler_w (zellner_logreg/main.py:139-144) with fixed normals and the Newton mode search
            # (converged to the rounding floor: a smooth function of (wts, pts), unlike BFGS's stopping point) -- bench.py's
            # fast config-3 loop
            class FixedNormals:
                def randn(self, a, b):
                    return E
            sampler = bc.samplers.LogisticLaplaceSampler(np.zeros(d), rng=FixedNormals(), solver='newton')
        blik = M.logistic_beta_lik
    return Z, S, beta, opt_itrs, kind, sampler, blik


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('cfg', _BCORES_CFGS)
def test_beta_coreset_steps_at_stated_size_vs_oracle(env, cfg, fused):
    """(`svi_*`: the same with `SparseVICoreset` on the plain log-likelihood, sparsevi.py:72-136.)
    `BetaCoreset.build(2, 2)` with `opt_itrs = 3` at BASELINE configs[1] (N = 1M, D = 64, beta-likelihood of the linear
    regression, model_neurlinr.py:102-110; Theta follows the coreset posterior) and configs[2] (N = 1M, D = 128, logistic
    beta-likelihood, model_lr.py:81-86; once with a fixed Theta, once with the Laplace sampler) against
    `oracle.coreset_ref.RefGreedyVI` (bcores.py:74-150) on ALL rows (its K1 in row chunks on the host, ~20-40 s per case): the
    selected rows are identical and the weights agree within 1e-5 after every build, through the fused store-free gradient
    (bc_vi_gradient) and through the materialising one.  Step sizes of 2e5 / (1 + i) take the weights to the N / M scale
    the drivers reach, so that the second selection sees a residual the first point has changed."""
    from concurrent.futures import ThreadPoolExecutor
    bc, torch, ctx = env
    Z, S, beta, opt_itrs, kind, sampler, blik = _bcores_case(bc, torch, cfg)
    sched = lambda i: 2e5 / (1. + i)
    if cfg not in _BCORES_ORACLE:
        Zh = Z.cpu().numpy()
        CH = 50_000
        pool = ThreadPoolExecutor(8)

        def one(args):
            with np.errstate(over='ignore'):
                return C.project(blik, args[0], args[1]) if beta is None else C.project_f(blik, args[0], args[1], beta)

        def proj(pts, th):                       # rows are independent (projector.py:53-55 centres per row)
            if pts.shape[0] <= CH:
                return one((pts, th))
            return np.concatenate(list(pool.map(one, [(pts[a:a + CH], th) for a in range(0, pts.shape[0], CH)])))
        if hasattr(sampler, '_mode'):
            sampler._mode = None                 # the oracle's run starts the mode search where the device run will
        ref = C.RefGreedyVI(Zh, proj, lambda w, p: sampler(S, w, p), opt_itrs, sched)
        out = []
        for m in range(2):
            ref.build(1, m + 1)
            out.append((ref.idcs.copy(), ref.wts.copy()))
        _BCORES_ORACLE[cfg] = out
        pool.shutdown()
        del ref, Zh
    if hasattr(sampler, '_mode'):
        sampler._mode = None
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    model = bc.likelihoods.LogisticRegression() if kind == 'logistic' else bc.likelihoods.LinearRegression(1.0)
    if beta is None:
        alg = bc.SparseVICoreset(data, bc.DeviceProjector(sampler, S, model, ctx=ctx), opt_itrs=opt_itrs, step_sched=sched,
                                 fused_gradient=fused)
    else:
        alg = bc.BetaCoreset(data, bc.DeviceBetaProjector(sampler, S, model, ctx=ctx), opt_itrs=opt_itrs, step_sched=sched, beta=beta,
                             learn_beta=False, fused_gradient=fused)
    for m in range(2):
        alg.build(1, m + 1)
        ridx, rw = _BCORES_ORACLE[cfg][m]
        np.testing.assert_array_equal(alg.idcs, ridx)
        np.testing.assert_allclose(alg.wts, rw, rtol=1e-5, atol=1e-12)
    np.testing.assert_array_equal(alg.pts, data.rows(alg.idcs))
    del alg, data, Z
    torch.cuda.empty_cache()


def test_config1_stated_size_giga_vs_oracle():
    """BASELINE configs[0] at its stated size: the examples/zellner_gaussian recipe (main.py:33-54: N = 10 000 clean rows,
    d = 8, three outlier clusters of N/50, N/50, N/10 rows, S = 200 samples from the exact posterior = `prj_optimal`,
    main.py:71,106), GIGA via HilbertCoreset, 50 greedy steps -- device (K1 Gaussian log-likelihood + fused greedy loop)
    against the oracle fed with the same RNG stream: selections bit-exact after every step, weights within 1e-5, the
    projection within 1e-11."""
    import importlib.util
    import beta_cores_amd as bc
    path = os.path.join(ROOT, 'examples', 'zellner_gaussian.py')
    spec = importlib.util.spec_from_file_location('zellner_gaussian_example_full', path)
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    N, d, S, steps, tr = 10_000, 8, 200, 50, 3
    res = ex.run('GIGAO', tr, N=N, d=d, M=steps, proj_dim=S, verbose=False)
    # the oracle side: the script's statements in the script's order (same seed -> same data, same Theta)
    np.random.seed(tr)
    mu0, Sig0 = np.zeros(d), np.eye(d)
    Sig = 500 * np.eye(d)
    th = np.zeros(d)
    Sig0inv, Siginv = np.linalg.inv(Sig0), np.linalg.inv(Sig)
    logdetSig = np.linalg.slogdet(Sig)[1]
    X = np.random.multivariate_normal(th, Sig, N)
    mup, LSigp, _ = M.gauss_weighted_post(mu0, Sig0inv, Siginv, X, np.ones(X.shape[0]))
    Xc = np.concatenate((X, np.random.multivariate_normal(th + 200, 0.5 * Sig, int(N / 50.)),
                         np.random.multivariate_normal(th + 150, 0.1 * Sig, int(N / 50.)),
                         np.random.multivariate_normal(th, 10 * Sig, int(N / 10.))))
    assert Xc.shape == (11_400, d) and np.array_equal(res['Xc'], Xc)
    theta = mup + np.random.randn(S, mup.shape[0]).dot(LSigp.T)          # prj_optimal's constructor draw (projector.py:18)
    ll = lambda x, t: M.gauss_loglik(x, t, Siginv, logdetSig)
    ref = C.RefHilbert(Xc, ll, theta)
    assert ref.vecs.shape[0] == Xc.shape[0]                              # no all-zero rows: indices are data rows
    dev_phi = np.asarray(bc.DeviceProjector(lambda n, w, p: theta, S, bc.likelihoods.GaussianLocation(Siginv, logdetSig)).project(Xc))
    assert np.abs(dev_phi - ref.vecs).max() <= 1e-11 * (1. + np.abs(ref.vecs).max())
    for m in range(1, steps + 1):
        ref.build(1, m)
        np.testing.assert_array_equal(res['idcs'][m], ref.idcs)
        np.testing.assert_allclose(res['w'][m], ref.wts, rtol=1e-5)
    assert len(res['idcs'][steps]) > 10
    assert res['rkl'][steps] < res['rkl'][1]                              # the coreset posterior approaches the exact one
