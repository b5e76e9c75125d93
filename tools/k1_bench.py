#!/usr/bin/env python3
"""K1 (projection kernel) timings per model and shape, HIP-event timed through the library's own kernel timer.

    python tools/k1_bench.py [--models linreg,linreg_beta,logistic,logistic_beta,gauss,gauss_beta] [--rows 1000000]
                             [--dim 128] [--samples 100] [--reps 5] [--store-free]

Also the workload of the PMC passes under profiles/ (rocprofv3 --pmc ... -- python3 tools/k1_bench.py ...)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--models', default='linreg,linreg_beta,logistic,logistic_beta')
    ap.add_argument('--rows', type=int, default=1_000_000)
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--samples', type=int, default=100)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--store-free', action='store_true')
    a = ap.parse_args()
    import torch
    import beta_cores_amd as bc
    ctx = bc.default_context()
    n, d, S = a.rows, a.dim, a.samples
    g = torch.Generator(device='cuda')
    g.manual_seed(5)
    rng = np.random.default_rng(1)
    th = rng.standard_normal((S, d)) * (1.0 / np.sqrt(d))
    Zl = torch.randn((n, d + 1), generator=g, dtype=torch.float64, device='cuda')
    Zg = Zl[:, :d].contiguous()
    dl, dg = bc.DeviceData.from_torch(Zl), bc.DeviceData.from_torch(Zg)
    Sig = np.eye(d) * 2.0
    models = {
        'linreg': (bc.likelihoods.LinearRegression(1.0), dl, None),
        'linreg_beta': (bc.likelihoods.LinearRegression(1.0), dl, 0.1),
        'logistic': (bc.likelihoods.LogisticRegression(), dg, None),
        'logistic_beta': (bc.likelihoods.LogisticRegression(), dg, 0.1),
        'gauss': (bc.likelihoods.GaussianLocation(np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]), dg, None),
        'gauss_beta': (bc.likelihoods.GaussianLocation(np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]), dg, 0.1),
    }
    for name in a.models.split(','):
        model, data, beta = models[name]
        prj = bc.DeviceBetaProjector(lambda k, w, p: th, S, model, ctx=ctx)
        if a.store_free:
            f = lambda: prj.colsum(data, beta=beta)
        else:
            f = (lambda: prj.project(data)) if beta is None else (lambda: prj.project_f(data, beta))
        r = f()
        r = None
        r = f()
        r = None
        ctx.enable_timing(1)
        ctx.kernel_time_reset()
        for _ in range(a.reps):
            r = f()
            r = None
        ms, cnt = ctx.kernel_time(1)
        ms /= max(cnt, 1)
        dz = data.shape[1]
        byt = 8.0 * n * dz + (0 if a.store_free else 8.0 * n * S)
        fl = 2.0 * n * d * S
        print('%-14s N=%d D=%d S=%d %s: %.4f ms  hbm %.3f  fp64-mfma %.3f' % (
            name, n, d, S, 'store-free' if a.store_free else 'materialised', ms, byt / (ms * 1e-3) / 8e12,
            fl / (ms * 1e-3) / 78.6e12))
        ctx.enable_timing(0)


if __name__ == '__main__':
    main()
