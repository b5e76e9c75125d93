#!/usr/bin/env python3
"""How selective does the int8 pre-filter stay as the coreset grows?  Builds the headline workload's coreset to M = 1000 in
blocks of 100 iterations and prints, per block: us per iteration, rows rescored per sweep, fp64 fallbacks.

    python tools/large_m_probe.py [rows] [blocks]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import beta_cores_amd as bc
import bench


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    D, s = 128, 100
    dev = torch.device('cuda', 0)
    ctx = bc.Context(0)
    bc.set_default_context(ctx)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(39)
    thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
    Z = bench.gen_rows(torch, dev, 0, n, D, thstar)
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    theta = bench.posterior_samples(bc, data, D, s, None)
    alg = bc.HilbertCoreset(data, bc.DeviceProjector(lambda k, w, p: theta, s, bc.likelihoods.LinearRegression(1.0), ctx=ctx))
    prev = (0, 0, 0)
    for b in range(blocks):
        ctx.sync()
        t0 = time.perf_counter()
        alg.snnls.build(100)
        ctx.sync()
        dt = time.perf_counter() - t0
        st = alg.snnls._eng.prefilter_stats()
        d = tuple(x - y for x, y in zip(st, prev))
        prev = st
        print('iterations %4d-%4d: %7.1f us/iteration  list length %4d  sweeps %d  rows rescored per sweep %.1f  fp64 fallbacks %d  error %.4e'
              % (100 * b + 1, 100 * b + 100, 1e6 * dt / 100, alg.snnls.size(), d[0], d[1] / max(d[0], 1), d[2], alg.snnls.error()), flush=True)


if __name__ == '__main__':
    main()
