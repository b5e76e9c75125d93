#!/usr/bin/env python3
"""Phase time line of the single-block step kernel (k_step_finish_pf<GIGA, RS = true>: rescoring + greedy step in one launch)
from `s_memtime` stamps of thread 0 -- a diagnostic build (-DBC_FIN_STAMPS, tools/build_variant.sh fin "bc_snnls.hip"
"-DBC_FIN_STAMPS"), never the shipped library.

  BETA_CORES_LIB=tools/libfin.bin python tools/fin_stamps.py [rows] [nnz,nnz,...]

Prints, per list length, the mean tick of every stamp over 60 steps (ticks = shader cycles of the CU the block ran on)
and the kernel's event-timed duration, so that ticks convert to time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import beta_cores_amd as bc
from beta_cores_amd import _native as N
import bench

ORDER = [0, 25, 1, 10, 23, 24, 11, 12, 14, 13, 2, 3, 4, 5, 6, 7, 15, 8, 9]
NAMES = {0: 'kernel start', 25: 'up-front loads requested', 23: 'rescoring: block lists consumed', 24: 'rescoring: barrier behind that', 1: 'up-front loads + barrier', 10: 'rescoring: Lmax', 11: 'rescoring: block/tile scan (B1+B2 merged)',
         12: 'rescoring: candidate list complete', 14: 'rescoring: candidate row + v in LDS', 13: 'rescoring: exact scores (chains)',
         2: 'rescoring: record written (winner, column)', 3: 'pick', 4: 'step sizes (5 wave sums, divisions)', 5: 'apply (list scale / append)',
         6: 'xw = A.w and error from the list', 7: 'guard + retry state', 15: 'prep: next sweep vectors', 8: 'prep: int8 digits',
         9: 'write back'}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nnzs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10, 100, 400]
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
    prj = bc.DeviceProjector(lambda k, w, p: theta, s, bc.likelihoods.LinearRegression(1.0), ctx=ctx)
    alg = bc.HilbertCoreset(data, prj)
    lib = N.load()
    if not hasattr(lib, 'bc_debug_fin_stamps'):
        raise SystemExit('this library has no stamps: build with tools/build_variant.sh fin "bc_snnls.hip" "-DBC_FIN_STAMPS" and set BETA_CORES_LIB')
    done = 0
    for nnz in nnzs:
        alg.build(nnz - done, 2000)
        done = nnz
        acc = np.zeros(64)
        cnt = 0
        ctx.timing_classes(0x21)
        ctx.enable_timing(1)
        ctx.kernel_time_reset()
        for it in range(60):
            alg.snnls.build(1)
            buf = (C.c_ulonglong * 64)()
            lib.bc_debug_fin_stamps(buf)
            t = np.array(buf[:], dtype=np.float64)
            acc += t - t[0]
            cnt += 1
        done += 60
        ms_fin, nl = ctx.kernel_time(5)
        ms_sw, nsw = ctx.kernel_time(0)
        ctx.enable_timing(0)
        t = acc / cnt
        print('--- list length %d..%d, %d rows: finish launch %.2f us (events), sweep %.2f us; %d stamps span %.0f ticks'
              % (nnz, nnz + 60, n, 1e3 * ms_fin / max(nl, 1), 1e3 * ms_sw / max(nsw, 1), len(ORDER), t[9]))
        prev = 0.
        for i in ORDER:
            if i != 0 and t[i] == 0.:
                continue
            print('  %-52s at %8.0f ticks  (+%6.0f)' % (NAMES[i], t[i], t[i] - prev))
            prev = t[i]
        print('  prefilter stats (sweeps, candidates, fallbacks):', alg.snnls._eng.prefilter_stats())
        print('  last step: blocks walked tile by tile %d, candidates straight from the block lists %d, lists in use %d' % (buf[20], buf[21], buf[22]))


if __name__ == '__main__':
    main()
