#!/usr/bin/env python3
"""Phase time line of the two-level sweep (k_sweep_i4) from s_memtime stamps of every block's thread 0 -- diagnostic build:
  tools/build_variant.sh i4s "bc_prefilter.hip" "-DBC_I4_STAMPS"; BETA_CORES_LIB=tools/libi4s.bin python tools/i4_stamps.py [rows]"""
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

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
D, s = 128, 100
dev = torch.device('cuda', 0)
ctx = bc.default_context()
g0 = torch.Generator(device=dev)
g0.manual_seed(39)
thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
Z = bench.gen_rows(torch, dev, 0, n, D, thstar)
data = bc.DeviceData.from_torch(Z)
del Z
theta = bench.posterior_samples(bc, data, D, s, None)
phi = bc.DeviceProjector(lambda k, w, p: theta, s, bc.likelihoods.LinearRegression(1.0)).project(data)
os.environ['BC_PREFILTER'] = '4'
sv = bc.snnls.GIGA(phi.T, phi.colsum())
lib = N.load()
lib.bc_debug_i4_stamps.argtypes = [C.c_void_p]
buf = np.zeros((1024, 8), dtype=np.uint64)
names = ['entry', 'seeds done (theta0)', 'stream done (wave 0)', 'level 2 done (wave 0)', 'block barrier', 'end', 'digits in LDS', '-']
order = [0, 6, 1, 2, 3, 4, 5]
sv.build(10)
acc = []
for it in range(20):
    sv.build(1)
    torch.cuda.synchronize()
    assert lib.bc_debug_i4_stamps(buf.ctypes.data) == 0
    nb = int((buf[:, 0] > 0).sum())
    st = buf[:nb].astype(np.int64)
    acc.append(st[:, order] - st[:, [0]])
a = np.stack(acc).astype(np.float64)          # [it, block, stamp]
print('blocks %d; ticks since the block entry: mean over blocks and 20 sweeps (min .. max over blocks of the per-block mean)' % a.shape[1])
m = a.mean(axis=0)
for k, o in enumerate(order):
    print('  %-24s %9.0f   (%9.0f .. %9.0f)' % (names[o], m[:, k].mean(), m[:, k].min(), m[:, k].max()))
e0 = np.stack([x for x in acc])[..., 0]
raw = buf[:nb, 0].astype(np.int64)
print('spread of the entry stamps over blocks, last sweep (same counter only within an XCD): %d ticks' % (raw.max() - raw.min()))
end = m[:, order.index(5)]
sd = m[:, order.index(2)] - m[:, order.index(1)]
print('per-block mean duration (ticks), deciles:', np.percentile(end, [0, 10, 25, 50, 75, 90, 100]).astype(int))
print('per-block stream phase (ticks), deciles:', np.percentile(sd, [0, 10, 25, 50, 75, 90, 100]).astype(int))
print('by block index / 8 (XCD = block % 8?): mean duration per residue mod 8:', [int(end[r::8].mean()) for r in range(8)])
print('first 32 blocks:', end[:32].astype(int))
print('last 32 blocks:', end[-32:].astype(int))
