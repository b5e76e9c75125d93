"""How selective are the two levels of the two-level pre-filter (csrc/bc_prefilter_i4.h), step by step, and what does a
step cost?  bench.py's workload at N rows; prints rows listed by the 4-bit sweep, rows refined from the int8 records,
rows rescored exactly and fallbacks per step, then the timed rate of the two-level and the one-level form.
    python3 tools/two_level_probe.py [N] [M]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import beta_cores_amd as bc

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 40
D, S = 128, 100
dev = torch.device('cuda:0')
ctx = bc.default_context()
g0 = torch.Generator(device=dev)
g0.manual_seed(39)
thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
if os.environ.get('PROBE_DATA') == 'randn':        # tests/test_gpu_prefilter.py::test_million_rows_identical
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    D = 32
    Z = torch.randn((N, D + 1), generator=g, dtype=torch.float64, device='cuda')
    data = bc.DeviceData.from_torch(Z)
    del Z
    theta = np.random.default_rng(1).standard_normal((S, D)) * 0.3
else:
    Z = bench.gen_rows(torch, dev, 0, N, D, thstar)
    data = bc.DeviceData.from_torch(Z)
    del Z
    theta = bench.posterior_samples(bc, data, D, S, None)
phi = bc.DeviceProjector(lambda k, w, p: theta, S, bc.likelihoods.LinearRegression(1.0)).project(data)


def make(form):
    os.environ['BC_PREFILTER'] = form
    try:
        return bc.snnls.GIGA(phi.T, phi.colsum())
    finally:
        os.environ.pop('BC_PREFILTER', None)


for form in ('8', '4', '8', '4'):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sv = make(form)
    torch.cuda.synchronize()
    print('solver create, BC_PREFILTER=%s: %.2f ms' % (form, 1e3 * (time.perf_counter() - t0)), flush=True)
    del sv
sv = make('4')
print('form', sv._eng.prefilter_form, 'rows', N, flush=True)
prev = (0, 0, 0)
pst = (0, 0, 0)
for it in range(M):
    sv.build(1)
    lv = sv._eng.prefilter_levels()
    st = sv._eng.prefilter_stats()
    print('step %3d  l1 sweeps +%d  listed %8d (%.3f %%)  refined %7d  rescored %3d  fallbacks %d' % (
        it, lv[0] - prev[0], lv[1] - prev[1], 100. * (lv[1] - prev[1]) / N, lv[2] - prev[2], st[1] - pst[1], st[2]), flush=True)
    prev, pst = lv, st
ref = make('8')
ref.build(M)
a, b = sv._eng.trace(), ref._eng.trace()
print('traces equal:', np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]))
for form in ('4', '8', '4', '8'):
    s2 = make(form)
    s2.build(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s2.build(100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('form %s: %.1f us per step, %.0f it/s  (fallbacks %d)' % (form, 1e4 * dt, 100 / dt, s2._eng.prefilter_stats()[2]), flush=True)
