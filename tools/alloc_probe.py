"""Does the two-level pre-filter's larger slab change what the next large allocations cost?  (from_host leg of bench.py)"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, beta_cores_amd as bc
N, D, S = 10_000_000, 128, 100
dev = torch.device('cuda:0')
ctx = bc.default_context()
g0 = torch.Generator(device=dev); g0.manual_seed(39)
thstar = torch.randn((D,), generator=g0, dtype=torch.float64, device=dev)
Z = bench.gen_rows(torch, dev, 0, N, D, thstar)
Z_host = Z.cpu().numpy()
data = bc.DeviceData.from_torch(Z)
theta = bench.posterior_samples(bc, data, D, S, None)
del data, Z
gc.collect(); torch.cuda.empty_cache()
prj = bc.DeviceProjector(lambda k, w, p: theta, S, bc.likelihoods.LinearRegression(1.0))
def sync(): torch.cuda.synchronize()
for form in ('8', '8', '4', '4', '8', '4'):
    os.environ['BC_PREFILTER'] = form
    sync(); t0 = time.perf_counter()
    dd = bc.DeviceData(Z_host)
    sync(); t1 = time.perf_counter()
    del dd; gc.collect()
    sync(); t2 = time.perf_counter()
    alg = bc.HilbertCoreset(Z_host, prj)
    sync(); t3 = time.perf_counter()
    alg.build(3, 10)
    sync(); t4 = time.perf_counter()
    del alg; gc.collect()
    sync(); t5 = time.perf_counter()
    print('form %s: DeviceData %.1f ms   HilbertCoreset(ndarray) %.1f ms   3 steps %.1f ms   destroy %.1f ms' % (form, 1e3*(t1-t0), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4)), flush=True)
