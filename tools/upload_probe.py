"""Where does the time of DeviceData(ndarray) go?  (one-off probe for csrc/bc_upload.hip; run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import beta_cores_amd as bc

ctx = bc.default_context()
n, dz = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 129
gb = n * dz * 8 / 1e9


def t(f):
    torch.cuda.synchronize(); ctx.sync()
    t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ctx.sync()
    return time.perf_counter() - t0, r


for rep in range(2):
    dt, x = t(lambda: torch.empty((n, dz), dtype=torch.float64, device='cuda'))
    print('torch.empty %.1f GB on device: %.1f ms' % (gb, 1e3 * dt))
    x.normal_()
    dt, h = t(lambda: x.cpu())
    print('x.cpu(): %.1f ms = %.1f GB/s' % (1e3 * dt, gb / dt))
    Zt = h.numpy()
    del x
    torch.cuda.empty_cache()
Zn = np.empty((n, dz)); Zn[:] = 1.5
for name, Z in (('numpy-filled', Zn), ('torch .cpu().numpy()', Zt)):
    for thr in ('8', '0', '8', '2', '16'):
        os.environ['BC_UPLOAD_THREADS'] = thr
        dt, dd = t(lambda: bc.DeviceData(Z))
        dt2, _ = t(lambda: dd.__del__() if False else None)
        t0 = time.perf_counter(); del dd; torch.cuda.synchronize(); tf = time.perf_counter() - t0
        print('%-22s threads=%-2s DeviceData(): %.1f ms = %.1f GB/s   (free: %.1f ms)' % (name, thr, 1e3 * dt, gb / dt, 1e3 * tf))
# the allocation alone, through the library
dt, s = t(lambda: bc.DeviceData.slot(dz, cap_rows=n))
print('bc_data_create (hipMalloc %.1f GB): %.1f ms' % (gb, 1e3 * dt))
os.environ['BC_UPLOAD_THREADS'] = '8'
dt, _ = t(lambda: s.update(Zn))
print('upload into the existing buffer, 8 threads: %.1f ms = %.1f GB/s' % (1e3 * dt, gb / dt))
os.environ['BC_UPLOAD_THREADS'] = '0'
dt, _ = t(lambda: s.update(Zn))
print('upload into the existing buffer, plain: %.1f ms = %.1f GB/s' % (1e3 * dt, gb / dt))
dt, _ = t(lambda: s.update(Zt))
print('upload (torch-born array) into the existing buffer, plain: %.1f ms = %.1f GB/s' % (1e3 * dt, gb / dt))
