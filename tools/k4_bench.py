import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import beta_cores_amd as bc
dev = torch.device('cuda', 0)
ctx = bc.Context(0)
bc.set_default_context(ctx)
ctx.enable_timing(True)
for (N, D) in [(2_000_000, 512), (10_000_000, 128), (4_000_000, 64)]:
    Z = torch.randn((N, D + 1), dtype=torch.float64, device=dev)
    data = bc.DeviceData.from_torch(Z, ctx=ctx)
    w = None
    bc.weighted_gram(data, w); ctx.kernel_time_reset()
    t0 = time.perf_counter()
    for _ in range(3): bc.weighted_gram(data, w)
    wall = (time.perf_counter() - t0) / 3
    ms, n = ctx.kernel_time(2); ms /= n
    bt = 128 if D > 64 else 64; nt = -(-D // bt); ntri = nt * (nt + 1) // 2
    ns = bt // 16                                     # diagonal tiles: sub-tiles on or above the diagonal only
    fl = 2. * N * bt * bt * ((ntri - nt) + nt * (ns * (ns + 1) // 2) / (ns * ns))
    print('gram N=%d D=%d: gram+reduce %.3f ms (wall %.1f ms)  executed %.1f TF = %.3f of 78.6  %.0f GB/s' % (N, D, ms, wall * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 78.6, 8. * N * (D + 1) / ms / 1e6))
    del data, Z
    torch.cuda.empty_cache()
