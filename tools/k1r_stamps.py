"""In-kernel clock and phase split of the Theta-resident K1 (k_project_r): needs a library built with -DBC_K1_STAMPS
(s_memtime stamps per wave and phase; BETA_CORES_LIB points at it).  Prints kernel time, ticks per wave -> the clock the
chip holds under this load, and cycles per 32-row group for contraction / row statistics / stores + column sums.
    hipcc ... -DBC_K1_STAMPS -c bc_project.hip ; link ; BETA_CORES_LIB=... python tools/k1r_stamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import beta_cores_amd as bc
dev = torch.device('cuda', 0)
ctx = bc.Context(0); bc.set_default_context(ctx)
S = 100
N, D = int(os.environ.get('N', 10_000_000)), int(os.environ.get('D', 128))
Z = torch.randn((N, D + 1), dtype=torch.float64, device=dev)
theta = np.random.randn(S, D) * 0.1
data = bc.DeviceData.from_torch(Z, ctx=ctx)
Zlog = Z[:, :D].contiguous()
data_log = bc.DeviceData.from_torch(Zlog, ctx=ctx)
st = torch.zeros((2048, 8), dtype=torch.int64, device=dev)
for name, prj, beta in [('linreg', bc.DeviceBetaProjector(lambda n, w, p: theta, S, bc.likelihoods.LinearRegression(1.0), ctx=ctx), None),
                        ('linreg_beta', bc.DeviceBetaProjector(lambda n, w, p: theta, S, bc.likelihoods.LinearRegression(1.0), ctx=ctx), 0.1),
                        ('logistic', bc.DeviceBetaProjector(lambda n, w, p: theta, S, bc.likelihoods.LogisticRegression(), ctx=ctx), None)]:
    dd = data_log if name == 'logistic' else data
    f = (lambda: prj.project(dd)) if beta is None else (lambda: prj.project_f(dd, beta))
    os.environ.pop('BC_K1_STAMP_PTR', None)
    for _ in range(3): phi = f(); phi = None
    ctx.sync()
    os.environ['BC_K1_STAMP_PTR'] = str(st.data_ptr())
    ctx.enable_timing(1); ctx.kernel_time_reset()
    phi = f(); ctx.sync(); torch.cuda.synchronize()
    ms, n = ctx.kernel_time(1)
    h = st.cpu().numpy().astype(np.float64)
    groups = (N + 31) // 32 / 2048.
    print('%s: kernel %.3f ms; per wave total %.0f cycles (min %.0f max %.0f); per group: contraction %.0f  row_stats %.0f  stores+colsum %.0f  (sum %.0f; MFMA floor 25600)' % (
        name, ms / n, h[:, 3].mean(), h[:, 3].min(), h[:, 3].max(), h[:, 0].mean() / groups, h[:, 1].mean() / groups, h[:, 2].mean() / groups, h[:, :3].sum(axis=1).mean() / groups))
    print('   start spread %.0f cycles; clock held = ticks per wave / kernel time = %.3f GHz' % (h[:, 4].max() - h[:, 4].min(), h[:, 3].mean() / (ms / n * 1e-3) / 1e9))
    phi = None
