"""CPU study (NumPy): can a row's bounds of an earlier greedy step be reused?  (No: the residual direction turns too fast.)"""
# feasibility of bound reuse across steps: how many rows can still reach the maximum when their (s0, s1) of `age` steps ago
# are known to +-(delta + ||v_now - v_then||)?
import numpy as np, sys
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
D, S, M = 128, 100, int(sys.argv[2]) if len(sys.argv) > 2 else 60
delta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0065
rng = np.random.default_rng(0)
X = rng.standard_normal((N, D)); th = rng.standard_normal(D)
y = X @ th + rng.standard_normal(N)
out = rng.random(N) < 0.1
y[out] = 10 + 0.5 * rng.standard_normal(out.sum())
A = X.T @ X + np.eye(D); mu = np.linalg.solve(A, X.T @ y); L = np.linalg.cholesky(np.linalg.inv(A))
Th = mu + rng.standard_normal((S, D)) @ L.T
Phi = -0.5 * (y[:, None] - X @ Th.T) ** 2
Phi -= Phi.mean(axis=1)[:, None]
del X
u = (Phi / np.linalg.norm(Phi, axis=1)[:, None]).astype(np.float32)
b = Phi.sum(axis=0); bn = b / np.linalg.norm(b)
del Phi
def upper(s0, s1, d0, d1):
    a = np.abs(s1) + d1; c = 1 - a * a
    ok = c > 1e-3
    cc = np.where(ok, c, 1)
    f = s0 / np.sqrt(np.maximum(1 - s1 * s1, 1e-30))
    e = d0 / np.sqrt(cc) + (np.abs(s0) + d0) * a * d1 / cc ** 1.5
    return np.where(ok, f + e, np.inf)
xw = np.zeros(S); hist = []
for it in range(M):
    nw = np.linalg.norm(xw); yv = xw / nw if nw > 0 else xw
    cd = bn - (bn @ yv) * yv; cd /= np.linalg.norm(cd)
    s0 = (u @ cd.astype(np.float32)).astype(np.float64); s1 = (u @ yv.astype(np.float32)).astype(np.float64)
    with np.errstate(all='ignore'):
        f = np.where(1 - s1 * s1 > 1e-12, s0 / np.sqrt(np.abs(1 - s1 * s1)), -np.inf)
    fmax = f.max(); fi = int(np.argmax(f))
    line = 'it %3d fmax %.4f |' % (it, fmax)
    for age, (cd0, yv0, s00, s10) in enumerate(reversed(hist[-8:]), 1):
        d0 = np.linalg.norm(cd - cd0); d1 = np.linalg.norm(yv - yv0)
        U = upper(s00, s10, delta + d0, delta + d1)
        line += ' a%d: d0 %.3f d1 %.3f cand %7d |' % (age, d0, d1, int((U >= fmax).sum()))
    print(line, flush=True)
    hist.append((cd.copy(), yv.copy(), s0, s1)); hist = hist[-8:]
    xf = u[fi].astype(np.float64)
    if nw == 0:
        xw_new = xf
    else:
        bxf = bn @ xf; bxw = bn @ yv; xwxf = yv @ xf
        gA = bxw - bxf * xwxf; gB = bxf - bxw * xwxf
        if gA < 0 or gB < 0: print('stop'); break
        xw_new = gA / (gA + gB) * yv + gB / (gA + gB) * xf
    xw = xw_new * (np.linalg.norm(b) * (xw_new @ bn) / (xw_new @ xw_new))
