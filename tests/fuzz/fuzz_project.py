"""Fuzz (not collected by pytest): random shapes and models through K1 -- staged and Theta-resident kernels (the latter from
262 144 rows), materialised and store-free -- against the NumPy oracle's projections (tolerance 1e-11 * (1 + max|f|), the
parity bar of golden F2) and against each other (store-free column sums == materialised ones, bit for bit; resident ==
staged within 1e-13).  Usage: python tests/fuzz/fuzz_project.py SEED SECONDS (needs a GPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import beta_cores_amd as bc
from oracle import models_ref as M

bc.default_context()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.)
cases = bad = 0
while time.time() < t_end:
    big = rng.rand() < 0.25
    n = int(rng.randint(262_144, 420_000)) if big else int(10 ** rng.uniform(0, 4.3))
    d = int(rng.choice([1, 3, 8, 17, 32, 63, 64, 65, 100, 128, 150]))
    s = int(rng.choice([1, 16, 40, 64, 97, 100, 104, 112, 128, 200, 256]))
    if big:
        d = int(rng.choice([8, 32, 64, 128])); s = int(rng.choice([64, 100, 112]))
    kind = rng.randint(3)
    th = rng.randn(s, d) * rng.choice([0.05, 0.3, 1.0])
    if kind == 0:
        sig = float(rng.uniform(0.3, 3.)); model = bc.likelihoods.LinearRegression(sig); Z = rng.randn(n, d + 1)
        ll = lambda z, t: M.linreg_loglik(z, t, sig); bl = lambda z, t, b: M.linreg_beta_lik(z, t, b, sig)
    elif kind == 1:
        model = bc.likelihoods.LogisticRegression(); Z = rng.randn(n, d) * rng.choice([0.3, 1., 6.])
        ll = M.logistic_loglik; bl = M.logistic_beta_lik
    else:
        Sig = np.diag(rng.uniform(0.5, 2.0, d)); Si = np.linalg.inv(Sig); ld = np.linalg.slogdet(Sig)[1]
        model = bc.likelihoods.GaussianLocation(Si, ld); Z = rng.randn(n, d)
        ll = lambda x, t: M.gauss_loglik(x, t, Si, ld); bl = lambda x, t, b: M.gauss_beta_lik(x, t, b, Si, ld)
    beta = None if rng.rand() < 0.4 else float(rng.choice([0.1, 0.3, 0.7]))
    if beta is not None and kind == 1 and rng.rand() < 0.5:      # the logistic beta-likelihood's power-table body over its whole range
        beta = float(rng.choice([0.01, 0.05, 1.0, 2.0, 8.0, 30.0]))
    prj = bc.DeviceBetaProjector(lambda k, w, p: th, s, model)
    dd = bc.DeviceData(Z)
    v = ll(Z, th) if beta is None else bl(Z, th, beta)       # un-centred model values: their maximum sets the tolerance
    fmax = np.abs(v[np.isfinite(v)]).max() if np.isfinite(v).any() else 0.
    ref = v - v.mean(axis=1)[:, np.newaxis]                 # projector.py:23-26 / 51-55
    tol = 1e-11 * (1. + fmax)
    res = {}
    for staged in ('0', '1'):
        os.environ['BC_K1_STAGED'] = staged
        phi = prj.project(dd) if beta is None else prj.project_f(dd, beta)
        got = phi.to_host()
        cs = phi.colsum()
        sf = prj.colsum(dd, beta=beta)
        res[staged] = got
        ok = np.abs(got - ref).max() <= tol
        ok2 = sf is None or np.array_equal(sf, cs)
        if not (ok and ok2):
            bad += 1
            print('MISMATCH n=%d d=%d s=%d kind=%d beta=%s staged=%s: |dev-ref| %.3e (tol %.1e), store-free==materialised %s'
                  % (n, d, s, kind, beta, staged, np.abs(got - ref).max(), tol, ok2))
        del phi
    if np.abs(res['0'] - res['1']).max() > 1e-13 * (1. + np.abs(res['1']).max()):
        bad += 1
        print('MISMATCH resident vs staged n=%d d=%d s=%d kind=%d beta=%s: %.3e' % (n, d, s, kind, beta, np.abs(res['0'] - res['1']).max()))
    os.environ.pop('BC_K1_STAGED', None)
    prj.forget() if hasattr(prj, 'forget') else None
    cases += 1
print('fuzz project: %d problems, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
