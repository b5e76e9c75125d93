"""Fuzz (not collected by pytest): the pipelined host-array projection (bc_project_from_host) against the resident path
(DeviceData, then bc_project) on random shapes, models, chunk sizes and upload modes -- Phi rows, norms, column sums and
norm statistics must be bit-identical.  Usage: python tests/fuzz/fuzz_fromhost.py SEED SECONDS (needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import beta_cores_amd as bc
bc.default_context()
rng = np.random.RandomState(int(sys.argv[1]))
t_end = time.time() + float(sys.argv[2])
cases = bad = 0
while time.time() < t_end:
    n = int(rng.choice([rng.randint(65536, 140000), rng.randint(262144, 700000), 65536 * rng.randint(1, 9), 128 * rng.randint(512, 4000) + rng.randint(0, 128)]))
    d = int(rng.choice([1, 3, 8, 24, 31, 32, 33, 64]))
    S = int(rng.choice([5, 16, 37, 64, 100, 101, 112, 200, 300]))
    kind = rng.choice(['linreg', 'linreg_beta', 'logistic', 'logistic_beta', 'gauss', 'gauss_beta'])
    if kind.startswith('linreg'):
        Z = rng.randn(n, d + 1)
        mdl = bc.likelihoods.LinearRegression(float(rng.uniform(0.5, 2.)))
    elif kind.startswith('logistic'):
        Z = rng.randn(n, d) * rng.choice([1., 30.], size=(n, 1), p=[0.97, 0.03])
        Z[rng.randint(0, n, 3)] = 0.
        mdl = bc.likelihoods.LogisticRegression()
    else:
        Z = rng.randn(n, d) * 2.
        Sig = np.eye(d) * 3.
        mdl = bc.likelihoods.GaussianLocation(np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1])
    th = rng.randn(S, d) * 0.4
    prj = bc.DeviceBetaProjector(lambda k, w, p: th, S, mdl)
    beta = float(rng.choice([0.1, 0.5])) if kind.endswith('beta') else None
    run = (lambda x: prj.project_f(x, beta)) if beta is not None else prj.project
    os.environ['BC_PIPE_CHUNK_ROWS'] = str(int(rng.choice([65536, 131072, 1 << 30])))
    os.environ['BC_UPLOAD_THREADS'] = str(int(rng.choice([0, 0, 3])))
    if rng.rand() < 0.3:
        os.environ['BC_K1_STAGED'] = '1'
    else:
        os.environ.pop('BC_K1_STAGED', None)
    res = run(bc.DeviceData(Z))
    pip = run(Z)
    rows = np.unique(np.concatenate((rng.randint(0, n, 200), [0, n - 1, 65535 % n, 65536 % n])))
    ok = (np.array_equal(pip.colsum(), res.colsum()) and np.array_equal(pip.norms(), res.norms())
          and pip.norm_stats() == res.norm_stats() and np.array_equal(pip.rows(rows), res.rows(rows), equal_nan=True))
    cases += 1
    if not ok:
        bad += 1
        print('MISMATCH', dict(n=n, d=d, S=S, kind=kind, beta=beta, chunk=os.environ['BC_PIPE_CHUNK_ROWS'], thr=os.environ['BC_UPLOAD_THREADS'],
                               staged=os.environ.get('BC_K1_STAGED')), flush=True)
    del res, pip, prj, Z
print('cases %d mismatches %d' % (cases, bad))
