"""The host-array path (the reference's API takes ndarrays: hilbert.py:11, bcores.py:44): uploads go through the
multi-threaded pinned-staging uploader (csrc/bc_upload.hip) and a large live array handed to project() is uploaded and
projected in ONE pipelined call (bc_project_from_host) -- chunk c is projected while chunks c+1.. are on the wire.
Everything must equal the resident path (DeviceData first, then bc_project) BIT FOR BIT: Phi, norms, column sums, and
therefore the greedy traces."""
import os

import numpy as np
import pytest

from oracle import models_ref as M
from oracle import coreset_ref as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def linreg(rng, n, d):
    X = rng.standard_normal((n, d))
    y = X.dot(rng.standard_normal(d)) + rng.standard_normal(n)
    return np.hstack((X, y[:, None]))


@pytest.mark.parametrize('n', [40_000, 70_001, 300_003])
@pytest.mark.parametrize('threads', [None, 0, 3])
def test_upload_round_trip(bc, n, threads):
    """DeviceData(ndarray) through the chunked uploader (>= 32 MB) and the plain copy (small / BC_UPLOAD_THREADS=0)."""
    rng = np.random.default_rng(n)
    Z = rng.standard_normal((n, 129))
    with env(BC_UPLOAD_THREADS=threads):
        data = bc.DeviceData(Z)
    idx = np.unique(np.concatenate((rng.integers(0, n, 500), [0, 1, n - 2, n - 1], np.arange(8190, 8200))))
    assert np.array_equal(data.rows(idx), Z[idx])
    # a second upload re-uses the staging buffers (their "DMA done" events are waited for before they are overwritten)
    with env(BC_UPLOAD_THREADS=threads):
        data2 = bc.DeviceData(Z[::-1].copy())
    assert np.array_equal(data2.rows(idx), Z[::-1][idx])


@pytest.mark.parametrize('n,chunk', [(70_000, 65536), (262_144 + 65_536 + 17, 65536), (600_001, 131072), (300_000, None)])
@pytest.mark.parametrize('model', ['linreg', 'linreg_beta', 'logistic_beta', 'gauss'])
def test_pipelined_projection_is_bit_identical_to_resident(bc, n, chunk, model):
    """n = 70 000: staged kernel (per-tile partials); >= 262 144 rows: the Theta-resident kernel, whose per-wave column
    partials are continued from chunk to chunk (part_init).  `chunk` forces several chunks on these small inputs."""
    rng = np.random.default_rng(1000 + n)
    S, d = 100, 24
    if model in ('linreg', 'linreg_beta'):
        Z = linreg(rng, n, d)
        mdl = bc.likelihoods.LinearRegression(1.0)
    elif model == 'logistic_beta':
        Z = rng.standard_normal((n, d)) * np.where(rng.random(n) < 0.5, 1., -1.)[:, None]
        Z[5] = 0.                                      # a constant row in the first chunk, one in a later chunk
        Z[n - 3] = 0.
        mdl = bc.likelihoods.LogisticRegression()
    else:
        Z = rng.standard_normal((n, d)) * 3.
        Sig = np.eye(d) * 4.
        mdl = bc.likelihoods.GaussianLocation(np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1])
    th = rng.standard_normal((S, d)) * 0.3
    prj = bc.DeviceBetaProjector(lambda k, w, p: th, S, mdl)
    beta = model in ('linreg_beta', 'logistic_beta')
    run = (lambda x: prj.project_f(x, 0.1)) if beta else prj.project
    res = run(bc.DeviceData(Z))                          # resident: upload, then one K1 launch
    with env(BC_PIPE_CHUNK_ROWS=chunk):
        pip = run(Z)                                     # live array: pipelined upload + K1 per chunk
    assert np.array_equal(pip.colsum(), res.colsum())
    assert np.array_equal(pip.norms(), res.norms())
    assert pip.norm_stats() == res.norm_stats()
    rows = np.unique(np.concatenate((rng.integers(0, n, 400), [0, 5, 127, 128, 65535, 65536, n - 3, n - 1])))
    assert np.array_equal(pip.rows(rows), res.rows(rows))
    e = rng.standard_normal(S)
    assert np.array_equal(pip.matvec(e), res.matvec(e))   # every row, through a device reduction
    # and against the oracle on a few rows
    if model == 'linreg':
        ref = C.project(lambda z, t: M.linreg_loglik(z, t, 1.0), Z[rows], th)
        assert np.abs(pip.rows(rows) - ref).max() <= 1e-10 * (1. + np.abs(ref).max())


def test_hilbert_from_host_equals_resident(bc):
    """HilbertCoreset(ndarray, DeviceProjector) -- the reference's call (hilbert.py:11-17) -- through the pipelined path
    selects, weights and errs exactly like the same construction over resident rows."""
    rng = np.random.default_rng(77)
    n, d, S = 400_000, 16, 100
    Z = linreg(rng, n, d)
    th = rng.standard_normal((S, d)) * 0.2
    prj = bc.DeviceProjector(lambda k, w, p: th, S, bc.likelihoods.LinearRegression(1.0))
    with env(BC_PIPE_CHUNK_ROWS=65536):
        a = bc.HilbertCoreset(Z, prj)
    b = bc.HilbertCoreset(bc.DeviceData(Z), prj)
    a.build(20, 20)
    b.build(20, 20)
    assert np.array_equal(a.snnls._eng.trace()[0], b.snnls._eng.trace()[0])
    wa, pa, ia = a.get()
    wb, pb, ib = b.get()
    assert np.array_equal(ia, ib) and np.array_equal(wa, wb) and np.array_equal(pa, Z[ia]) and a.error() == b.error()


def test_from_host_argument_errors(bc):
    rng = np.random.default_rng(5)
    Z = linreg(rng, 70_000, 8)
    th = rng.standard_normal((20, 9))                     # wrong sample width
    prj = bc.DeviceProjector(lambda k, w, p: th, 20, bc.likelihoods.LinearRegression(1.0))
    with pytest.raises(ValueError):
        prj.project(Z)


def test_pipelined_k1_launches_count_in_the_k1_timer(bc):
    """K1 launches made by bc_project_from_host are bracketed like bc_project's (timer class 1): one span per chunk."""
    rng = np.random.default_rng(6)
    n, d, S = 300_000, 16, 64
    Z = linreg(rng, n, d)
    th = rng.standard_normal((S, d)) * 0.2
    prj = bc.DeviceProjector(lambda k, w, p: th, S, bc.likelihoods.LinearRegression(1.0))
    ctx = bc.default_context()
    with env(BC_PIPE_CHUNK_ROWS=65536):
        prj.project(Z)                                    # (first use of this kernel instantiation: code-object load, not timed)
    ctx.enable_timing(1)
    try:
        ctx.kernel_time_reset()
        with env(BC_PIPE_CHUNK_ROWS=65536):
            prj.project(Z)
        ms, launches = ctx.kernel_time(1)
        assert launches == -(-n // 65536) and ms > 0.
        ctx.kernel_time_reset()
        prj.project(bc.DeviceData(Z))
        ms1, launches1 = ctx.kernel_time(1)
        assert launches1 == 1 and 0.2 * ms1 < ms < 20. * ms1      # five chunk launches ~ one launch over all rows
    finally:
        ctx.enable_timing(0)
        ctx.kernel_time_reset()
