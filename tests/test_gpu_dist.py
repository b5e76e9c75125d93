"""The sharded path on real kernels: two ranks share GPU 0, each owns half of the rows
(K1 / K2 / K3 on its shard), candidate records travel over gloo, the finish kernel runs
replicated.  Results must equal the single-rank device run and the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_dist_cpu import launch                    # noqa: E402
from dist_worker import linreg_problem              # noqa: E402
from oracle import models_ref as M                  # noqa: E402
from oracle import coreset_ref as C                 # noqa: E402
from oracle import RefGIGA, RefFrankWolfe           # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['gpu_hilbert', 'gpu_fw'])
def test_two_ranks_one_gpu_hilbert(tmp_path, mode):
    import beta_cores_amd as bc
    Z, th = linreg_problem()
    ll = lambda z, t: M.linreg_loglik(z, t, 1.0)
    ref = C.RefHilbert(Z, ll, th, RefFrankWolfe if mode == 'gpu_fw' else RefGIGA)
    ref.build(25, 25)
    single = bc.HilbertCoreset(Z, bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0)),
                               snnls=bc.snnls.FrankWolfe if mode == 'gpu_fw' else bc.snnls.GIGA)
    single.build(25, 25)
    r0, r1 = launch(mode, tmp_path)
    for r in (r0, r1):
        np.testing.assert_array_equal(r['idx'], ref.idcs)
        np.testing.assert_array_equal(r['idx'], single.idcs)
        np.testing.assert_allclose(r['val'], ref.wts, rtol=1e-5)
        np.testing.assert_allclose(r['val'], single.wts, rtol=1e-9)     # only b's summation order differs
        np.testing.assert_array_equal(r['trace_f'], single.snnls._eng.trace()[0])
    assert np.array_equal(r0['val'], r1['val'])                         # replicated state is bit-identical
    # each rank fills pts for the rows it owns; together they cover the coreset
    pts = np.where(np.isnan(r0['pts']), r1['pts'], r0['pts'])
    assert np.array_equal(pts, Z[ref.idcs])


def test_two_ranks_one_gpu_beta_coreset(tmp_path):
    import beta_cores_amd as bc
    Z, th = linreg_problem()
    D = Z.shape[1] - 1
    E = np.random.RandomState(3).randn(th.shape[0], D)

    def sampler(sz, wts, pts):
        if pts.shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, Z.shape[1]))
        mu, L, _ = M.linreg_weighted_post(np.zeros(D), np.eye(D), 1.0, pts, wts)
        return mu + E.dot(L.T)
    ref = C.RefGreedyVI(Z, lambda pts, t: C.project_f(lambda z, tt, b: M.linreg_beta_lik(z, tt, b, 1.0), pts, t, 0.1),
                        lambda w, p: sampler(0, w, p), 5, lambda i: 0.1 / (1. + i))
    ref.build(6)
    r0, r1 = launch('gpu_bcores', tmp_path)
    for r in (r0, r1):
        np.testing.assert_array_equal(r['idx'], ref.idcs)
        np.testing.assert_allclose(r['val'], ref.wts, rtol=1e-5, atol=1e-12)
        assert np.array_equal(r['pts'], ref.pts)                        # selected rows are broadcast exactly


@pytest.mark.parametrize('mode', ['gpu_nccl1', 'gpu_nccl1_torch'])
def test_rccl_exchange_path_single_rank(tmp_path, mode):
    """`bench.py --gpus N` runs the fused loop with the record all-gather over RCCL -- issued by the C library
    itself (native, default) or by torch.distributed on the kernels' stream (fallback); rehearse exactly those
    code paths with a 1-rank NCCL group (all this box has is one GPU)."""
    import beta_cores_amd as bc
    Z, th = linreg_problem()
    single = bc.HilbertCoreset(Z, bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0)))
    single.build(25, 25)
    (r0,) = launch(mode, tmp_path, world=1)
    np.testing.assert_array_equal(r0['idx'], single.idcs)
    assert np.array_equal(r0['val'], single.wts)                       # same kernels, same order: bit identical
    np.testing.assert_array_equal(r0['trace_f'], single.snnls._eng.trace()[0])
    assert np.array_equal(r0['pts'], Z[single.idcs])
    assert int(r0['next_f']) == single.snnls._select()


def test_native_colsum_matches_local_sum(tmp_path):
    """bc_phi_colsum_all (RCCL all-gather + rank-order sum inside the library) on a 1-rank group returns this
    rank's own column sums bit for bit."""
    (r0,) = launch('gpu_nccl1', tmp_path, world=1)
    assert np.array_equal(r0['colsum_native'], r0['colsum_local'])


@pytest.mark.parametrize('mode,world', [('gpu_overflow', 2), ('gpu_nccl1_overflow', 1)])
def test_prefilter_overflow_across_ranks(tmp_path, mode, world):
    """A pre-filter overflow on ANY rank makes every rank redo that step with the exact fp64 sweep (the marker
    travels in the gathered records): results equal the single-rank fp64-sweep run, in the fused loop over both
    transports and in the step-wise protocol."""
    import os
    import beta_cores_amd as bc
    from dist_worker import overflow_problem
    phi = overflow_problem()
    b = phi.sum(axis=0)
    old = os.environ.get('BC_PREFILTER')
    os.environ['BC_PREFILTER'] = '0'
    try:
        refs = {nm: cls(phi.T, b) for nm, cls in (('giga', bc.snnls.GIGA), ('fw', bc.snnls.FrankWolfe))}
    finally:
        if old is None:
            os.environ.pop('BC_PREFILTER', None)
        else:
            os.environ['BC_PREFILTER'] = old
    res = launch(mode, tmp_path, world=world)
    for nm, ref in refs.items():
        ref.build(20)
        ridx, rval = ref.sparse_weights()
        nxt = ref._select()
        for r in res:
            np.testing.assert_array_equal(r[nm + '_trace_f'], ref._eng.trace()[0])
            np.testing.assert_array_equal(r[nm + '_idx'], ridx)
            if nm == 'fw' and world > 1:
                # Frank-Wolfe scales by the sum of ALL row norms: two shard totals added in rank order instead of one
                # total -- the one quantity of the solver that depends on the world size, in the last bits
                np.testing.assert_allclose(r[nm + '_val'], rval, rtol=1e-12)
                np.testing.assert_allclose(float(r[nm + '_err']), ref.error(), rtol=1e-12)
            else:
                np.testing.assert_array_equal(r[nm + '_val'], rval)
                assert float(r[nm + '_err']) == ref.error()
            assert int(r[nm + '_next']) == nxt
        assert sum(int(r[nm + '_fallbacks']) for r in res) >= 1
