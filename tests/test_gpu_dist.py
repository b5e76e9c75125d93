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


def test_four_ranks_one_gpu_hilbert(tmp_path):
    """World size 4 (four ranks sharing GPU 0 over gloo; the pool allows six processes on a card): each rank sweeps a quarter
    of the rows with the real kernels, the replicated finish picks among FOUR records per step -- the same coreset as the
    single-rank device run and the oracle, bit-identical state on all ranks."""
    import beta_cores_amd as bc
    Z, th = linreg_problem()
    ref = C.RefHilbert(Z, lambda z, t: M.linreg_loglik(z, t, 1.0), th, RefGIGA)
    ref.build(25, 25)
    single = bc.HilbertCoreset(Z, bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0)))
    single.build(25, 25)
    res = launch('gpu_hilbert', tmp_path, world=4, timeout=400)
    for r in res:
        np.testing.assert_array_equal(r['idx'], ref.idcs)
        np.testing.assert_allclose(r['val'], single.wts, rtol=1e-9)
        np.testing.assert_array_equal(r['trace_f'], single.snnls._eng.trace()[0])
        assert np.array_equal(r['val'], res[0]['val'])
    pts = res[0]['pts']
    for r in res[1:]:
        pts = np.where(np.isnan(pts), r['pts'], pts)
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


@pytest.mark.parametrize('mode,world', [('gpu_overflow', 2), ('gpu_nccl1_overflow', 1), ('gpu_overflow4', 2), ('gpu_nccl1_overflow4', 1)])
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


# ------------------------------------------------------------------ row-sharded forms of the remaining reference modes (round 3)
def _merge_pts(a, b):
    return np.where(np.isnan(a), b, a)


@pytest.mark.parametrize('what', ['f8', 'f11'])
def test_sharded_grouped_tangent_spaces_reproduce_goldens(tmp_path, what):
    """Grouped (F8) and grouped + sub-sampled (F11) tangent spaces with the rows sharded over two ranks (bcores.py:46-61):
    group sums completed across ranks in rank order, random groups / rows drawn from the shared global stream --
    selected groups, indices, weights (1e-5) and the RNG position equal the reference's on both ranks."""
    from conftest import load_golden
    g = load_golden('f8_grouped_vi' if what == 'f8' else 'f11_grouped_subsampled')
    Z = g['Z']
    r0, r1 = launch('gpu_golden_' + what, tmp_path)
    nb = 4 if what == 'f8' else 5
    for nm in ('bcores', 'svi'):
        for m in range(nb):
            for r in (r0, r1):
                np.testing.assert_array_equal(r['%s_idcs_%d' % (nm, m)], g['%s_allidcs_%d' % (nm, m)])
                np.testing.assert_array_equal(r['%s_groups_%d' % (nm, m)], g['%s_groups_%d' % (nm, m)])
                np.testing.assert_allclose(r['%s_w_%d' % (nm, m)], g['%s_allw_%d' % (nm, m)], rtol=1e-5, atol=1e-12)
                assert np.array_equal(r['%s_pts_%d' % (nm, m)], Z[g['%s_allidcs_%d' % (nm, m)]])      # rows fetched from their owners
            assert np.array_equal(r0['%s_w_%d' % (nm, m)], r1['%s_w_%d' % (nm, m)])                    # replicated state
        if what == 'f11':
            for r in (r0, r1):
                assert float(r['%s_rng_after' % nm]) == float(g['%s_rng_after' % nm])


def test_sharded_subsampled_tangent_space_reproduces_golden_f9(tmp_path):
    """Sub-sampled selection and gradient steps (bcores.py:51-55) over sharded rows: every rank projects the drawn rows it
    owns, the argmax compares (score, position in the draw) across ranks -- golden F9 selections / weights / RNG position."""
    from conftest import load_golden
    g = load_golden('f9_subsampled_gaussian')
    r0, r1 = launch('gpu_golden_f9', tmp_path)
    for nm in ('bcores', 'svi'):
        for m in range(6):
            for r in (r0, r1):
                np.testing.assert_array_equal(r['%s_idcs_%d' % (nm, m)], g['%s_allidcs_%d' % (nm, m)])
                np.testing.assert_allclose(r['%s_w_%d' % (nm, m)], g['%s_allw_%d' % (nm, m)], rtol=1e-5, atol=1e-12)
        for r in (r0, r1):
            assert float(r['%s_rng_after' % nm]) == float(g['%s_rng_after' % nm])


def test_sharded_hilbert_zero_rows_and_subsampling(tmp_path):
    """HilbertCoreset over sharded rows with all-zero projection rows (golden F12: the index shift of hilbert.py:16,32 from
    the exchanged list of dropped rows) and with n_subsample (hilbert.py:12-15: shared draw, rows collected from their
    owners, the sub-sample sharded again by position) -- equal to the reference / to the single-rank device run."""
    import beta_cores_amd as bc
    from conftest import load_golden
    g = load_golden('f12_constant_rows')
    r0, r1 = launch('gpu_golden_f12', tmp_path)
    seen = 0
    for kind in ('lin', 'log'):
        for S in (100, 200):
            tag = '%s_S%d_' % (kind, S)
            for an in ('giga', 'fw'):
                if tag + an + '_idcs' not in r0.files:
                    continue
                seen += 1
                Z = g[tag + 'Z']
                for r in (r0, r1):
                    np.testing.assert_array_equal(r[tag + an + '_idcs'], g[tag + an + '_idcs'])
                    np.testing.assert_allclose(r[tag + an + '_wts'], g[tag + an + '_wts'], rtol=1e-5)
                pts = _merge_pts(r0[tag + an + '_pts'], r1[tag + an + '_pts'])
                assert np.array_equal(pts, Z[g[tag + an + '_idcs']])
    assert seen >= 2
    Z, th = linreg_problem()
    np.random.seed(77)
    single = bc.HilbertCoreset(Z, bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0)), n_subsample=700)
    single.build(20, 20)
    for r in (r0, r1):
        np.testing.assert_array_equal(r['sub_idcs'], single.idcs)
        np.testing.assert_allclose(r['sub_wts'], single.wts, rtol=1e-9)
    assert np.array_equal(_merge_pts(r0['sub_pts'], r1['sub_pts']), Z[single.idcs])


def _gpus():
    import torch
    return torch.cuda.device_count()


def _beta_reference():
    Z, th = linreg_problem(n=9000)
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
    return ref


def test_fused_gradient_over_the_native_communicator_single_rank(tmp_path):
    """BetaCoreset with `comm` on a 1-rank NCCL group (BC_FORCE_EXCHANGE): every gradient is one bc_vi_gradient call whose
    column sums go through ncclAllGather + the rank-order sum inside the library -- the code path of a multi-GPU run."""
    ref = _beta_reference()
    (r0,) = launch('gpu_nccl1_bcores', tmp_path, world=1)
    np.testing.assert_array_equal(r0['idx'], ref.idcs)
    np.testing.assert_allclose(r0['val'], ref.wts, rtol=1e-5, atol=1e-12)
    assert np.array_equal(r0['pts'], ref.pts)
    assert int(r0['fused_calls']) == 6 * 5


@pytest.mark.skipif(_gpus() < 2, reason='needs two GPUs: world > 1 over RCCL (one process per GPU)')
def test_two_gpus_native_rccl_paths(tmp_path):
    """World size 2 over RCCL, one GPU per rank (skipped on one-GPU boxes): the fused greedy loop with the in-library
    all-gather, bc_phi_colsum_all / k_sum_rank_order with two ranks, the pre-filter overflow marker travelling through
    the gathered records, and the fused beta-Cores gradient with its collective -- against single-rank results."""
    import beta_cores_amd as bc
    Z, th = linreg_problem()
    single = bc.HilbertCoreset(Z, bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0)))
    single.build(25, 25)
    r0, r1 = launch('gpu_ncclN_hilbert', tmp_path, world=2)
    for r in (r0, r1):
        np.testing.assert_array_equal(r['idx'], single.idcs)
        np.testing.assert_allclose(r['val'], single.wts, rtol=1e-9)
        np.testing.assert_array_equal(r['trace_f'], single.snnls._eng.trace()[0])
        assert np.array_equal(r['colsum_native'], r['colsum_host'])           # device rank-order sum == host rank-order sum
    assert np.array_equal(r0['val'], r1['val'])
    ref = _beta_reference()
    for r in launch('gpu_ncclN_bcores', tmp_path, world=2):
        np.testing.assert_array_equal(r['idx'], ref.idcs)
        np.testing.assert_allclose(r['val'], ref.wts, rtol=1e-5, atol=1e-12)
        assert int(r['fused_calls']) == 6 * 5
    res = launch('gpu_ncclN_overflow', tmp_path, world=2)
    assert sum(int(r['giga_fallbacks']) for r in res) >= 1
    assert np.array_equal(res[0]['giga_val'], res[1]['giga_val']) and np.array_equal(res[0]['giga_idx'], res[1]['giga_idx'])
