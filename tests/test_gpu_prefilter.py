"""The reduced-precision (fp32 / fp16 storage) pre-filter of the sweep (bc_prefilter.hip) must change NOTHING: same selected rows (bit-exact
trace), same weights (bit-identical, they are computed from the fp64 columns), on generic, adversarial
(exact ties, near ties at 1e-12, rows parallel to the iterate) and degenerate inputs, and when its
candidate list overflows into the fp64 fallback."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[32, 16, 8, 'bb', 'lists', 4])
def prec(request, monkeypatch):
    """Storage precision of the mirror; 'bb' = the int8 mirror swept in the branch-and-bound form (csrc/bc_prefilter_bb.h:
    the sweep blocks rescore their candidates themselves), forced on for these small inputs -- 8 forces the two-pass form;
    'lists' = two-pass with the sweep blocks' own candidate lists (BC_I8_BLKLIST=1: the rescoring stage skips the tile walk);
    4 = the two-level form (csrc/bc_prefilter_i4.h: a 4-bit first level, int8 records behind it; S <= 256)."""
    if request.param == 'bb':
        monkeypatch.setenv('BC_I8_BB', '1')
        return 8
    monkeypatch.setenv('BC_I8_BB', '0')
    if request.param == 'lists':
        monkeypatch.setenv('BC_I8_BLKLIST', '1')
        return 8
    monkeypatch.setenv('BC_I8_BLKLIST', '0')
    return request.param


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


class prefilter:
    def __init__(self, on, cap=None):
        self.on, self.cap = on, cap

    def __enter__(self):
        self.old = (os.environ.get('BC_PREFILTER'), os.environ.get('BC_PREFILTER_CAP'))
        os.environ['BC_PREFILTER'] = str(int(self.on))        # 0 = off, 8 / 16 / 32 = storage precision
        if self.cap is not None:
            os.environ['BC_PREFILTER_CAP'] = str(self.cap)
        else:
            os.environ.pop('BC_PREFILTER_CAP', None)

    def __exit__(self, *a):
        for k, v in zip(('BC_PREFILTER', 'BC_PREFILTER_CAP'), self.old):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run(bc, cls, phi, steps, on, cap=None, stepwise=False):
    with prefilter(on, cap):
        s = cls(phi.T, phi.sum(axis=0))
    assert s._eng.prefilter == (8 if on == 4 else on)
    if on == 4:
        assert s._eng.prefilter_form == (3 if phi.shape[1] <= 256 else 1)
    if on == 8 and os.environ.get('BC_I8_BB') is not None:
        # (the branch-and-bound form keeps the winner's row in a 256-double LDS strip: wider projections stay two-pass)
        assert s._eng.prefilter_form == (2 if os.environ['BC_I8_BB'] == '1' and phi.shape[1] <= 256 else 1)
    if stepwise:
        s.build_stepwise(steps)
        tr = None
    else:
        s.build(steps)
        tr = s._eng.trace()
    idx, val = s._eng.sparse_weights()
    return tr, idx, val, s.error()


def same(a, b):
    (tra, ia, va, ea), (trb, ib, vb, eb) = a, b
    if tra is not None:
        assert np.array_equal(tra[0], trb[0]) and np.array_equal(tra[1], trb[1]) and np.array_equal(tra[2], trb[2])
    assert np.array_equal(ia, ib) and np.array_equal(va, vb) and ea == eb


def correlated(rng, n, s):
    base = rng.randn(n, 12).dot(rng.randn(12, s)) + 0.3 * rng.randn(n, s)
    return base - base.mean(axis=1)[:, None]


@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
@pytest.mark.parametrize('n,s', [(30000, 100), (5000, 37), (257, 8), (1, 5)])
def test_identical_to_fp64_sweep(bc, alg, n, s, prec):
    rng = np.random.RandomState(n + s)
    phi = correlated(rng, n, s)
    cls = dict(giga=bc.snnls.GIGA, fw=bc.snnls.FrankWolfe, omp=bc.snnls.OrthoPursuit)[alg]
    steps = min(n, 60 if alg != 'omp' else 20)
    same(run(bc, cls, phi, steps, prec, stepwise=(alg == 'omp')), run(bc, cls, phi, steps, 0, stepwise=(alg == 'omp')))


@pytest.mark.parametrize('alg', ['giga', 'fw'])
def test_exact_and_near_ties(bc, alg, prec):
    """Duplicated rows (exact ties -> lowest index wins) and rows perturbed in the last bits (scores a few
    ulps apart: far inside the fp32 uncertainty, so only the fp64 rescoring can order them)."""
    rng = np.random.RandomState(3)
    n, s = 6000, 64
    phi = correlated(rng, n, s)
    for src in rng.choice(n, 40, replace=False):
        for dst in rng.choice(n, 3, replace=False):
            phi[dst] = phi[src]                                            # exact duplicates at scattered indices
        dst = rng.randint(n)
        phi[dst] = phi[src] * (1. + 1e-15 * rng.randint(-4, 5, size=s))    # near duplicates
    cls = bc.snnls.GIGA if alg == 'giga' else bc.snnls.FrankWolfe
    same(run(bc, cls, phi, 80, prec), run(bc, cls, phi, 80, 0))


@pytest.mark.parametrize('cap', [1, 2, 7])
def test_candidate_overflow_falls_back_to_fp64(bc, cap, prec):
    rng = np.random.RandomState(4)
    n, s = 4000, 32
    phi = correlated(rng, n, s)
    phi[rng.choice(n, 200, replace=False)] = phi[17]                       # 200 exact copies of one row
    same(run(bc, bc.snnls.GIGA, phi, 40, prec, cap=cap), run(bc, bc.snnls.GIGA, phi, 40, 0))
    same(run(bc, bc.snnls.FrankWolfe, phi, 40, prec, cap=cap), run(bc, bc.snnls.FrankWolfe, phi, 40, 0))
    # the fallback really ran (and only then): the copies tie at the top in the first sweeps
    with prefilter(prec, cap):
        s = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    s.build(5)
    assert s._eng.prefilter_fallbacks() >= 1
    phi2 = correlated(rng, 3000, 32)
    with prefilter(prec):
        s = bc.snnls.GIGA(phi2.T, np.ones(32))
    s.build(5)
    if s._eng.prefilter_form not in (2, 3):
        assert s._eng.prefilter_fallbacks() == 0
    else:
        # b = 1 is orthogonal to every (centred) row: all 3000 scores tie at rounding noise.  The two-pass form's lists hold 4096
        # candidates; a branch-and-bound block rescores at most 48 rows per sweep and hands such a step to the exact sweep, and
        # so does the two-level form once its blocks are left with more rows in play than their lists and the spill list hold
        assert s._eng.prefilter_fallbacks() >= 1
        with prefilter(prec):
            s = bc.snnls.GIGA(phi2.T, phi2.sum(axis=0))
        s.build(5)
        assert s._eng.prefilter_fallbacks() == 0


def test_fallback_with_many_tiles_and_partial_last_tile(bc, prec):
    """Overflow on a shard large enough that every helper block of the in-launch fp64 fallback has tiles,
    with a ragged last tile; stepwise (select/reweight) and fused paths."""
    rng = np.random.RandomState(12)
    n, s = 150_001, 24
    phi = correlated(rng, n, s)
    phi[rng.choice(n, 3000, replace=False)] = phi[5]
    a = run(bc, bc.snnls.GIGA, phi, 30, prec, cap=3)
    same(a, run(bc, bc.snnls.GIGA, phi, 30, 0))
    b = run(bc, bc.snnls.FrankWolfe, phi, 30, prec, cap=3, stepwise=True)
    same(b, run(bc, bc.snnls.FrankWolfe, phi, 30, 0, stepwise=True))


F1 = load_golden('f1_snnls')


@pytest.mark.parametrize('case', list(F1['cases']))
def test_degenerate_designs_identical(bc, case, prec):
    """bin / colinear / axis-aligned designs: ties, zero-error states, precision-limit retries."""
    X = F1[case + '_X']
    steps = min(X.shape[0], 25)
    for cls in (bc.snnls.GIGA, bc.snnls.FrankWolfe):
        same(run(bc, cls, X, steps, prec), run(bc, cls, X, steps, 0))


def test_zero_rows_are_masked(bc, prec):
    rng = np.random.RandomState(6)
    phi = correlated(rng, 3000, 20)
    phi[[0, 5, 1000, 2999]] = 0.
    for on in (prec, 0):
        with prefilter(on):
            s = bc.snnls.GIGA(phi.T, phi.sum(axis=0), allow_zero_rows=True)
        s.build(30)
        res = s._eng.sparse_weights(), s._eng.trace()[0]
        if on:
            first = res
    assert np.array_equal(first[0][0], res[0][0]) and np.array_equal(first[0][1], res[0][1]) and np.array_equal(first[1], res[1])
    assert not set(first[0][0].tolist()) & {0, 5, 1000, 2999}


def test_wide_dynamic_range_rows(bc, prec):
    """Rows dominated by one coordinate: after normalisation most entries fall into (or below) the fp16
    subnormal range, which the fp16 bound covers with its absolute term."""
    rng = np.random.RandomState(8)
    n, s = 20000, 48
    phi = rng.randn(n, s) * 10. ** rng.uniform(-9, 0, size=(n, s))
    phi[np.arange(n), rng.randint(s, size=n)] = rng.choice([-1., 1.], n) * rng.uniform(1, 3, n)
    for cls in (bc.snnls.GIGA, bc.snnls.FrankWolfe):
        same(run(bc, cls, phi, 60, prec), run(bc, cls, phi, 60, 0))


@pytest.mark.parametrize('s', [300, 1000])
def test_many_samples(bc, s, prec):
    rng = np.random.RandomState(s)
    phi = correlated(rng, 4000, s)
    same(run(bc, bc.snnls.GIGA, phi, 40, prec), run(bc, bc.snnls.GIGA, phi, 40, 0))


@pytest.mark.parametrize('noise,ncopies', [(1e-4, 1500), (1e-9, 300), (3e-3, 3000)])
def test_clusters_of_near_duplicates(bc, prec, noise, ncopies):
    """Hundreds to thousands of rows within the reduced-precision window of the best one: the candidate list is
    long (per-thread exact rescoring instead of the wave-per-candidate path), tiles hold more than four local
    candidates (the int8 mirror then hands over whole tiles), yet nothing overflows into the fp64 fallback for the
    smaller clusters -- and the selections must still be those of the fp64 sweep."""
    rng = np.random.RandomState(21)
    n, s = 40000, 40
    phi = correlated(rng, n, s)
    src = phi[123].copy()
    where = rng.choice(n, ncopies, replace=False)
    phi[where] = src * (1. + noise * rng.randn(ncopies, s))
    for cls in (bc.snnls.GIGA, bc.snnls.FrankWolfe):
        same(run(bc, cls, phi, 30, prec), run(bc, cls, phi, 30, 0))


def test_prefilter_statistics(bc, prec):
    rng = np.random.RandomState(22)
    phi = correlated(rng, 60000, 50)
    with prefilter(prec):
        sv = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    sv.build(25)
    sweeps, cands, falls = sv._eng.prefilter_stats()
    assert sweeps >= 25 and falls == 0
    # at least the winner each time, and a selective filter.  (Branch-and-bound form on this small input -- one tile per wave,
    # every block's first posts are judged against a bound that has seen next to nothing: it is meant for waves that walk many
    # tiles -- so only the trivial bound is asked of it here: fewer rows than a block may rescore, per block.)
    assert sweeps <= cands <= (64 if sv._eng.prefilter_form != 2 else 48 * 59) * sweeps


@pytest.mark.parametrize('n,s', [(30000, 100), (30081, 64), (128 * 235 + 1, 104), (257, 8), (70000, 97)])
def test_int8_mirror_builders_agree(bc, n, s, monkeypatch):
    """S <= 104: the mirror is built in one pass over Phi (k_build_i8_r: the row stays in registers); otherwise, or with
    BC_BUILD_I8_TWO_PASS=1, in two (k_build_i8).  Same digits, same scales, same measured deltas: the sweeps select the same
    candidates step for step (identical rescoring counters) and the solver traces are identical -- on row counts whose last
    mirror tile is only partly backed by Phi tiles (an odd number of 128-row tiles; the rows past the end are dead rows)."""
    rng = np.random.RandomState(n + s)
    phi = correlated(rng, n, s)
    out = []
    for two_pass in ('0', '1'):
        monkeypatch.setenv('BC_BUILD_I8_TWO_PASS', two_pass)
        with prefilter(8):
            sv = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
        assert sv._eng.prefilter == 8
        sv.build(25)
        out.append((sv._eng.trace(), sv._eng.sparse_weights(), sv._eng.prefilter_stats()))
    (ta, wa, sa), (tb, wb, sb) = out
    assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
    assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1])
    assert sa == sb


def test_two_level_watch_puts_the_first_level_aside_where_it_does_not_select(bc):
    """The host watches the share of rows the 4-bit first level passes on (csrc/bc_prefilter.hip: bc_pref_adapt, at the end
    of every build call): on rows whose scores the 4-bit bounds cannot tell apart (here
    linear-regression rows under widely scattered parameter samples: a tenth of them and more pass) the solver goes back to the one-level int8 sweep; selections are the fp64 sweep's either way."""
    import torch
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    n, d, s = 400_000, 32, 100
    Z = torch.randn((n, d + 1), generator=g, dtype=torch.float64, device='cuda')
    th = np.random.default_rng(1).standard_normal((s, d)) * 0.3
    phi = bc.DeviceProjector(lambda k, w, p: th, s, bc.likelihoods.LinearRegression(1.0)).project(bc.DeviceData.from_torch(Z))
    with prefilter(4):
        sv = bc.snnls.GIGA(phi.T, phi.colsum())
    assert sv._eng.prefilter_form == 3
    sv.build(12)
    l1, listed, _ = sv._eng.prefilter_levels()
    assert l1 >= 4 and listed > 0.04 * l1 * n
    N_ = __import__('beta_cores_amd._native', fromlist=['x'])
    import ctypes as C
    form = C.c_int()
    N_.call('bc_snnls_prefilter_form', sv._eng.h, C.byref(form))
    assert form.value == 1                                   # put aside
    sv.build(12)
    assert sv._eng.prefilter_levels()[0] == l1               # no further first-level sweeps
    with prefilter(0):
        ref = bc.snnls.GIGA(phi.T, phi.colsum())
    ref.build(24)
    assert np.array_equal(sv._eng.trace()[0], ref._eng.trace()[0])


@pytest.mark.parametrize('alg', ['giga', 'fw'])
def test_two_level_form_across_reset_and_repeated_build_calls(bc, alg):
    """The two-level form keeps state between sweeps (the seeds of its first level, the host's watch): a solver that is built in
    several calls, reset and built again must return what a fresh fp64-sweep solver returns every time."""
    rng = np.random.RandomState(31)
    phi = correlated(rng, 50000, 64)
    cls = bc.snnls.GIGA if alg == 'giga' else bc.snnls.FrankWolfe

    def fresh(steps):
        with prefilter(0):
            r = cls(phi.T, phi.sum(axis=0))
        r.build(steps)
        return r._eng.trace()[0], r._eng.sparse_weights(), r.error()

    with prefilter(4):
        sv = cls(phi.T, phi.sum(axis=0))
    assert sv._eng.prefilter_form == 3
    for k in (1, 1, 5, 13):                              # 20 steps in four calls (the first sweep of a solver is the int8 one)
        sv.build(k)
    t, w, e = fresh(20)
    assert np.array_equal(sv._eng.trace()[0], t) and np.array_equal(sv._eng.sparse_weights()[0], w[0])
    assert np.array_equal(sv._eng.sparse_weights()[1], w[1]) and sv.error() == e
    sv.reset()                                           # the seeds now describe the END of the previous run: still valid bounds
    sv.build(12)
    t, w, e = fresh(12)
    assert np.array_equal(sv._eng.trace()[0], t) and np.array_equal(sv._eng.sparse_weights()[1], w[1]) and sv.error() == e
    assert sv._eng.prefilter_levels()[0] >= 4            # (on 50 000 rows the host's watch may put the first level aside after a few calls)


def _fuzz_problem(rng, n, s, kind):
    if kind == 0:
        phi = rng.randn(n, s)
    elif kind == 1:
        r = max(1, s // 6)
        phi = rng.randn(n, r).dot(rng.randn(r, s)) + 0.2 * rng.randn(n, s)
    elif kind == 2:                                                  # sparse rows
        phi = rng.randn(n, s) * (rng.rand(n, s) < 0.15)
        phi[np.abs(phi).sum(axis=1) == 0, 0] = 1.0
    elif kind == 3:                                                  # row scales over 12 decades
        phi = rng.randn(n, s) * 10.0 ** rng.uniform(-6, 6, size=(n, 1))
    elif kind == 4:                                                  # 50 copies of each prototype, perturbed at 1e-6
        base = rng.randn(max(1, n // 50), s)
        phi = base[rng.randint(base.shape[0], size=n)] * (1 + 1e-6 * rng.randn(n, s))
    else:                                                            # element scales over 5 decades
        phi = rng.randn(n, s) * 10.0 ** rng.uniform(-5, 0, size=(n, s))
    return phi - phi.mean(axis=1)[:, None] if s > 1 else phi


def test_random_problems_all_mirrors(bc):
    """A fixed-seed slice of the fuzz run recorded in profiles/r01_notes.md (5986 random problems x 3 mirrors, no
    mismatch): sizes 1..200k, S 1..130, six data families, GIGA and Frank-Wolfe."""
    rng = np.random.RandomState(77)
    done = 0
    while done < 40:
        n = int(10 ** rng.uniform(0, 5.0)); s = rng.randint(1, 131); kind = rng.randint(6)
        phi = _fuzz_problem(rng, n, s, kind)
        if not np.isfinite(phi).all() or (np.linalg.norm(phi, axis=1) == 0).any():
            continue
        cls = bc.snnls.GIGA if rng.rand() < 0.6 else bc.snnls.FrankWolfe
        steps = min(n, rng.randint(1, 40))
        ref = run(bc, cls, phi, steps, 0)
        for prec in (8, 16, 32, 4):
            same(run(bc, cls, phi, steps, prec), ref)
        os.environ['BC_I8_BB'] = '1'
        try:
            same(run(bc, cls, phi, steps, 8), ref)
        finally:
            os.environ.pop('BC_I8_BB', None)
        done += 1


def test_million_rows_identical(bc):
    import torch
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    n, d, s = 1_000_000, 32, 100
    Z = torch.randn((n, d + 1), generator=g, dtype=torch.float64, device='cuda')
    th = np.random.default_rng(1).standard_normal((s, d)) * 0.3
    data = bc.DeviceData.from_torch(Z)
    phi = bc.DeviceProjector(lambda k, w, p: th, s, bc.likelihoods.LinearRegression(1.0)).project(data)
    out = []
    for on in (32, 16, 8, 'bb', 4, 0):
        if on == 'bb':
            os.environ['BC_I8_BB'] = '1'
        try:
            with prefilter(8 if on == 'bb' else on):
                sv = bc.snnls.GIGA(phi.T, phi.colsum())
        finally:
            os.environ.pop('BC_I8_BB', None)
        assert sv._eng.prefilter == (8 if on in ('bb', 4) else on) and (on != 'bb' or sv._eng.prefilter_form == 2)
        assert on != 4 or sv._eng.prefilter_form == 3
        sv.build(60)
        if on == 4:
            l1, listed, refined = sv._eng.prefilter_levels()
            assert 1 <= l1 and 0 < refined == listed      # (the first sweep has no seeds: plain int8)
        out.append((sv._eng.trace(), sv._eng.sparse_weights(), sv.error()))
    tb, wb, eb = out[-1]
    for ta, wa, ea in out[:-1]:
        assert np.array_equal(ta[0], tb[0]) and np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1]) and ea == eb


def test_int8_sweep_many_tiles_per_wave(bc, monkeypatch):
    """The int8 sweep parks per-tile results in LDS and writes them out in bursts of 32 tiles per wave.  With one wave
    per SIMD-quarter of the chip (BC_PREF_WAVES_PER_CU=1) and 2.3M rows a wave walks 35 tiles: the burst in the middle of
    the walk and the one at its end must both land where the rescoring looks for them."""
    import torch
    g = torch.Generator(device='cuda'); g.manual_seed(21)
    n, d, s = 2_300_000, 8, 16
    Z = torch.randn((n, d + 1), generator=g, dtype=torch.float64, device='cuda')
    th = np.random.default_rng(2).standard_normal((s, d)) * 0.3
    data = bc.DeviceData.from_torch(Z)
    phi = bc.DeviceProjector(lambda k, w, p: th, s, bc.likelihoods.LinearRegression(1.0)).project(data)
    monkeypatch.setenv('BC_PREF_WAVES_PER_CU', '1')
    out = []
    for on in (8, 0):
        with prefilter(on):
            sv = bc.snnls.GIGA(phi.T, phi.colsum())
        assert sv._eng.prefilter == on
        sv.build(25)
        out.append((sv._eng.trace(), sv._eng.sparse_weights(), sv.error()))
    (ta, wa, ea), (tb, wb, eb) = out
    assert np.array_equal(ta[0], tb[0]) and np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1]) and ea == eb


@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
@pytest.mark.parametrize('n,s', [(30000, 100), (5000, 37), (300, 5), (70000, 300)])
def test_sweep_vector_quantised_once_per_step_changes_nothing(bc, alg, n, s):
    """Round 4: the step kernels leave the int8 digits of the next sweep vector behind (bc_i8_quant.h) and the sweep blocks
    copy them instead of quantising the vector in every block's prologue.  Same arithmetic on both routes: traces, weights,
    errors AND the candidate statistics of the pre-filter (sweeps, rows rescored, fallbacks) are identical to BC_I8_QV=0."""
    rng = np.random.RandomState(n + 3 * s)
    phi = correlated(rng, n, s)
    cls = dict(giga=bc.snnls.GIGA, fw=bc.snnls.FrankWolfe, omp=bc.snnls.OrthoPursuit)[alg]
    steps = 40 if alg != 'omp' else 15
    res, stats = [], []
    for qv in ('1', '0'):
        old = os.environ.get('BC_I8_QV')
        os.environ['BC_I8_QV'] = qv
        try:
            with prefilter(8):
                sv = cls(phi.T, phi.sum(axis=0))
        finally:
            if old is None:
                os.environ.pop('BC_I8_QV', None)
            else:
                os.environ['BC_I8_QV'] = old
        assert sv._eng.prefilter == 8
        if alg == 'omp':
            sv.build_stepwise(steps)
            tr = None
        else:
            sv.build(steps)
            tr = sv._eng.trace()
        idx, val = sv._eng.sparse_weights()
        res.append((tr, idx, val, sv.error()))
        stats.append(tuple(sv._eng.prefilter_stats()))
    same(res[0], res[1])
    assert stats[0] == stats[1] and stats[0][0] > 0          # (a small problem may reach its numeric limit before `steps`)
    same(res[0], run(bc, cls, phi, steps, 0, stepwise=(alg == 'omp')))      # and both equal the fp64 sweep
