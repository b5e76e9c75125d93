"""GPU parity: device SNNLS solvers (through the C ABI) vs the CPU oracle and the
golden fixtures generated from the reference.  Bar: selected indices bit-exact,
weights within 1e-5 relative (north_star); in practice they agree to ~1e-10."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

WTOL = 1e-5


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


def _algs(bc):
    return dict(giga=bc.snnls.GIGA, fw=bc.snnls.FrankWolfe, omp=bc.snnls.OrthoPursuit)


def _oracle_algs():
    from oracle import RefGIGA, RefFrankWolfe, RefOrthoPursuit
    return dict(giga=RefGIGA, fw=RefFrankWolfe, omp=RefOrthoPursuit)


def run_stepwise(s, steps, fused):
    """build(1) x steps; returns per-step dense weights, errors, limit flags."""
    n = s.n_total
    W = np.zeros((steps, n))
    err = np.zeros(steps)
    lim = np.zeros(steps, dtype=np.int8)
    for m in range(steps):
        if fused:
            s.build(1)
        else:
            if not s.reached_numeric_limit:
                s.build_stepwise(1)
        W[m] = s.weights()
        err[m] = s.error()
        lim[m] = s.reached_numeric_limit
    return W, err, lim


# ------------------------------------------------------------------ Phi storage round trips
@pytest.mark.parametrize('n,s', [(1, 1), (5, 3), (127, 7), (128, 10), (129, 33), (1000, 100), (4099, 200), (300, 257)])
def test_phi_layout_roundtrip_and_stats(bc, n, s):
    rng = np.random.RandomState(n * 7 + s)
    phi = rng.randn(n, s)
    d = bc.DevicePhi.from_host(phi)
    assert d.shape == (n, s)
    assert np.array_equal(d.to_host(), phi)                       # pure data movement: bit exact
    np.testing.assert_allclose(d.colsum(), phi.sum(axis=0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(d.norms(), np.sqrt((phi ** 2).sum(axis=1)), rtol=1e-14)
    nz, ns = d.norm_stats()
    assert nz == 0
    np.testing.assert_allclose(ns, np.sqrt((phi ** 2).sum(axis=1)).sum(), rtol=1e-12)
    idx = rng.randint(0, n, size=min(n, 9))
    assert np.array_equal(d.rows(idx), phi[idx])
    v = rng.randn(s)
    np.testing.assert_allclose(d.matvec(v), phi.dot(v), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize('n,s', [(1, 4), (130, 9), (5000, 100), (20000, 64)])
def test_k3_argmax_matches_numpy(bc, n, s):
    rng = np.random.RandomState(n + s)
    phi = rng.randn(n, s)
    phi[n // 2] = 0.                                               # a zero row must be skipped
    d = bc.DevicePhi.from_host(phi)
    nrm = np.sqrt((phi ** 2).sum(axis=1))
    ok = nrm > 0
    # dot mode
    v = rng.randn(s)
    sc = np.full(n, -np.inf)
    sc[ok] = phi[ok].dot(v) / nrm[ok] / 3.0
    f, val = d.argmax(v, mode=1, post_div=3.0)
    if ok.any():
        assert f == int(np.argmax(sc))
        assert abs(val - sc.max()) <= 1e-12 * abs(sc.max())
    else:
        assert f == -1
    # GIGA mode
    c = rng.randn(s); c /= np.linalg.norm(c)
    x = rng.randn(s); x /= np.linalg.norm(x)
    An = phi[ok] / nrm[ok][:, None]
    s0, s1 = An.dot(c), An.dot(x)
    good = np.logical_and(s1 > -1. + 1e-14, 1. - s1 ** 2 > 0.)
    den = np.where(good, np.sqrt(np.abs(1. - s1 ** 2)), np.inf)
    full = np.full(n, -np.inf)
    full[ok] = s0 / den
    f, val = d.argmax(np.stack((c, x), axis=1), mode=0)
    if ok.any():
        assert f == int(np.argmax(full))


def test_k3_ties_take_lowest_index(bc):
    phi = np.zeros((300, 6))
    phi[:, 2] = 1.5                                               # all rows identical -> exact ties
    d = bc.DevicePhi.from_host(phi)
    f, _ = d.argmax(np.ones(6), mode=1)
    assert f == 0
    phi[0] = 0.
    phi[1, 2] = -1.5
    d = bc.DevicePhi.from_host(phi)
    f, _ = d.argmax(np.ones(6), mode=1)
    assert f == 2                                                 # row 0 masked (zero), row 1 negative


# ------------------------------------------------------------------ F1 known answers
F1 = load_golden('f1_snnls')
EXACT = [c for c in F1['cases'] if c.startswith('gauss_N') or c.startswith('axis_aligned')]
DEGENERATE = [c for c in F1['cases'] if c not in EXACT]


@pytest.mark.parametrize('case', EXACT)
@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
@pytest.mark.parametrize('fused', [True, False])
def test_f1_exact_sequences(bc, case, alg, fused):
    X = F1[case + '_X']
    Wg, eg, lg = F1['%s_%s_W' % (case, alg)], F1['%s_%s_err' % (case, alg)], F1['%s_%s_lim' % (case, alg)]
    steps = Wg.shape[0]
    s = _algs(bc)[alg](X.T, X.sum(axis=0))
    W, err, lim = run_stepwise(s, steps, fused)
    # compare while the reference is in its well-conditioned regime (error above rounding noise)
    scale = np.sqrt((X.sum(axis=0) ** 2).sum())
    for m in range(steps):
        if eg[m] < 1e-9 * scale or lg[m]:
            break
        assert np.array_equal(W[m] > 0, Wg[m] > 0), 'support differs at step %d' % m
        np.testing.assert_allclose(W[m], Wg[m], rtol=WTOL, atol=1e-12)
        np.testing.assert_allclose(err[m], eg[m], rtol=1e-6, atol=1e-9 * scale)
    assert np.all(W >= 0)


@pytest.mark.parametrize('case', DEGENERATE)
@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
def test_f1_degenerate_invariants(bc, case, alg):
    """bin / colinear designs: exact ties and zero-error states make the reference's own
    selection unstable (tests/test_snnls/test_deterministic.py:102-104), so check the
    invariants that file lists instead: nnz <= m, w >= 0, monotone error, error() consistent."""
    X = F1[case + '_X']
    steps = F1['%s_%s_W' % (case, alg)].shape[0]
    s = _algs(bc)[alg](X.T, X.sum(axis=0))
    xs = X.sum(axis=0)
    prev = np.inf
    for m in range(1, steps + 1):
        s.build(1)
        w = s.weights()
        assert (w > 0).sum() <= m and (w > 0).sum() == s.size() and np.all(w >= 0)
        e = np.sqrt((((w[:, None] * X).sum(axis=0) - xs) ** 2).sum())
        assert e - prev < 1e-6
        assert abs(s.error() - e) < 1e-6
        prev = e
    if 'colinear' in case and alg == 'giga':
        assert prev < 1e-6 * max(1., np.sqrt((xs ** 2).sum()))
    s.reset()
    assert s.size() == 0 and not s.reached_numeric_limit and abs(s.error() - np.sqrt((xs ** 2).sum())) < 1e-9


# ------------------------------------------------------------------ F3 pipeline goldens (Phi from the reference)
@pytest.mark.parametrize('nm', ['ll', 'bl'])
@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
@pytest.mark.parametrize('fused', [True, False])
def test_f3_solver_on_reference_phi(bc, nm, alg, fused):
    g = load_golden('f3_hilbert_linreg')
    phi = g['phi_' + nm]
    key = '%s_%s_' % (nm, alg)
    steps = g[key + 'sel'].shape[0]
    s = _algs(bc)[alg](phi.T, phi.sum(axis=0))
    if fused:
        s.build(steps)
    else:
        s.build_stepwise(steps)
    idx, val = s.sparse_weights()
    np.testing.assert_array_equal(idx, g[key + 'idcs'])
    np.testing.assert_allclose(val, g[key + 'wts'], rtol=WTOL)
    np.testing.assert_allclose(s.error(), g[key + 'err'][-1], rtol=1e-7)
    if alg != 'omp' and fused:
        f, st, er = s._eng.trace()
        np.testing.assert_array_equal(f, g[key + 'sel'])
        np.testing.assert_allclose(er, g[key + 'err'], rtol=1e-7)
    s.optimize()
    idx, val = s.sparse_weights()
    np.testing.assert_array_equal(idx, g[key + 'opt_idcs'])
    np.testing.assert_allclose(val, g[key + 'opt_wts'], rtol=WTOL)


# ------------------------------------------------------------------ larger seeded parity vs the oracle
@pytest.mark.parametrize('alg,steps', [('giga', 80), ('fw', 80), ('omp', 25)])
def test_seeded_parity_vs_oracle(bc, alg, steps):
    rng = np.random.RandomState(11)
    n, s_ = 30000, 100
    base = rng.randn(n, 12).dot(rng.randn(12, s_)) + 0.3 * rng.randn(n, s_)     # correlated columns, like real Phi
    phi = base - base.mean(axis=1)[:, None]
    ref = _oracle_algs()[alg](phi.T, phi.sum(axis=0))
    ref.build(steps)
    dev = _algs(bc)[alg](phi.T, phi.sum(axis=0))
    dev.build(steps)
    ridx = np.where(ref.w > 0)[0]
    idx, val = dev.sparse_weights()
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_allclose(val, ref.w[ridx], rtol=WTOL)
    np.testing.assert_allclose(dev.error(), ref.error(), rtol=1e-7)
    if alg != 'omp':
        f, st, er = dev._eng.trace()
        np.testing.assert_array_equal(f, [t[0] for t in ref.trace])
        np.testing.assert_array_equal(st, [t[1] for t in ref.trace])


def test_incremental_equals_oneshot(bc):
    rng = np.random.RandomState(5)
    phi = rng.randn(5000, 40)
    a = bc.snnls.GIGA(phi.T, phi.sum(axis=0)); a.build(30)
    b = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    for _ in range(30):
        b.build(1)
    ia, va = a.sparse_weights(); ib, vb = b.sparse_weights()
    assert np.array_equal(ia, ib) and np.array_equal(va, vb)       # same kernels, same order: bit identical


# ------------------------------------------------------------------ error conventions
def test_error_conventions(bc):
    phi = np.random.RandomState(1).randn(50, 5)
    z = phi.copy(); z[7] = 0.
    for cls in (bc.snnls.GIGA, bc.snnls.FrankWolfe, bc.snnls.OrthoPursuit):
        with pytest.raises(ValueError):
            cls(z.T, z.sum(axis=0))                                 # giga.py:11-12
    with pytest.raises(bc.NumericalPrecisionError):
        bc.snnls.GIGA(phi.T, np.zeros(5))                           # giga.py:16-17
    s = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    s.build(3)
    w0 = s.weights()
    assert w0 is not s.weights() and np.array_equal(w0, s.weights())  # weights() returns a fresh copy
    s.reached_numeric_limit = True
    s.build(5)                                                      # snnls.py:32-34: returns immediately
    assert np.array_equal(w0, s.weights())
    with pytest.raises(ValueError):
        bc.snnls.GIGA(phi.T, np.zeros(4))


def test_sampling_solvers(bc):
    rng = np.random.RandomState(3)
    phi = rng.randn(400, 8)
    from oracle import RefImportanceSampling, RefUniformSampling
    for cls, ref_cls in ((bc.snnls.ImportanceSampling, RefImportanceSampling), (bc.snnls.UniformSampling, RefUniformSampling)):
        np.random.seed(42)
        ref = ref_cls(phi.T, phi.sum(axis=0)); ref.build(20)
        np.random.seed(42)
        dev = cls(phi.T, phi.sum(axis=0)); dev.build(20)
        np.testing.assert_allclose(dev.weights(), ref.w, rtol=1e-12)
        np.testing.assert_allclose(dev.error(), ref.error(), rtol=1e-9)


def test_empty_and_tiny_inputs(bc):
    """snnls.py:36-38: an empty A makes build() a logged no-op; N = 1 works (test_deterministic.py N in {1,...})."""
    s = bc.snnls.GIGA(np.zeros((4, 0)), np.ones(4))
    s.build(3)
    assert s.size() == 0 and s.weights().shape == (0,) and not s.reached_numeric_limit
    assert abs(s.error() - 2.0) < 1e-15
    x = np.array([[0.3, -1.2, 0.5]])
    for cls in (bc.snnls.GIGA, bc.snnls.FrankWolfe, bc.snnls.OrthoPursuit):
        s = cls(x.T, x.sum(axis=0))
        s.build(2)
        assert s.size() == 1 and abs(s.weights()[0] - 1.0) < 1e-12 and s.error() < 1e-12   # one point is immediately optimal
    d = bc.DevicePhi.from_host(np.zeros((0, 3)))
    assert d.shape == (0, 3) and d.to_host().shape == (0, 3) and np.array_equal(d.colsum(), np.zeros(3))
    assert d.argmax(np.ones(3), mode=1)[0] == -1


def test_large_active_set_growth(bc):
    """More selected points than the initial device capacity (256 entries): lists, cached columns and the
    trace are re-allocated between build calls without losing state."""
    rng = np.random.RandomState(2)
    phi = rng.randn(4000, 150)
    from oracle import RefGIGA
    ref = RefGIGA(phi.T, phi.sum(axis=0)); ref.build(400)
    dev = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    for _ in range(4):
        dev.build(100)
    one = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    one.build(400)
    idx, val = dev.sparse_weights()
    i1, v1 = one.sparse_weights()
    assert len(idx) > 256
    assert np.array_equal(idx, i1) and np.array_equal(val, v1)        # incremental == one-shot, bit for bit
    ridx = np.where(ref.w > 0)[0]
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_allclose(val, ref.w[ridx], rtol=1e-5)
    f, st, er = dev._eng.trace()
    np.testing.assert_array_equal(f, [t[0] for t in ref.trace])


def test_pure_c_consumer_runs(tmp_path, bc):
    """A plain-C program (tests/c_abi_smoke.c) drives K1 -> K2 -> fused greedy loop through the C ABI alone."""
    import subprocess
    from test_abi_cpu import _build_c_consumer
    exe = _build_c_consumer(tmp_path)
    out = subprocess.run([exe, 'run'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'coreset of' in out.stdout


# ------------------------------------------------------------------ F10: sampling "solvers" against the reference
@pytest.mark.parametrize('nm', ['imp', 'unif'])
def test_f10_sampling_solvers_golden(bc, nm):
    """ImportanceSampling / UniformSampling (snnls/sampling.py:6-37): draws from the global NumPy RNG with
    probabilities proportional to the device row norms; sequence, weights, error() and the RNG position after 30
    draws equal the reference's -- directly and through HilbertCoreset(snnls=...) on a device projection."""
    g = load_golden('f10_sampling')
    phi, Z, th = g['phi'], g['Z'], g['th']
    cls = bc.snnls.ImportanceSampling if nm == 'imp' else bc.snnls.UniformSampling
    steps = g[nm + '_sel'].shape[0]
    np.random.seed(100)
    s = cls(phi.T, phi.sum(axis=0))
    picks = []
    orig = s._select

    def logged():
        f = orig()
        picks.append(int(f))
        return f
    s._select = logged
    for m in range(steps):
        s.build(1)
        assert picks[-1] == g[nm + '_sel'][m]
        np.testing.assert_allclose(s.weights(), g[nm + '_W'][m], rtol=WTOL, atol=0)
        np.testing.assert_allclose(s.error(), g[nm + '_err'][m], rtol=1e-9)
        assert bool(s.reached_numeric_limit) == bool(g[nm + '_lim'][m])
    assert np.random.rand() == float(g[nm + '_rng_after'])
    np.random.seed(101)
    prj = bc.DeviceProjector(lambda n, w, p: th, th.shape[0], bc.likelihoods.LinearRegression(1.0))
    h = bc.HilbertCoreset(Z, prj, snnls=cls)
    h.build(steps, steps)
    wts, pts, idcs = h.get()
    np.testing.assert_array_equal(idcs, g[nm + '_h_idcs'])
    np.testing.assert_allclose(wts, g[nm + '_h_wts'], rtol=WTOL)
    np.testing.assert_allclose(h.error(), float(g[nm + '_h_err']), rtol=1e-9)
    assert np.array_equal(pts, Z[idcs])
