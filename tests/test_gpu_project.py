"""GPU parity of K1 (projection kernel, through bc_project) against the oracle's
formulas and the golden fixtures generated from the reference.

Tolerance (written here as the north_star asks): Phi is a row-centred difference of
values of magnitude max|f|; the device contraction sums D products in a different order
and uses the ROCm libm (exp/log1p/pow <= 2 ulp), so |Phi_dev - Phi_ref| <= 1e-11 * (1 + max|f|)
element-wise.  Downstream selections must still be bit-exact, weights within 1e-5."""
import numpy as np
import pytest

import os

from conftest import numpy_uses_svml_exp, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import models_ref as M
from oracle import coreset_ref as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bc():
    import beta_cores_amd as bc
    bc.default_context()
    return bc


def centred(raw):
    return raw - raw.mean(axis=1)[:, None]


def check_phi(dev_phi, raw, tol=1e-11):
    ref = centred(raw)
    got = np.asarray(dev_phi)
    scale = 1. + np.abs(raw).max()
    assert got.shape == ref.shape
    assert np.all(np.isfinite(got))
    assert np.abs(got - ref).max() <= tol * scale, np.abs(got - ref).max() / scale
    np.testing.assert_allclose(dev_phi.colsum(), got.sum(axis=0), rtol=1e-10, atol=1e-9 * scale)
    np.testing.assert_allclose(dev_phi.norms(), np.sqrt((got ** 2).sum(axis=1)), rtol=1e-12, atol=1e-300)


def fixed(th):
    return lambda n, w, p: th


def test_f2_linreg(bc):
    g = load_golden('f2_formulas')
    Z, th = g['lin_Z'], g['lin_th']
    for sig in (1.0, 2.5):
        prj = bc.DeviceBetaProjector(fixed(th), th.shape[0], bc.likelihoods.LinearRegression(sig))
        check_phi(prj.project(Z), g['lin_ll_sig%g' % sig])
        for beta in (0.1, 0.2, 0.5):
            check_phi(prj.project_f(Z, beta), g['lin_bl_sig%g_b%g' % (sig, beta)])


def test_f2_logistic_incl_overflow_branches(bc):
    g = load_golden('f2_formulas')
    Z, th = g['log_Z'], g['log_th']
    prj = bc.DeviceBetaProjector(fixed(th), th.shape[0], bc.likelihoods.LogisticRegression())
    check_phi(prj.project(Z), g['log_ll'])                        # rows with |m| = 120, 800 exercise the m<100 branch
    for beta in (0.1, 0.2, 0.5):
        check_phi(prj.project_f(Z, beta), g['log_bl_b%g' % beta])   # exp overflow -> inf -> pow(inf, -b) = 0, no NaN


def test_f22_logistic_beta_exp_overflow_with_small_beta(bc):
    """Past np.exp's overflow (m > 709.78) the reference's (1 + inf)**(-beta) is exactly 0: at beta = 0.01 the value jumps by
    0.083 there, which the library's first power reproduces with the same cutoff (csrc/bc_k1_math.h); golden F22, rows with
    margins 700 ... 1500 and -700 ... -1500 on both sides of the jump, beta in {0.01, 0.05, 0.1}."""
    g = load_golden('f22_logistic_beta_overflow')
    Z, th = g['Z'], g['th']
    prj = bc.DeviceBetaProjector(fixed(th), th.shape[0], bc.likelihoods.LogisticRegression())
    for beta in (0.01, 0.05, 0.1):
        check_phi(prj.project_f(Z, beta), g['bl_b%g' % beta])

def test_f2_gaussian_location(bc):
    g = load_golden('f2_formulas')
    X, th = g['gau_X'], g['gau_th']
    for nm in ('iso', 'full'):
        model = bc.likelihoods.GaussianLocation(g['gau_%s_Siginv' % nm], float(g['gau_%s_logdet' % nm]))
        prj = bc.DeviceBetaProjector(fixed(th), th.shape[0], model)
        check_phi(prj.project(X), g['gau_%s_ll' % nm], tol=1e-10)
        for beta in (0.1, 0.5):
            bl, bg = prj.project_f(X, beta, grad=True)
            check_phi(bl, g['gau_%s_bl_b%g' % (nm, beta)], tol=1e-10)
            check_phi(bg, g['gau_%s_bg_b%g' % (nm, beta)], tol=1e-10)


# a spread of shapes that touches every NT template (S<=64, <=112, <=208, <=256), both JT
# variants, partial D-chunks, partial tiles and single-row / single-sample inputs
_RAGGED = [(1, 1, 1), (2, 3, 16), (127, 31, 50), (128, 32, 64), (129, 33, 100), (1000, 64, 100), (1000, 100, 112),
           (129, 64, 113), (1000, 33, 200), (127, 100, 256), (1000, 1, 256), (128, 3, 113), (2, 100, 200),
           (1, 64, 100), (1000, 32, 16), (129, 31, 1)]


@pytest.mark.parametrize('n,d,s', _RAGGED)
def test_ragged_shapes_linreg(bc, n, d, s):
    rng = np.random.RandomState(n * 1000 + d * 10 + s)
    Z = rng.randn(n, d + 1)
    th = rng.randn(s, d) * 0.5
    prj = bc.DeviceBetaProjector(fixed(th), s, bc.likelihoods.LinearRegression(1.3))
    check_phi(prj.project(Z), M.linreg_loglik(Z, th, 1.3))
    check_phi(prj.project_f(Z, 0.3), M.linreg_beta_lik(Z, th, 0.3, 1.3))


def test_gaussian_location_large_d(bc):
    """d = 100 (the reference example's dimension, zellner_gaussian/main.py:37): Siginv no longer fits the
    LDS staging of the row quadratic form."""
    rng = np.random.RandomState(100)
    n, d, s = 700, 100, 200
    A = rng.randn(d, d) * 0.1
    Sig = A.dot(A.T) + 5. * np.eye(d)
    Siginv, logdet = np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]
    X = rng.randn(n, d) * 2.
    th = rng.randn(s, d)
    prj = bc.DeviceBetaProjector(fixed(th), s, bc.likelihoods.GaussianLocation(Siginv, logdet))
    check_phi(prj.project(X), M.gauss_loglik(X, th, Siginv, logdet), tol=1e-10)
    check_phi(prj.project_f(X, 0.1), M.gauss_beta_lik(X, th, 0.1, Siginv, logdet), tol=1e-10)


@pytest.mark.parametrize('d,s', [(16, 100), (128, 100), (40, 200)])
def test_logistic_random(bc, d, s):
    rng = np.random.RandomState(d + s)
    n = 3000
    X = rng.randn(n, d)
    y = np.where(rng.rand(n) < 0.5, 1., -1.)
    Z = y[:, None] * X
    th = rng.randn(s, d) / np.sqrt(d) * 3
    prj = bc.DeviceBetaProjector(fixed(th), s, bc.likelihoods.LogisticRegression())
    check_phi(prj.project(Z), M.logistic_loglik(Z, th))
    check_phi(prj.project_f(Z, 0.1), M.logistic_beta_lik(Z, th, 0.1))


@pytest.mark.parametrize('n,d,s', [(300, 20, 300), (129, 33, 513), (1000, 64, 257), (5, 3, 1000)])
def test_wide_projection_dimension(bc, n, d, s):
    """S > 256 runs as passes of <= 256 samples (un-centred) plus one centring pass over Phi."""
    rng = np.random.RandomState(n + s)
    Z = rng.randn(n, d + 1)
    th = rng.randn(s, d) * 0.5
    prj = bc.DeviceBetaProjector(fixed(th), s, bc.likelihoods.LinearRegression(0.9))
    check_phi(prj.project(Z), M.linreg_loglik(Z, th, 0.9))
    check_phi(prj.project_f(Z, 0.2), M.linreg_beta_lik(Z, th, 0.2, 0.9))
    Zl = rng.randn(n, d) * 0.4
    prl = bc.DeviceProjector(fixed(th), s, bc.likelihoods.LogisticRegression())
    check_phi(prl.project(Zl), M.logistic_loglik(Zl, th))
    # and the solver runs on it (S-vector kernels are generic in S)
    phi = prj.project(Z)
    g = bc.snnls.GIGA(phi.T, phi.colsum())
    g.build(min(n, 5))
    from oracle import RefGIGA
    P = np.asarray(phi)
    ref = RefGIGA(P.T, P.sum(axis=0)); ref.build(min(n, 5))
    np.testing.assert_array_equal(g.sparse_weights()[0], np.where(ref.w > 0)[0])


def test_bad_shapes_are_rejected(bc):
    with pytest.raises(ValueError):
        bc.DeviceProjector(fixed(np.zeros((8, 4))), 8, bc.likelihoods.LinearRegression(1.0)).project(np.zeros((10, 7)))


# ------------------------------------------------------------------ pipelines: K1 -> K2 -> fused greedy loop
@pytest.mark.parametrize('nm', ['ll', 'bl'])
@pytest.mark.parametrize('alg', ['giga', 'fw', 'omp'])
def test_f3_hilbert_pipeline(bc, nm, alg):
    g = load_golden('f3_hilbert_linreg')
    Z, th = g['Z'], g['th']
    model = bc.likelihoods.LinearRegression(1.0)
    if nm == 'll':
        prj = bc.DeviceProjector(fixed(th), th.shape[0], model)
    else:
        class BetaAsPlain(bc.DeviceBetaProjector):          # HilbertCoreset calls .project(); bind beta = 0.1
            def project(self, pts, grad=False):
                return self.project_f(pts, 0.1)
        prj = BetaAsPlain(fixed(th), th.shape[0], model)
    cls = dict(giga=bc.snnls.GIGA, fw=bc.snnls.FrankWolfe, omp=bc.snnls.OrthoPursuit)[alg]
    key = '%s_%s_' % (nm, alg)
    steps = g[key + 'sel'].shape[0]
    h = bc.HilbertCoreset(Z, prj, snnls=cls)
    np.testing.assert_allclose(h.snnls.b, g[key + 'b'], rtol=1e-9, atol=1e-9)
    for m in range(steps):
        h.build(1, m + 1)
        np.testing.assert_allclose(h.error(), g[key + 'err'][m], rtol=1e-6)
    wts, pts, idcs = h.get()
    np.testing.assert_array_equal(idcs, g[key + 'idcs'])
    np.testing.assert_allclose(wts, g[key + 'wts'], rtol=1e-5)
    assert np.array_equal(pts, Z[idcs])
    if alg != 'omp':
        f, st, er = h.snnls._eng.trace()
        np.testing.assert_array_equal(f, g[key + 'sel'])
    h.optimize()
    wts, pts, idcs = h.get()
    np.testing.assert_array_equal(idcs, g[key + 'opt_idcs'])
    np.testing.assert_allclose(wts, g[key + 'opt_wts'], rtol=1e-5)
    with pytest.raises(ValueError):
        h.build(1, h.size() - 1 if h.size() > 0 else -1)          # coreset.py:38-39 cannot shrink


def test_f4_logistic_and_gauss_pipelines(bc):
    g = load_golden('f4_hilbert_logistic_gauss')
    Z, th = g['log_Z'], g['log_th']

    class BetaAsPlain(bc.DeviceBetaProjector):
        def project(self, pts, grad=False):
            return self.project_f(pts, 0.1)
    for nm, prj in (('ll', bc.DeviceProjector(fixed(th), th.shape[0], bc.likelihoods.LogisticRegression())),
                    ('bl', BetaAsPlain(fixed(th), th.shape[0], bc.likelihoods.LogisticRegression()))):
        key = 'log_%s_' % nm
        steps = g[key + 'sel'].shape[0]
        h = bc.HilbertCoreset(Z, prj)
        h.build(steps, steps)
        wts, pts, idcs = h.get()
        np.testing.assert_array_equal(idcs, g[key + 'idcs'])
        np.testing.assert_allclose(wts, g[key + 'wts'], rtol=1e-5)
        np.testing.assert_array_equal(h.snnls._eng.trace()[0], g[key + 'sel'])
    # config-1 plumbing shape, shrunk: Gaussian location, d=8, S=200, outlier clusters
    X, thg = g['gau_X'], g['gau_th']
    model = bc.likelihoods.GaussianLocation(g['gau_Siginv'], float(g['gau_logdet']))
    h = bc.HilbertCoreset(X, bc.DeviceProjector(fixed(thg), thg.shape[0], model))
    steps = g['gau_ll_sel'].shape[0]
    h.build(steps, steps)
    wts, pts, idcs = h.get()
    np.testing.assert_array_equal(idcs, g['gau_ll_idcs'])
    np.testing.assert_allclose(wts, g['gau_ll_wts'], rtol=1e-5)
    np.testing.assert_array_equal(h.snnls._eng.trace()[0], g['gau_ll_sel'])


def test_blackbox_projector_still_works(bc):
    """The reference's own projector class (host callable) feeding the device solver."""
    g = load_golden('f3_hilbert_linreg')
    Z, th = g['Z'], g['th']
    prj = bc.BlackBoxProjector(fixed(th), th.shape[0], lambda z, t: M.linreg_loglik(z, t, 1.0))
    h = bc.HilbertCoreset(Z, prj)
    h.build(50, 50)
    wts, pts, idcs = h.get()
    np.testing.assert_array_equal(idcs, g['ll_giga_idcs'])
    np.testing.assert_allclose(wts, g['ll_giga_wts'], rtol=1e-5)


def test_zero_rows_follow_reference_index_quirk(bc):
    """hilbert.py:16 drops all-zero rows and then indexes the filtered matrix (hilbert.py:32).  S = 16 makes every
    constant row exactly zero (the pairwise mean of 16 equal doubles is exact); test_f12_* covers S = 100 / 200, where
    most constant rows keep a rounding residue and are NOT dropped."""
    rng = np.random.RandomState(9)
    n, d, s = 600, 5, 16
    Z = rng.randn(n, d + 1)
    th = np.tile(rng.randn(1, d), (s, 1))
    th[:, 0] += rng.randn(s) * 0.3                                 # only x_0 distinguishes the samples
    Z[[3, 100, 101, 400], 0] = 0.                                  # -> f(z, th_s) constant in s -> centred row == 0
    ll = lambda z, t: M.linreg_loglik(z, t, 1.0)
    ref = C.RefHilbert(Z, ll, th)
    assert ref.vecs.shape[0] == n - 4
    ref.build(15, 15)
    h = bc.HilbertCoreset(Z, bc.DeviceProjector(fixed(th), s, bc.likelihoods.LinearRegression(1.0)))
    h.build(15, 15)
    wts, pts, idcs = h.get()
    np.testing.assert_array_equal(idcs, ref.idcs)
    np.testing.assert_allclose(wts, ref.wts, rtol=1e-5)
    assert np.array_equal(pts, Z[idcs])


@pytest.mark.parametrize('S', [100, 200])
@pytest.mark.parametrize('kind', ['lin', 'log'])
def test_f12_constant_rows_follow_numpy_mean(bc, S, kind):
    """Rows with all-zero features project to S equal numbers c.  The reference subtracts NumPy's pairwise mean, so the
    row becomes the constant c - mean (non-zero for ~3/4 of the rows at S = 100) and STAYS in the matrix; only rows whose
    residue is exactly 0 are dropped and shift later indices (hilbert.py:16,32).  Device Phi of those rows must carry the
    reference's bits, the kept / dropped split, the selections and the returned idcs must be the reference's."""
    g = load_golden('f12_constant_rows')
    za = g['zero_at']
    tag = '%s_S%d_' % (kind, S)
    Z, th = g[tag + 'Z'], g[tag + 'th']
    model = bc.likelihoods.LinearRegression(1.0) if kind == 'lin' else bc.likelihoods.LogisticRegression()
    prj = bc.DeviceProjector(fixed(th), S, model)
    phi = prj.project(Z)
    host = np.asarray(phi)
    assert np.array_equal(host[za], g[tag + 'phi_const'])              # bit-exact: c - np.mean(S copies of c)
    kept = phi.norms() > 0.
    assert np.array_equal(kept, g[tag + 'kept'])
    assert phi.norm_stats()[0] == int((~g[tag + 'kept']).sum())
    for an, cls in (('giga', bc.snnls.GIGA), ('fw', bc.snnls.FrankWolfe)):
        if tag + an + '_sel' not in g.files:
            continue
        steps = g[tag + an + '_sel'].shape[0]
        h = bc.HilbertCoreset(Z, prj, snnls=cls)
        h.build(steps, steps)
        wts, pts, idcs = h.get()
        np.testing.assert_array_equal(idcs, g[tag + an + '_idcs'])
        np.testing.assert_allclose(wts, g[tag + an + '_wts'], rtol=1e-5)
        assert np.array_equal(pts, Z[idcs])
        # the trace holds row numbers of the un-filtered matrix; the reference's are those of the filtered one
        sel = h.snnls._eng.trace()[0]
        shift = np.cumsum(~g[tag + 'kept'])
        np.testing.assert_array_equal(sel - shift[sel], g[tag + an + '_sel'])


def const_row_ok(dev_row, c, S):
    """A constant row on the device: all S entries equal, and equal to c' - np.mean(S copies of c') for a c' within
    2 ulp of the reference's c (the ROCm libm and glibc may round exp / log1p differently)."""
    if not np.all(dev_row == dev_row[0]):
        return False
    cands = [c]
    for _ in range(2):
        cands = [np.nextafter(cands[0], -np.inf)] + cands + [np.nextafter(cands[-1], np.inf)]
    return any(dev_row[0] == cc - np.full(S, cc).mean() for cc in cands)


def test_constant_rows_wide_projection_and_other_models(bc):
    """The same rule through the S > 256 path (centring pass k_center_tiles) and for the beta / Gaussian models,
    against NumPy's own mean (the oracle projection)."""
    rng = np.random.RandomState(12)
    n, d = 300, 6
    za = np.array([0, 5, 128, 129, 299])
    for S in (97, 112, 130, 256, 300, 515):
        Z = rng.randn(n, d + 1)
        Z[za, :d] = 0.
        th = rng.randn(S, d) * 0.4
        prj = bc.DeviceBetaProjector(fixed(th), S, bc.likelihoods.LinearRegression(1.7))
        raw = M.linreg_loglik(Z, th, 1.7)                          # no transcendental: c is reproduced bit for bit
        ref = raw - raw.mean(axis=1)[:, None]
        got = prj.project(Z)
        assert np.array_equal(np.asarray(got)[za], ref[za]), S
        assert np.array_equal(got.norms() > 0, np.sqrt((ref ** 2).sum(axis=1)) > 0), S
        rawb = M.linreg_beta_lik(Z, th, 0.2, 1.7)                   # one exp: restated with NumPy's own bits for constant rows
        refb = rawb - rawb.mean(axis=1)[:, None]
        devb = prj.project_f(Z, 0.2)
        hostb = np.asarray(devb)
        if numpy_uses_svml_exp():
            assert np.array_equal(hostb[za], refb[za]), S           # (bc_np_exp.h; tests/test_np_exp_cpu.py)
            assert np.array_equal(devb.norms() > 0, np.sqrt((refb ** 2).sum(axis=1)) > 0), S
        for r in za:
            assert const_row_ok(hostb[r], rawb[r, 0], S), (S, r)
    # Gaussian location: x = 0 rows are not constant (theta^T Siginv theta varies) -- but far outliers under the
    # beta-likelihood are: exp(-beta q / 2) underflows and every sample gives -(1+beta)^(-d/2-1)
    Sig = 500. * np.eye(d)
    Si, ld = np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]
    X = rng.multivariate_normal(np.zeros(d), Sig, n)
    X[za] += 4000.
    for S in (40, 100):
        thg = rng.randn(S, d) * 3.
        prj = bc.DeviceBetaProjector(fixed(thg), S, bc.likelihoods.GaussianLocation(Si, ld))
        raw = M.gauss_beta_lik(X, thg, 0.3, Si, ld)
        assert np.all(raw[za] == raw[za][:, :1])
        host = np.asarray(prj.project_f(X, 0.3))
        for r in za:
            assert const_row_ok(host[r], raw[r, 0], S), (S, r)


def test_group_sum_matches_numpy_bits(bc):
    """bc_phi_group_sum: per-group sums of Phi rows, accumulated in member order -- bit-identical to
    np.array([phi[g].sum(axis=0) for g in groups]) (bcores.py:46-50), for contiguous, scattered, repeated-member,
    single-row and empty groups; the result is a regular DevicePhi (norms, column sums, argmax work on it)."""
    rng = np.random.RandomState(5)
    n, s = 3001, 37
    phi = rng.randn(n, s) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))
    dev = bc.DevicePhi.from_host(phi)
    groups = [list(range(0, 130)), list(range(130, 131)), [], list(rng.choice(n, 400, replace=False)),
              [5, 5, 5, 2999, 3000], list(range(2000, 3001))]
    got = dev.group_sum(groups)
    want = np.array([phi[g].sum(axis=0) if len(g) else np.zeros(s) for g in groups])
    assert got.shape == (len(groups), s)
    assert np.array_equal(np.asarray(got), want)
    np.testing.assert_allclose(got.norms(), np.linalg.norm(want, axis=1), rtol=1e-14)
    np.testing.assert_allclose(got.colsum(), want.sum(axis=0), rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        dev.group_sum([[0, n]])


# ------------------------------------------------------------------ projector residency rules (round 2)
def test_project_results_never_alias(bc):
    """Every project() returns its own buffers while an earlier result is alive (the reference returns a fresh array,
    projector.py:24): a solver built on the first Phi must not see the second projection."""
    rng = np.random.RandomState(31)
    n, d, s = 6000, 7, 24
    Z = rng.randn(n, d + 1)
    ths = [rng.randn(s, d) * 0.4, rng.randn(s, d) * 0.4]
    cur = [0]
    prj = bc.DeviceProjector(lambda k, w, p: ths[cur[0]], s, bc.likelihoods.LinearRegression(1.0))
    phi1 = prj.project(Z)
    host1 = np.asarray(phi1).copy()
    b1 = phi1.sum(axis=0)
    solver = bc.snnls.GIGA(phi1.T, b1)
    solver.build(5)
    cur[0] = 1
    prj.update(np.array([]), np.array([]))
    phi2 = prj.project(Z)
    assert phi2.h.value != phi1.h.value
    assert np.array_equal(np.asarray(phi1), host1)                      # untouched by the second projection
    assert not np.array_equal(np.asarray(phi2), host1)
    ref = bc.snnls.GIGA(host1.T, b1)
    ref.build(15)
    solver.build(10)
    assert np.array_equal(solver.sparse_weights()[0], ref.sparse_weights()[0])
    assert np.array_equal(solver.sparse_weights()[1], ref.sparse_weights()[1])


def test_live_arrays_are_read_and_pins_are_loud(bc):
    """Default: project(ndarray) reads the LIVE array (an in-place edit is seen, like projector.py:24).  pin() keeps
    a device copy and makes the host array read-only for as long as it is pinned."""
    rng = np.random.RandomState(32)
    n, d, s = 5000, 5, 16
    Z = rng.randn(n, d + 1)
    th = rng.randn(s, d) * 0.3
    prj = bc.DeviceProjector(fixed(th), s, bc.likelihoods.LinearRegression(1.0))
    a = np.asarray(prj.project(Z)).copy()
    Z[10] += 1.0                                                        # in-place edit of a >= 4096-row array
    b = np.asarray(prj.project(Z))
    assert not np.array_equal(a[10], b[10]) and np.array_equal(np.delete(a, 10, 0), np.delete(b, 10, 0))
    check_phi(prj.project(Z), M.linreg_loglik(Z, th, 1.0))
    dd = prj.pin(Z)
    assert prj.pin(Z) is dd and not Z.flags.writeable
    with pytest.raises(ValueError):
        Z[11] += 1.0                                                    # loud instead of silently stale
    c = np.asarray(prj.project(Z))
    assert np.array_equal(b, c)
    prj.unpin(Z)
    assert not Z.flags.writeable            # pins nest (pin was called twice above)
    prj.unpin(Z)
    assert Z.flags.writeable
    Z[11] += 1.0
    assert not np.array_equal(np.asarray(prj.project(Z))[11], c[11])
    # a coreset that re-projects all rows per gradient step pins them itself and lets go when it dies
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=2)
    assert not Z.flags.writeable
    alg.build(1, 1)
    del alg
    import gc
    gc.collect()
    assert Z.flags.writeable


def test_view_data_is_copied_once_with_a_warning(bc):
    """A view (Z[:n]) cannot be made read-only through its base, so it is not pinned -- but a coreset that re-projects all
    rows per gradient must not upload them per gradient either: the rows are copied to the device once, with a UserWarning."""
    rng = np.random.RandomState(35)
    n, d, s = 6000, 5, 16
    base = rng.randn(n + 100, d + 1)
    Z = base[:n]
    assert Z.base is base
    th = rng.randn(s, d) * 0.3
    prj = bc.DeviceProjector(fixed(th), s, bc.likelihoods.LinearRegression(1.0))
    with pytest.warns(UserWarning, match='view'):
        alg = bc.SparseVICoreset(Z, prj, opt_itrs=3)
    assert alg._dev_data is not None and alg._dev_data.shape == (n, d + 1) and Z.flags.writeable
    ref = bc.SparseVICoreset(Z.copy(), prj, opt_itrs=3)                 # an owning array: the pinned path
    for m in range(3):
        alg.build(1, m + 1)
        ref.build(1, m + 1)
    assert np.array_equal(alg.idcs, ref.idcs) and np.array_equal(alg.wts, ref.wts)
    with pytest.warns(UserWarning, match='view'):
        bc.BatchPSVICoreset(Z, prj, opt_itrs=2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        bc.SparseVICoreset(Z, prj, opt_itrs=3, pin_data=False)          # opting out: no copy, no warning


def test_repeated_large_subsamples_do_not_grow_device_memory(bc):
    """BetaCoreset / SparseVI with n_subsample >= 4096 project a NEW data[sub_idcs] array per gradient step: its
    device copy and its Phi must be recycled, not accumulated."""
    import torch
    rng = np.random.RandomState(33)
    n, d, s = 40000, 6, 32
    Z = rng.randn(n, d + 1)
    th = rng.randn(s, d) * 0.3
    prj = bc.DeviceBetaProjector(fixed(th), s, bc.likelihoods.LinearRegression(1.0))
    free = []
    for it in range(40):
        sub = Z[rng.randint(n, size=10000)]
        phi = prj.project_f(sub, 0.2)
        _ = phi.colsum()
        del phi, sub
        if it in (9, 39):
            bc.default_context().sync()
            free.append(torch.cuda.mem_get_info()[0])
    assert free[0] - free[1] < 8 * 2 ** 20, (free[0] - free[1]) / 2 ** 20      # < 8 MiB drift over 30 projections


def test_device_tolerance_follows_util_tol(bc):
    """The reference reads util.TOL at every use (giga.py:28): set_tolerance() after a solver exists must reach the
    device-side numeric-limit checks."""
    rng = np.random.RandomState(34)
    phi = rng.randn(3000, 20)
    phi -= phi.mean(axis=1)[:, None]
    s = bc.snnls.GIGA(phi.T, phi.sum(axis=0))
    s.build(3)
    assert s.size() == 3 and not s.reached_numeric_limit
    old = bc.util.TOL
    try:
        bc.util.set_tolerance(10.)            # ||cdir|| <= 1 < TOL: every _select now raises (giga.py:28-29)
        s.build(3)
        assert s.reached_numeric_limit and s.size() == 3
    finally:
        bc.util.set_tolerance(old)


@pytest.mark.parametrize('S', [97, 100])
def test_nearly_constant_rows_are_not_flattened(bc, S):
    """K1 (96 < S <= 100) finds constant rows AFTER centring: a vanishing norm makes a row a suspect, an exact check on
    the centred values decides.  Rows whose S values differ only in their last digits are suspects too and must come
    out untouched: centred with their own mean, not constant, norm > 0 -- next to truly constant rows (which take
    NumPy's rounded mean) and ordinary ones in the same tile and wave."""
    rng = np.random.RandomState(S)
    n, d = 700, 7
    Z = rng.randn(n, d + 1)
    th = rng.randn(S, d)
    near = np.arange(5, n, 37)
    Z[near, :d] = 1e-13 * rng.randn(near.size, d)              # spread of f over the samples ~1e-13 of its size
    Z[near, d] = 2.                                            # (y away from 0: the spread is 2 y x.th / (2 sigsq))
    const = np.arange(11, n, 53)
    Z[const, :d] = 0.
    phi = bc.DeviceProjector(fixed(th), S, bc.likelihoods.LinearRegression(1.0)).project(Z)
    dev, nrm = phi.to_host(), phi.norms()
    ref = M.linreg_loglik(Z, th, 1.0)
    c = ref[:, 0].copy()
    ref -= ref.mean(axis=1)[:, None]
    plain = np.setdiff1d(np.arange(n), np.concatenate((near, const)))
    np.testing.assert_allclose(dev[plain], ref[plain], rtol=0., atol=1e-11 * (1. + np.abs(ref).max()))
    for i in const:
        np.testing.assert_array_equal(dev[i], ref[i])          # the reference's bits (pairwise mean of S copies of c)
    for i in near:
        assert not np.all(dev[i] == dev[i, 0])                  # still a row with structure
        assert np.abs(dev[i] - ref[i]).max() <= 8 * np.spacing(abs(c[i]))        # the means may differ by an ulp or two of c
        assert np.abs(dev[i]).max() > 50 * np.spacing(abs(c[i]))                 # ... which is far below the row's own size
        assert nrm[i] > 0. and abs(nrm[i] - np.sqrt((dev[i] ** 2).sum())) <= 1e-12 * nrm[i]


def test_logistic_models_special_values(bc):
    """The logistic projections evaluate log1p(exp(m)) and exp() with their own short implementations
    (csrc/bc_project.hip: bc_log1p_exp_neg, bc_exp_nonpos): saturation, the branch at m = 100, huge |m| and NaN rows
    must behave like the reference's expressions."""
    rng = np.random.RandomState(3)
    S, d = 100, 4
    th = rng.randn(S, d)
    th[:, 0] = np.abs(th[:, 0]) + 0.5
    Z = rng.randn(64, d)
    Z[1, :] = 0.
    Z[2, :] = [-150., 0., 0., 0.]          # m = +150 th0 > 100 on most samples: the linear branch
    Z[3, :] = [900., 0., 0., 0.]           # m very negative: log1p(exp(m)) underflows to 0
    Z[4, :] = [-99.9, 0., 0., 0.]
    Z[5, 2] = np.nan
    for beta, raw in ((None, M.logistic_loglik(Z, th)), (0.3, M.logistic_beta_lik(Z, th, 0.3))):
        prj = bc.DeviceBetaProjector(fixed(th), S, bc.likelihoods.LogisticRegression())
        got = np.asarray(prj.project(Z) if beta is None else prj.project_f(Z, beta))
        ref = raw - raw.mean(axis=1)[:, None]
        ok = np.ones(64, dtype=bool)
        ok[5] = False
        assert np.all(np.isnan(got[5])) and np.all(np.isnan(ref[5]))
        np.testing.assert_allclose(got[ok], ref[ok], rtol=0., atol=1e-11 * (1. + np.abs(raw[ok]).max()))
        # element-wise relative accuracy where the values are not tiny
        big = ok[:, None] & (np.abs(raw) > 1e-3)
        rel = np.abs((got + raw.mean(axis=1)[:, None]) - raw)[big] / np.abs(raw)[big]
        assert rel.max() < 1e-12, rel.max()


# ------------------------------------------------------------------ Theta-resident K1 (round 3)
@pytest.mark.parametrize('n,d,s', [(300_000, 64, 100), (262_144 + 77, 37, 100), (270_000, 128, 64), (300_001, 20, 112),
                                   (280_000, 13, 16), (263_000, 160, 97)])
def test_theta_resident_kernel_matches_staged_and_oracle(bc, n, d, s):
    """Large shards (>= 2048 tiles) run k_project_r: Theta resident in LDS, B operands read straight from global memory,
    waves independent, per-wave column partials.  Against the staged kernel (BC_K1_STAGED=1, same process) on the same
    inputs -- values / norms / column sums agree to rounding (the two contract the D axis in a different order) -- and
    against the oracle on scattered rows; ragged N (dead rows in the last group), D not a multiple of 16 or 32 (masked
    columns), constant (zero-feature) rows, every model."""
    import os
    rng = np.random.RandomState(n % 1000 + d + s)
    pick = np.unique(np.concatenate(([0, 1, 31, 32, 127, 128, n - 33, n - 32, n - 1], rng.choice(n, 300, replace=False))))
    zero_rows = pick[5:8]
    Sig = np.diag(rng.uniform(0.5, 2.0, d))
    cases = [('linreg', bc.likelihoods.LinearRegression(1.3), d + 1, (None, 0.3),
              lambda Z, th, b: M.linreg_loglik(Z, th, 1.3) if b is None else M.linreg_beta_lik(Z, th, b, 1.3)),
             ('logistic', bc.likelihoods.LogisticRegression(), d, (None, 0.2),
              lambda Z, th, b: M.logistic_loglik(Z, th) if b is None else M.logistic_beta_lik(Z, th, b)),
             ('gauss', bc.likelihoods.GaussianLocation(np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]), d, (None, 0.5),
              lambda Z, th, b: M.gauss_loglik(Z, th, np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]) if b is None
              else M.gauss_beta_lik(Z, th, b, np.linalg.inv(Sig), np.linalg.slogdet(Sig)[1]))]
    for name, model, dz, betas, ref_fn in cases:
        Z = rng.randn(n, dz)
        Z[zero_rows, :d] = 0.
        th = rng.randn(s, d) * (0.6 / np.sqrt(d))
        prj = bc.DeviceBetaProjector(fixed(th), s, model)
        dd = bc.DeviceData(Z)
        for beta in betas:
            run = lambda: prj.project(dd) if beta is None else prj.project_f(dd, beta)
            os.environ.pop('BC_K1_STAGED', None)
            res = run()
            rows_r, norms_r, cs_r = res.rows(pick), res.norms(), res.colsum()
            sf = prj.colsum(dd, beta=beta)
            assert np.array_equal(sf, cs_r), name                       # store-free == materialised, bit for bit
            del res
            os.environ['BC_K1_STAGED'] = '1'
            try:
                old = run()
                rows_s, norms_s, cs_s = old.rows(pick), old.norms(), old.colsum()
                del old
            finally:
                os.environ.pop('BC_K1_STAGED', None)
            raw = ref_fn(Z[pick], th, beta)
            ref = raw - raw.mean(axis=1)[:, None]
            scale = 1. + np.abs(raw).max()
            assert np.abs(rows_r - ref).max() <= 1e-11 * scale, (name, beta)
            assert np.abs(rows_r - rows_s).max() <= 1e-12 * scale, (name, beta)
            np.testing.assert_allclose(norms_r, norms_s, rtol=1e-11, atol=1e-13 * scale)
            np.testing.assert_allclose(cs_r, cs_s, rtol=1e-9, atol=1e-9 * scale)
            np.testing.assert_allclose(norms_r[pick], np.sqrt((ref ** 2).sum(axis=1)), rtol=1e-9, atol=1e-12 * scale)
            if name != 'gauss':
                assert np.array_equal(rows_r[5:8] if False else res_const(rows_r, pick, zero_rows), res_const(rows_s, pick, zero_rows))


def res_const(rows, pick, zero_rows):
    """the rows of `rows` (gathered at indices `pick`) that belong to the constant (zero-feature) data rows"""
    return rows[np.isin(pick, zero_rows)]


def test_abi_null_sweep_with_a_live_context():
    """Every entry point refuses NULL arguments, and a live context refuses malformed sizes (tests/abi_null_sweep.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tests', 'abi_null_sweep.py')], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'live-context checks ok' in res.stdout and 'swept 8' in res.stdout, res.stdout + res.stderr[-2000:]


# ------------------------------------------------------------------ round 5: constant rows evaluated on the host
def test_constant_rows_take_host_evaluated_values(bc, monkeypatch):
    """On a host whose NumPy does not evaluate np.exp with the routine csrc/bc_np_exp.h restates, the projector evaluates
    the constants of all-zero-feature rows itself (LinearRegression.host_constants: the reference's expression,
    model_neurlinr.py:102-110) and hands (y, value) pairs to K1 (bc_ctx_set_constant_row_values).  The probe is forced to
    "not the restated NumPy" here: the host route must give the reference's rows bit for bit (this host's NumPy IS the
    reference's), through host arrays, resident rows and the store-free column sums -- and the kernel must really take its
    constants from the table (a table entry moved by one ulp moves the device row with it)."""
    import ctypes as C
    from beta_cores_amd.util import numpy_bits
    from beta_cores_amd import _native as N
    from beta_cores_amd.device import _ptr
    monkeypatch.setattr(numpy_bits, '_cached', False)
    monkeypatch.setattr(numpy_bits, '_warned', set())
    rng = np.random.RandomState(14)
    n, d, S = 5000, 6, 100
    za = np.array([0, 5, 128, 129, 4095, 4999])
    Z = rng.randn(n, d + 1)
    Z[za, :d] = 0.
    Z[129, d] = Z[5, d]                                   # two constant rows with the same y share a table entry
    th = rng.randn(S, d) * 0.4
    model = bc.likelihoods.LinearRegression(1.7)
    with pytest.warns(UserWarning, match='evaluated on the host'):
        prj = bc.DeviceBetaProjector(fixed(th), S, model)
    rawb = M.linreg_beta_lik(Z, th, 0.2, 1.7)
    refb = rawb - rawb.mean(axis=1)[:, None]
    for src in (Z, bc.DeviceData(Z)):
        devb = prj.project_f(src, 0.2)
        assert prj.constant_rows_from_host == 5           # six rows, five distinct y
        hostb = np.asarray(devb)
        assert np.array_equal(hostb[za], refb[za])
        assert np.array_equal(devb.norms() > 0, np.sqrt((refb ** 2).sum(axis=1)) > 0)
        np.testing.assert_array_equal(prj.colsum(src, beta=0.2), devb.colsum())
    # the plain log-likelihood needs no table (no transcendental in its constant)
    prj.project(Z)
    # ... and the kernel reads the table: nudge one entry by an ulp (inside the 1e-13 agreement guard) and the row follows
    params = np.ascontiguousarray(model.params(beta=0.2))
    keys = np.unique(Z[za, d])
    vals = model.host_constants(keys, 0.2)
    vals2 = vals.copy()
    k = int(np.searchsorted(keys, Z[128, d]))
    vals2[k] = np.nextafter(vals[k], np.inf)
    monkeypatch.setattr(prj, '_host_constants', False)    # keep the projector from restoring the table
    N.call('bc_ctx_set_constant_row_values', prj.ctx.h, int(model.beta_model_id), _ptr(params), int(params.shape[0]),
           _ptr(keys), _ptr(vals2), int(keys.size))
    try:
        row = np.asarray(prj.project_f(Z, 0.2))[128]
        c2 = vals2[k]
        assert np.all(row == c2 - np.full(S, c2).mean()) and not np.array_equal(row, refb[128])
        # other parameters than the table was made for: not used
        row_other = np.asarray(prj.project_f(Z, 0.3))[128]
        rawo = M.linreg_beta_lik(Z[128:129], th, 0.3, 1.7)
        assert np.array_equal(row_other, (rawo - rawo.mean(axis=1)[:, None])[0])      # (the library's restated exp: an AVX-512 host)
    finally:
        N.call('bc_ctx_set_constant_row_values', prj.ctx.h, 0, None, 0, None, None, 0)
    with pytest.raises(ValueError):                       # keys must be strictly increasing
        N.call('bc_ctx_set_constant_row_values', prj.ctx.h, int(model.beta_model_id), _ptr(params), int(params.shape[0]),
               _ptr(keys[::-1].copy()), _ptr(vals), int(keys.size))
